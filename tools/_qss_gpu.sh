set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_d_gputest.log 2>&1 || { echo TESTFAIL; tail -30 gpurun_out/r04_d_gputest.log; exit 1; }
tail -3 gpurun_out/r04_d_gputest.log
mkdir -p $R/gpurun_out/qprof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qprof -o q -- python3 $R/tools/time_qss.py 1 256 768 1024 > $R/gpurun_out/qprof/run.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/qprof/run.log | tail -30
