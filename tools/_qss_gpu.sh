set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 tools/time_qss.py 1 64 256 512 768 1024 > gpurun_out/qss_matrix.log 2>&1 || { echo TIMEFAIL; tail -5 gpurun_out/qss_matrix.log; exit 1; }
grep -v "amdgpu.ids" gpurun_out/qss_matrix.log
RL_QSS_DF=1 timeout -k 10 600 python3 tools/validate_qss.py 60 > gpurun_out/qss_df_val60.log 2>&1 || { echo VALFAIL; tail -20 gpurun_out/qss_df_val60.log; exit 1; }
tail -2 gpurun_out/qss_df_val60.log
