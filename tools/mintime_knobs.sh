#!/bin/bash
# Sweep of the strategy constants of the min-time solve (RL_MT_* environment switches) on the 1024-instance batch:
#   bash tools/mintime_knobs.sh            (edit the list below; each line is one run with those switches set)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "$@"; env "$@" timeout -k 10 200 python tools/bench_mintime.py 1024 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   wall %.2f conv %d it mean %.1f max %.0f' % (d['wall_s'], d['converged'], d['iterations_mean'], d['iterations_max']))"; }
run RL_MT_GROUPS=3
run RL_MT_MU0=0.02
run RL_MT_MU0=0.05
run RL_MT_MU0=0.2
run RL_MT_MU0=0.5
run RL_MT_MU_FAC=0.1
run RL_MT_MU_FAC=0.3
run RL_MT_MU_POW=1.3
run RL_MT_MU_KAPPA=10
run RL_MT_D_UP=5
run RL_MT_DELTA0=1e-2
