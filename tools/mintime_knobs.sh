#!/bin/bash
# Sweep of the strategy constants of the min-time solve (RL_MT_* environment switches) on the 1024-instance batch.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "$@"; env "$@" timeout -k 10 200 python tools/bench_mintime.py 1024 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   wall %.2f conv %d it mean %.1f max %.0f' % (d['wall_s'], d['converged'], d['iterations_mean'], d['iterations_max']))"; }
run RL_MT_MU_KAPPA=30 RL_MT_D_UP=3
run RL_MT_MU_KAPPA=30 RL_MT_A_LO=0.2
run RL_MT_MU_KAPPA=30 RL_MT_D_UP=3 RL_MT_A_LO=0.2
run RL_MT_MU_KAPPA=20
run RL_MT_MU_KAPPA=50
run RL_MT_MU_KAPPA=30 RL_MT_MU_FAC=0.15
run RL_MT_MU_KAPPA=30 RL_MT_D_DOWN=0.35
run RL_MT_D_UP=2
run RL_MT_A_LO=0.1
run RL_MT_A_HI=0.8
