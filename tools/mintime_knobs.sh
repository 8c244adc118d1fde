#!/bin/bash
# Sweep of the strategy constants of the min-time solve (RL_MT_* environment switches) on the 1024-instance batch:
#   bash tools/mintime_knobs.sh            (edit the list below; each line is one run with those switches set)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "$@"; env "$@" timeout -k 10 200 python tools/bench_mintime.py 1024 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   wall %.2f conv %d it mean %.1f max %.0f' % (d['wall_s'], d['converged'], d['iterations_mean'], d['iterations_max']))"; }
rob() { echo "robustness $@"; env "$@" timeout -k 10 600 python tools/mintime_robustness.py 2>/dev/null | python3 -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('   converged %d of %d; failing configs: %s' % (sum(r.get('converged',0) for r in rows), sum(r.get('of',0) for r in rows), [(r['track'],r['interval'],r['model'],r['of']-r['converged']) for r in rows if r.get('of',0)!=r.get('converged',0)]))"; }
for dc in 0 1 2 4; do
run RL_MT_DUAL_CAP=$dc
rob RL_MT_DUAL_CAP=$dc
done
