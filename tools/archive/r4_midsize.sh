#!/bin/bash
# Round 4: does the pipelined form pay below 20 rounds of the waves now?  49 500 and 25 300 hexes (the N = 2 / 4 rank sizes), serial (default) against two segments.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
run() { w=$1; shift; env "$@" timeout -k 10 250 python3 $R/bench.py $w --steps 100 --no-cpu-baseline --cold-idle-s 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f (%s)' % (d['ms_per_step'], d['config']['assembly'][:10]), end='')"; }
for rep in 1 2; do
  echo "rep $rep: 49 500 hexes default $(run "--nz 45" A=1)  forced 2 segments $(run "--nz 45" CEED_MI355X_PIPE_MIN_TOTAL=8 CEED_MI355X_PIPE_MIN_ROUNDS=2 CEED_MI355X_PIPE_SEGMENTS=2) | 25 300 hexes default $(run "--nz 23" A=1)  forced $(run "--nz 23" CEED_MI355X_PIPE_MIN_TOTAL=4 CEED_MI355X_PIPE_MIN_ROUNDS=2 CEED_MI355X_PIPE_SEGMENTS=2 CEED_MI355X_PIPE_LAST=2)"
done
