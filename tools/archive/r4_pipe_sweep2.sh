#!/bin/bash
# Round 4: pipelined (3 segments, the default) against 2 segments and the serial form, alternating, three times, one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
run() { env "$@" timeout -k 10 200 python3 $R/bench.py --steps 100 --no-cpu-baseline --cold-idle-s 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep: default(3 seg) $(run A=1)  2 seg last 4 $(run CEED_MI355X_PIPE_SEGMENTS=2 CEED_MI355X_PIPE_LAST=4)  2 seg last 3 $(run CEED_MI355X_PIPE_SEGMENTS=2 CEED_MI355X_PIPE_LAST=3)  serial $(run CEED_MI355X_ASSEMBLE=serial)"
done
