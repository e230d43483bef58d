#!/bin/bash
# Round 4: MB of E-vector per segment of the pipelined transpose (CEED_MI355X_PIPE_MB), alternating, one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
run() { w=$1; shift; env "$@" timeout -k 10 250 python3 $R/bench.py $w --no-cpu-baseline --cold-idle-s 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f (%s)' % (d['ms_per_step'], d['config']['assembly'].split(' on ')[0].replace('pipelined: ','')))"; }
for rep in 1 2; do
  for mb in 90 120 160; do
    echo "rep $rep  $mb MB:  config 4 $(run "--steps 100" CEED_MI355X_PIPE_MB=$mb)   2 x config 4 $(run "--nz 180" CEED_MI355X_PIPE_MB=$mb)   whole box $(run "--workload box --degree 6 --nr 64 --nth 64 --nz 64 --steps 20" CEED_MI355X_PIPE_MB=$mb)   hyperSS $(run "--problem hyperSS" CEED_MI355X_PIPE_MB=$mb)"
  done
done
