#!/bin/bash
# Persistent grid shrunk to equal group counts per wave (CEED_MI355X_BALANCE=1) against the full grid, mid-size launches.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for nz in 5 12 23 45 90; do for bal in 0 1 0 1; do
  CEED_MI355X_BALANCE=$bal python3 $R/bench.py --nz $nz --no-cpu-baseline --cold-idle-s 0 --steps 100 > $O/bal_${nz}_$bal.json 2> $O/bal_${nz}_$bal.err || { echo "nz $nz bal $bal failed"; continue; }
  python3 -c "
import json; d=json.loads(open('$O/bal_${nz}_$bal.json').read()); print('nz %3d  %6d hexes  balance %d  %8.4f ms  %7.2f GDoF/s' % ($nz, d['config']['elements_per_gpu'], $bal, d['ms_per_step'], d['value']/1e3))"
done; done
