#!/bin/bash
# Swept-element path (GEO = 3) against the general per-point recompute
# (CEED_MI355X_SWEPT=0), alternating, same box: ms per step.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for w in ${WL:-c4 nz12 ss lin p2 mesh em box6 box6ss box64}; do
  case $w in c4) A="";; nz12) A="--nz 12 --steps 100";; ss) A="--problem hyperSS";; lin) A="--problem linElas";; p2) A="--degree 2";; mesh) A="--workload mesh";; em) A="--emulate-rank 3 --of 8 --steps 200 --warmup 10";;
    box6) A="--workload box --nr 32 --nth 32 --nz 32 --degree 6";; box6ss) A="--workload box --nr 32 --nth 32 --nz 32 --degree 6 --problem hyperSS";; box64) A="--workload box --nr 64 --nth 64 --nz 64 --degree 6 --steps 20";; esac
  line="$w:"
  for rep in 1 2; do for sw in 1 0; do
    CEED_MI355X_SWEPT=$sw timeout -k 10 200 python3 $R/bench.py $A --no-cpu-baseline --cold-idle-s 0 > $O/sw_$w.json 2> $O/sw_$w.err || { echo "$w failed"; tail -3 $O/sw_$w.err; }
    line="$line  on=$sw $(python3 -c "
import json; d=[json.loads(l) for l in open('$O/sw_$w.json') if l.startswith('{')][-1]; print('%.4f' % d['ms_per_step'])")"
  done; done
  echo "$line"
done
