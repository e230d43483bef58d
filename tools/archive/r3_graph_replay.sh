#!/bin/bash
# bench.py's event placement and the caller-recorded step (bench.py --graph), same box: emulated rank 3 of 8 (cylinder, box), config 4; us per step.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for w in cyl box c4; do
  case $w in cyl) A="--emulate-rank 3 --of 8 --steps 200 --warmup 10";; box) A="--emulate-rank 3 --of 8 --workload box --degree 6 --nr 64 --nth 64 --nz 64 --steps 100 --warmup 10";; c4) A="";; esac
  line="$w"
  for rep in 1 2; do for a in "" "--per-apply-events" "--graph"; do
    timeout -k 10 200 python3 $R/bench.py $A $a --no-cpu-baseline --cold-idle-s 0 > $O/g_$w.json 2> $O/g_$w.err || { echo "$w $a failed"; tail -3 $O/g_$w.err; }
    line="$line  [${a:-one event pair}] $(python3 -c "
import json; d=[json.loads(l) for l in open('$O/g_$w.json') if l.startswith('{')][-1]; print('%.1f (events %.1f)' % (1e3*d['ms_per_step'], d['roofline']['kernel_avg_us']))")"
  done; done
  echo "$line"
done
