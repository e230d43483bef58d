#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "static CEED_MI355X_SCHED=static" "dynamic X=1"; do
  set -- $v; n=$1; shift
  rm -rf /tmp/kt_$n
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -- python3 $R/bench.py --nz 4 --steps 100 --warmup 5 --prewarm-ms 30 --no-cpu-baseline > $O/exp22_$n.json 2> $O/exp22_$n.err
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  echo "$n: $(grep -E 'k_assemble|k_fused_pencil<5, 5, 6' $f | awk -F, '{print $1, "avg", $(NF-4)/1000, "min", $(NF-2)/1000, "max", $(NF-1)/1000}' | sed -e 's/cps:://g; s/(.*)//' | tr '\n' ' ')"
  t=$(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1)
  python3 - $t <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_fused_pencil<5, 5, 6' in r['Kernel_Name'] or 'k_assemble' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[-12]['Start_Timestamp'])
for r in rows[-12:]:
    print('   ', 'fused' if 'fused' in r['Kernel_Name'] else 'asm  ', round((int(r['Start_Timestamp'])-t0)/1e3,1), round((int(r['End_Timestamp'])-t0)/1e3,1))
PY
done
