#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "serial CEED_MI355X_ASSEMBLE=serial" "folded X=1" "fold_nodrain CEED_MI355X_FOLD_DBG=1" "fold_noloop CEED_MI355X_FOLD_DBG=2" "fold_nothing CEED_MI355X_FOLD_DBG=3"; do
  set -- $v; n=$1; shift
  rm -rf /tmp/kt_$n
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/exp13_$n.json 2> $O/exp13_$n.err
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  echo "$n: $(tail -1 $O/exp13_$n.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4))') $(grep -E 'k_assemble|k_fused_pencil<5, 5, 6' $f | awk -F, '{print $1, $(NF-4)/1000}' | sed -e 's/cps:://g; s/(.*)//' | tr '\n' ' ')"
done
