#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R && timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest_gpu_pair.log 2>&1; rc=$?; echo "pytest gpu rc $rc"; tail -5 $O/pytest_gpu_pair.log
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
for v in "nopair CEED_MI355X_PAIR=0" "pair X=1" "nopair2 CEED_MI355X_PAIR=0" "pair2 X=1"; do
  set -- $v; n=$1; shift
  rm -rf /tmp/kt_$n
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/exp24_$n.json 2> $O/exp24_$n.err
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  echo "$n: $(tail -1 $O/exp24_$n.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4))') $(grep -E 'k_assemble|k_fused_pencil<5, 5, 6' $f | awk -F, '{print $1, $(NF-4)/1000}' | sed -e 's/cps:://g; s/(.*)//' | tr '\n' ' ')"
done
