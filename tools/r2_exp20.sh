#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R && timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 -k "gated_assembly" > $O/pytest_gated.log 2>&1; rc=$?; echo "pytest gated rc $rc"; tail -3 $O/pytest_gated.log
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
for v in "serial X=1" "gated6 CEED_MI355X_ASSEMBLE=gated CEED_MI355X_ASM_WAVES=6" "gated8 CEED_MI355X_ASSEMBLE=gated CEED_MI355X_ASM_WAVES=8" "gated4 CEED_MI355X_ASSEMBLE=gated CEED_MI355X_ASM_WAVES=4" "serial2 X=1" "gated8b CEED_MI355X_ASSEMBLE=gated CEED_MI355X_ASM_WAVES=8"; do
  set -- $v; n=$1; shift
  rm -rf /tmp/kt_$n
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/exp20_$n.json 2> $O/exp20_$n.err
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  echo "$n: $(tail -1 $O/exp20_$n.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4))') $(grep -E 'k_assemble|k_fused_pencil<5, 5, 6' $f | awk -F, '{print $1, $(NF-4)/1000}' | sed -e 's/cps:://g; s/(.*)//' | tr '\n' ' ')"
done
