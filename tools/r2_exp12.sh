#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R
CEED_MI355X_ASM_SPINS=3000 timeout -k 5 60 python -u tools/r2_diag2.py tiny > $O/diag_fold_tiny.log 2>&1; rc=$?; echo "fold tiny rc $rc"; tail -3 $O/diag_fold_tiny.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
CEED_MI355X_ASM_SPINS=3000 timeout -k 5 60 python -u tools/r2_diag2.py mid > $O/diag_fold_mid.log 2>&1; rc=$?; echo "fold mid rc $rc"; tail -3 $O/diag_fold_mid.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -v --timeout 300 -k "gated or bitwise or split or all_operators or graph" > $O/pytest_gated.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest_gated.log
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
for nz in 90 30 10 4; do
for v in "serial CEED_MI355X_ASSEMBLE=serial" "folded X=1"; do
  set -- $v; n=$1; shift
  env "$@" timeout -k 10 200 python3 $R/bench.py --nz $nz --steps 50 --warmup 5 --no-cpu-baseline > $O/exp12_${n}_$nz.json 2> $O/exp12_${n}_$nz.err
  echo "nz=$nz $n: $(tail -1 $O/exp12_${n}_$nz.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4), "kernel_us", round(d["roofline"]["kernel_avg_us"],1), "GDoF/s", round(d["value"]/1e3,2))')"
done; done
rm -rf /tmp/kt_g; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_g -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/exp12_trace.json 2> $O/exp12_trace.err
f=$(find /tmp/kt_g -name "*kernel_stats.csv" | head -1); grep -E 'k_assemble|k_fused_pencil<5, 5, 6' $f | awk -F, '{print $1, $(NF-4)/1000}' | sed -e 's/cps:://g; s/(.*)//' | tr '\n' ' '
