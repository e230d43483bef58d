#!/bin/bash
# round-2 experiment 2: gated assembly -- correctness first, then timing against the serial assembly
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R && timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -v --timeout 300 -k "gated or bitwise or split or all_operators or graph" > $O/pytest_gated.log 2>&1; rc=$?; echo "pytest gated rc $rc"; tail -15 $O/pytest_gated.log
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; env "$@" timeout -k 10 200 python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/exp2_$n.json 2> $O/exp2_$n.err || { echo "$n FAILED"; tail -3 $O/exp2_$n.err; return; }; python3 -c "
import json,sys
d=json.loads(open('$O/exp2_$n.json').read()); print('$n', round(d['ms_per_step'],4), round(d['value']), round(d['roofline']['kernel_avg_us'],1))"; }
run serial CEED_MI355X_ASSEMBLE=serial
run gated X=1
run gated2 CEED_MI355X_ASM_WAVES=2
run gated8 CEED_MI355X_ASM_WAVES=8
run serial2 CEED_MI355X_ASSEMBLE=serial
run gated_b X=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/kt3.log 2>&1
cp $(find /tmp/kt3 -name "*kernel_trace.csv" | head -1) $O/kernel_trace_gated.csv
cp $(find /tmp/kt3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_gated.csv
head -6 $O/kernel_stats_gated.csv
