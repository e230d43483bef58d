#!/bin/bash
# round-2 experiment 1: prewarm, dynamic schedule, ungated concurrent assembly (timing only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; env "$@" python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/exp1_$n.json 2> $O/exp1_$n.err || { echo "$n FAILED"; tail -3 $O/exp1_$n.err; }; python3 -c "
import json,sys
d=json.loads(open('$O/exp1_$n.json').read()); print('$n', round(d['ms_per_step'],4), round(d['value']), round(d['roofline']['kernel_avg_us'],1))"; }

run base X=1
run dyn CEED_MI355X_SCHED=dynamic
run ovl CEED_MI355X_ASM_OVERLAP=1
run dynovl CEED_MI355X_SCHED=dynamic CEED_MI355X_ASM_OVERLAP=1
run base2 X=1
run dyn2 CEED_MI355X_SCHED=dynamic
# kernel trace of the dynamic + overlapped variant
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/kt2.log 2>&1
cp $(find /tmp/kt2 -name "*kernel_trace.csv" | head -1) $O/kernel_trace_dynovl.csv
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_prewarm.csv
head -4 $O/kernel_stats_prewarm.csv
cd $R && CEED_MI355X_SCHED=dynamic timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_dyn.log 2>&1; echo "pytest dyn rc $?"; tail -3 $O/pytest_dyn.log
