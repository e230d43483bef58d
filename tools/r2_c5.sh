#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
run() { tag=$1; shift; python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag: %.2f GDoF/s %.4f ms kernel %.1f us | %s' % (d['value']/1000, d['ms_per_step'], d['roofline']['kernel_avg_us'], d['config']['assembly'][:30]))"; }
for rep in 1 2; do
CEED_MI355X_ASSEMBLE=serial run "config5 serial" --workload box --degree 6 --nr 64 --nth 64 --nz 64
run "config5 default" --workload box --degree 6 --nr 64 --nth 64 --nz 64
done
CEED_MI355X_ASSEMBLE=serial run "p6box32 serial" --workload box --degree 6 --nr 32 --nth 32 --nz 32
run "p6box32 default" --workload box --degree 6 --nr 32 --nth 32 --nz 32
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
