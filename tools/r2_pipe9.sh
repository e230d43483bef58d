#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
run() { tag=$1; shift; python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag: %.2f GDoF/s %.4f ms | %s' % (d['value']/1000, d['ms_per_step'], d['config']['assembly']))"; }
run "config4 auto"
run "cyl2x auto" --nz 180
run "config5 auto" --workload box --degree 6 --nr 64 --nth 64 --nz 64
run "p2box96 auto" --workload box --degree 2 --nr 96 --nth 96 --nz 96
run "p3box64 auto" --workload box --degree 3 --nr 64 --nth 64 --nz 64
run "p1box128 auto" --workload box --degree 1 --nr 128 --nth 128 --nz 128
CEED_MI355X_ASSEMBLE=serial run "p1box128 serial" --workload box --degree 1 --nr 128 --nth 128 --nz 128
run "linElas auto" --problem linElas
