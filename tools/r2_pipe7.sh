#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
run() { tag=$1; shift; python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag: %.2f GDoF/s %.4f ms | %s' % (d['value']/1000, d['ms_per_step'], d['config']['assembly']))"; }
for S in 3 4 5 6 8; do export CEED_MI355X_PIPE_SEGMENTS=$S
run "cyl2x S=$S" --nz 180
run "config5 S=$S" --workload box --degree 6 --nr 64 --nth 64 --nz 64
run "linElas S=$S" --problem linElas
done
export CEED_MI355X_PIPE_SEGMENTS=3
for L in 2 3 6; do CEED_MI355X_PIPE_LAST=$L run "linElas S=3 last=$L" --problem linElas; CEED_MI355X_PIPE_LAST=$L run "config5 S=3 last=$L" --workload box --degree 6 --nr 64 --nth 64 --nz 64; done
