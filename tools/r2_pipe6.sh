#!/bin/bash
# pipelined (default) against serial for the other models / shapes, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd /tmp
run() { tag=$1; shift; for m in default serial; do
  if [ $m = serial ]; then export CEED_MI355X_ASSEMBLE=serial; else unset CEED_MI355X_ASSEMBLE; fi
  python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag $m: %.2f GDoF/s %.4f ms | %s' % (d['value']/1000, d['ms_per_step'], d['config']['assembly']))"; done; unset CEED_MI355X_ASSEMBLE; }
run config4
run mesh45k --workload mesh
run hyperSS --problem hyperSS
run linElas --problem linElas
run p6box32 --workload box --degree 6 --nr 32 --nth 32 --nz 32
run config5_whole --workload box --degree 6 --nr 64 --nth 64 --nz 64
run p2box96 --workload box --degree 2 --nr 96 --nth 96 --nz 96
run p3box64 --workload box --degree 3 --nr 64 --nth 64 --nz 64
run cyl2x --nz 180
