#!/bin/bash
# A/B the two fused kernels over shapes: prints GDoF/s and ms per apply
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["config"]["kernel"], round(d["value"]/1000,2), "GDoF/s", round(d["ms_per_step"],4), "ms")'
for v in rows pencil; do
  CEED_MI355X_FUSED=$v python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" "cyl-p4-hyperFS"
  CEED_MI355X_FUSED=$v python bench.py --steps 50 --warmup 5 --no-cpu-baseline --problem hyperSS 2>/dev/null | python -c "$P" "cyl-p4-hyperSS"
  CEED_MI355X_FUSED=$v python bench.py --steps 50 --warmup 5 --no-cpu-baseline --problem linElas 2>/dev/null | python -c "$P" "cyl-p4-linElas"
  CEED_MI355X_FUSED=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload box --degree 6 --nr 32 --nth 32 --nz 32 2>/dev/null | python -c "$P" "box32-p6-hyperFS"
  CEED_MI355X_FUSED=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload box --degree 5 --nr 36 --nth 36 --nz 36 2>/dev/null | python -c "$P" "box36-p5-hyperFS"
  CEED_MI355X_FUSED=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload box --degree 3 --nr 64 --nth 64 --nz 64 2>/dev/null | python -c "$P" "box64-p3-hyperFS"
  CEED_MI355X_FUSED=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline --workload box --degree 2 --nr 96 --nth 96 --nz 96 2>/dev/null | python -c "$P" "box96-p2-hyperFS"
done
