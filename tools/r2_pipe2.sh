#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd /tmp
run() { python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_us'])"; }
run serial
for S in 2 3 4 6; do CEED_MI355X_ASSEMBLE=pipelined CEED_MI355X_PIPE_CHAINS=1 CEED_MI355X_PIPE_SEGMENTS=$S run "chains S=$S"; done
for B in 256 1024; do CEED_MI355X_ASSEMBLE=pipelined CEED_MI355X_PIPE_CHAINS=1 CEED_MI355X_PIPE_SEGMENTS=2 CEED_MI355X_PIPE_BLOCKS=$B run "chains S=2 blocks=$B"; done
run serial
