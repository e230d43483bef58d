#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd /tmp
run() { python3 $R/bench.py --no-cpu-baseline 2>$O/pipe3.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_us'])"; grep -h "segment" $O/pipe3.err | head -6; }
run serial
export CEED_MI355X_ASSEMBLE=pipelined CEED_MI355X_PIPE_CHAINS=1
CEED_MI355X_PIPE_DEBUG=1 CEED_MI355X_PIPE_SEGMENTS=3 CEED_MI355X_PIPE_LAST=3 run "chains S=3 last=3"
for cfg in "2 2" "2 4" "3 2" "3 4" "3 6" "4 3" "3 0"; do set -- $cfg; CEED_MI355X_PIPE_SEGMENTS=$1 CEED_MI355X_PIPE_LAST=$2 run "chains S=$1 last=$2"; done
unset CEED_MI355X_ASSEMBLE CEED_MI355X_PIPE_CHAINS
run serial
