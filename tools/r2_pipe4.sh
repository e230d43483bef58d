#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd /tmp
run() { python3 $R/bench.py --no-cpu-baseline 2>$O/pipe4.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"; }
for rep in 1 2 3; do
run serial
export CEED_MI355X_ASSEMBLE=pipelined CEED_MI355X_PIPE_CHAINS=1
for cfg in "3 3" "3 4" "3 5" "3 2"; do set -- $cfg; CEED_MI355X_PIPE_SEGMENTS=$1 CEED_MI355X_PIPE_LAST=$2 run "chains S=$1 last=$2"; done
unset CEED_MI355X_ASSEMBLE CEED_MI355X_PIPE_CHAINS
done
