#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 > $O/pipe_tests5.log 2>&1 || { tail -30 $O/pipe_tests5.log; exit 1; }
tail -1 $O/pipe_tests5.log
cd /tmp
run() { python3 $R/bench.py --no-cpu-baseline 2>$O/pipe5.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['config']['assembly'])"; }
run default
CEED_MI355X_ASSEMBLE=serial run serial
CEED_MI355X_PIPE_BLOCKS=512 run "blocks 512"
CEED_MI355X_PIPE_LAST=5 run "last 5"
python3 $R/bench.py --workload mesh --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mesh', round(d['value']), round(d['ms_per_step'],4), d['config']['assembly'])"
CEED_MI355X_ASSEMBLE=serial python3 $R/bench.py --workload mesh --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mesh serial', round(d['value']), round(d['ms_per_step'],4), d['config']['assembly'])"
run default
CEED_MI355X_ASSEMBLE=serial run serial
