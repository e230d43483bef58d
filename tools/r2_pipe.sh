#!/bin/bash
# pipelined assembly: bitwise test, then same-box timings against the serial default
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 400 -k "bitwise and pipelined" > $O/pipe_tests.log 2>&1 || { tail -20 $O/pipe_tests.log; exit 1; }
tail -1 $O/pipe_tests.log
cd /tmp
run() { python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_avg_us'])"; }
run serial
for S in 2 3 4 6 8; do CEED_MI355X_ASSEMBLE=pipelined CEED_MI355X_PIPE_SEGMENTS=$S run "pipelined S=$S"; done
for B in 256 512 1024; do CEED_MI355X_ASSEMBLE=pipelined CEED_MI355X_PIPE_SEGMENTS=4 CEED_MI355X_PIPE_BLOCKS=$B run "pipelined S=4 blocks=$B"; done
run serial
