#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 400 -k "bitwise" > $O/pipe_tests10.log 2>&1 || { tail -30 $O/pipe_tests10.log; exit 1; }
tail -1 $O/pipe_tests10.log
cd /tmp
run() { tag=$1; shift; python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag: %.2f GDoF/s %.4f ms | %s' % (d['value']/1000, d['ms_per_step'], d['config']['assembly'][:24]))"; }
run "p2box96" --workload box --degree 2 --nr 96 --nth 96 --nz 96
run "p3box64" --workload box --degree 3 --nr 64 --nth 64 --nz 64
run "p6box32" --workload box --degree 6 --nr 32 --nth 32 --nz 32
run "config4"
