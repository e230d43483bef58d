#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R && timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -v --timeout 200 -k "rccl" > $O/pytest_rccl.log 2>&1; rc=$?; echo "pytest rccl rc $rc"; tail -12 $O/pytest_rccl.log | cut -c1-220
export HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --prewarm-ms 20 --no-cpu-baseline > $O/bench_2rank_strong_gloo.json 2> $O/bench_2rank_strong_gloo.err; echo "2-rank strong cyl rc $?"; tail -c 900 $O/bench_2rank_strong_gloo.json
BENCH_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --workload box --nr 16 --nth 16 --nz 16 --degree 6 --steps 10 --warmup 2 --prewarm-ms 20 --no-cpu-baseline > $O/bench_4rank_box_gloo.json 2> $O/bench_4rank_box_gloo.err; echo "4-rank box rc $?"; tail -c 900 $O/bench_4rank_box_gloo.json; tail -3 $O/bench_4rank_box_gloo.err
