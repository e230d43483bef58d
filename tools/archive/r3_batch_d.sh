#!/bin/bash
# Round-3 GPU batch D: N > 1 flow of bench.py rehearsed over gloo on the one GPU (2 ranks cylinder, 4 ranks box).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 $R/bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "rc=$?"; tail -c 1500 $O/bench_2rank_gloo.json; echo
BENCH_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29522 $R/bench.py --gpus 4 --steps 10 --warmup 2 --workload box --degree 6 --nr 32 --nth 32 --nz 32 > $O/bench_4rank_box_gloo.json 2> $O/bench_4rank_box_gloo.err; echo "rc=$?"; tail -c 1200 $O/bench_4rank_box_gloo.json; echo
tail -3 $O/bench_2rank_gloo.err
