#!/bin/bash
# round-2 baseline: GPU tests, default bench line, per-dispatch kernel trace of a 50-step run (VERDICT weak #3)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline > $O/bench0.json 2> $O/bench0.err; tail -c 600 $O/bench0.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktrace -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/ktrace.log 2>&1
cp $(find /tmp/ktrace -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
cp $(find /tmp/ktrace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
head -5 $O/kernel_stats.csv
