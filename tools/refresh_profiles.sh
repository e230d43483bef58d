#!/bin/bash
# Refresh profiles/r01_* on the GPU box: default bench line, rocprofv3 kernel stats, FETCH/WRITE PMC passes (separate, with
# the axpby calibration launch) and their reduction.  Outputs land in gpurun_out/profiles_new/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_new; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/r01_bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/stats_run.log 2>&1 || { echo stats failed; tail -5 $O/stats_run.log; exit 1; }
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/r01_kernel_stats.csv
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --calibrate-traffic > $O/pmc_$C.log 2>&1 || { echo pmc $C failed; tail -5 $O/pmc_$C.log; exit 1; }
  cp $(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1) $O/r01_pmc_$C.csv
  echo "pmc $C done"
done
cd $R && python3 tools/collect_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/r01_bench.json > $O/traffic.log 2>&1; tail -3 $O/traffic.log
cp gpurun_out/r01_traffic.json $O/r01_traffic.json 2>/dev/null
head -12 $O/r01_kernel_stats.csv
