#!/bin/bash
# Refresh profiles/r02_* on the GPU box: default bench line, rocprofv3 kernel stats + per-dispatch series (with and without
# the pre-warm), FETCH_SIZE / WRITE_SIZE PMC passes (separate, with the axpby calibration launch) and their reduction, the
# three SQ counter passes over the fused kernel, the unstructured-mesh workload.  Outputs land in gpurun_out/profiles_new/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_new; mkdir -p $O; T=${1:-r02}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${T}_bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
echo "bench done: $(python3 -c "import json; d=json.loads(open('$O/${T}_bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])")"
python3 $R/bench.py --workload mesh --no-cpu-baseline > $O/${T}_bench_unstructured_mesh.json 2>> $O/bench.err
python3 $R/bench.py --nz 41 --no-cpu-baseline > $O/${T}_bench_structured_45100e.json 2>> $O/bench.err
for pw in 150 0; do
  rm -rf /tmp/prof_stats_$pw
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$pw -- python3 $R/bench.py --steps 50 --warmup 5 --prewarm-ms $pw --no-cpu-baseline > $O/stats_run_$pw.log 2>&1 || { echo stats failed; tail -5 $O/stats_run_$pw.log; exit 1; }
done
cp $(find /tmp/prof_stats_150 -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
cp $(find /tmp/prof_stats_0 -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_no_prewarm.csv
python3 - <<PY
import csv, glob
out = open("$O/${T}_dispatch_series.txt", "w")
for pw in (0, 150):
    f = glob.glob(f"/tmp/prof_stats_{pw}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_fused_pencil<5, 5, 6" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    out.write(f"# k_fused_pencil<5,5,HyperFSdF,geo,eo> per dispatch, bench.py --steps 50 --warmup 5 --prewarm-ms {pw}: index, start (us after the first), duration (us)\n")
    for i, r in enumerate(rows):
        out.write(f"{i}\t{(int(r['Start_Timestamp']) - t0) / 1e3:.1f}\t{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}\n")
    last = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[-50:]]
    m = sum(last) / len(last); sd = (sum((x - m) ** 2 for x in last) / len(last)) ** 0.5
    out.write(f"# last 50 dispatches (the timed ones): mean {m:.1f} us, sigma {sd:.1f} us = {100 * sd / m:.1f} %\n\n")
out.close()
PY
tail -2 $O/${T}_dispatch_series.txt
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python3 $R/bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --calibrate-traffic > $O/pmc_$C.log 2>&1 || { echo pmc $C failed; tail -5 $O/pmc_$C.log; exit 1; }
  cp $(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1) $O/${T}_pmc_$C.csv
  echo "pmc $C done"
done
cd $R && python3 tools/collect_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/${T}_bench.json $T $COMMIT > $O/traffic.log 2>&1; tail -3 $O/traffic.log
cp gpurun_out/${T}_traffic.json $O/${T}_traffic.json 2>/dev/null

bash tools/pmc_fused.sh $T > $O/${T}_pmc_pencil_summary.txt 2>&1; cp gpurun_out/pmc_${T}_pass*.csv $O/ 2>/dev/null
head -4 $O/${T}_kernel_stats.csv
