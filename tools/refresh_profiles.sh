#!/bin/bash
# Refresh a set of committed profiles on the GPU box: bench line, rocprofv3 kernel stats + per-apply spans (with and
# without the pre-warm), FETCH_SIZE / WRITE_SIZE PMC passes (separate, with the axpby calibration launch) and their
# reduction, the three SQ counter passes over the fused kernel.  Outputs land in gpurun_out/profiles_new/.
#   usage: refresh_profiles.sh <tag> <commit> <fused-kernel pattern in the trace, e.g. "k_fused_pencil<5, 5, 6"> [bench.py args ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/profiles_new; mkdir -p $O; T=${1:-r03}; COMMIT=${2:-unknown}; KPAT=${3:-"k_fused_pencil<5, 5, 6"}; shift 3
export KPAT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $O/${T}_bench.json 2> $O/${T}_bench.err || { echo bench failed; tail -5 $O/${T}_bench.err; exit 1; }
echo "bench done: $(python3 -c "import json; d=json.loads(open('$O/${T}_bench.json').read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_cold'), d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('value'), (d['roofline'].get('valu_issue') or {}).get('valu_issue_frac'))")"
for pw in 150 0; do
  rm -rf /tmp/prof_stats_$pw
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$pw -- python3 $R/bench.py "$@" --steps 50 --warmup 5 --blocks 1 --no-clock-probe --prewarm-ms $pw --cold-idle-s 0 --no-cpu-baseline > $O/${T}_stats_run_$pw.log 2>&1 || { echo stats failed; tail -5 $O/${T}_stats_run_$pw.log; exit 1; }
done
rm -rf /tmp/prof_stats_serial
CEED_MI355X_ASSEMBLE=serial timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_serial -- python3 $R/bench.py "$@" --steps 50 --warmup 5 --blocks 1 --no-clock-probe --cold-idle-s 0 --no-cpu-baseline > $O/${T}_stats_run_serial.log 2>&1 || { echo serial stats failed; tail -5 $O/${T}_stats_run_serial.log; exit 1; }
grep '^{"metric"' $O/${T}_stats_run_serial.log > $O/${T}_bench_serial_form.json
cp $(find /tmp/prof_stats_serial -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_serial_form.csv
cp $(find /tmp/prof_stats_150 -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
cp $(find /tmp/prof_stats_0 -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_no_prewarm.csv
python3 - <<PY
import csv, glob, os
KP = os.environ["KPAT"]
# one APPLY = the fused launches and the k_assemble launches between two joins: its time is the span from the first
# kernel's start to the last kernel's end (the pipelined form overlaps its kernels: per-kernel durations do not add up)
out = open("$O/${T}_dispatch_series.txt", "w")
for pw in (0, 150):
    f = glob.glob(f"/tmp/prof_stats_{pw}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if KP in r["Kernel_Name"] or "k_assemble" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = next(i for i, r in enumerate(rows) if KP in r["Kernel_Name"])
    rows = rows[first:]
    applies, cur, cur_end, seen_asm = [], [], 0, False
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if cur and "k_fused" in r["Kernel_Name"] and seen_asm and s >= cur_end:
            applies.append(cur); cur, seen_asm, cur_end = [], False, 0
        cur.append(r); cur_end = max(cur_end, e); seen_asm = seen_asm or "k_assemble" in r["Kernel_Name"]
    if cur: applies.append(cur)
    t0 = int(applies[0][0]["Start_Timestamp"])
    out.write(f"# one line per CeedOperatorApply of bench.py --steps 50 --warmup 5 --prewarm-ms {pw}: index, start (us after the first), span first kernel start .. last kernel end (us), kernels in it, sum of the fused kernels' own durations (us)\n")
    spans = []
    for i, a in enumerate(applies):
        s0 = min(int(r["Start_Timestamp"]) for r in a); e1 = max(int(r["End_Timestamp"]) for r in a)
        fsum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in a if "k_fused" in r["Kernel_Name"])
        spans.append((e1 - s0) / 1e3)
        out.write(f"{i}\t{(s0 - t0) / 1e3:.1f}\t{(e1 - s0) / 1e3:.1f}\t{len(a)}\t{fsum / 1e3:.1f}\n")
    last = spans[-50:]
    m = sum(last) / len(last); sd = (sum((x - m) ** 2 for x in last) / len(last)) ** 0.5
    out.write(f"# last 50 applies (the timed ones): mean span {m:.1f} us, sigma {sd:.1f} us = {100 * sd / m:.1f} %\n\n")
out.close()
PY
tail -2 $O/${T}_dispatch_series.txt
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python3 $R/bench.py "$@" --steps 5 --warmup 2 --blocks 1 --no-clock-probe --prewarm-ms 0 --cold-idle-s 0 --no-cpu-baseline --calibrate-traffic > $O/${T}_pmc_$C.log 2>&1 || { echo pmc $C failed; tail -5 $O/${T}_pmc_$C.log; exit 1; }
  cp $(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1) $O/${T}_pmc_$C.csv
  echo "pmc $C done"
done
cd $R && python3 tools/collect_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/${T}_bench.json $T $COMMIT > $O/${T}_traffic.log 2>&1; tail -3 $O/${T}_traffic.log
cp gpurun_out/${T}_traffic.json $O/${T}_traffic.json 2>/dev/null

BENCH_ARGS="$*" bash tools/pmc_fused.sh $T > $O/${T}_pmc_pencil_summary.txt 2>&1; cp gpurun_out/pmc_${T}_pass*.csv $O/ 2>/dev/null
head -4 $O/${T}_kernel_stats.csv
