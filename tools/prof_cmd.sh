#!/bin/bash
# Per-kernel profile of ONE python command on the GPU box: rocprofv3 kernel stats, then FETCH_SIZE and WRITE_SIZE in separate
# PMC passes (MI355X_MICROARCH.md: never with a trace domain beside them), reduced to bytes per launch per kernel.
#   usage: prof_cmd.sh <tag> <kernel name patterns, '|'-separated> python3 <script> [args ...]
# gfx950 corrections (guide, and calibrated on a known axpby in rounds 1-4: 2.000 / 1.000): FETCH_SIZE counts 64 B per 128-B
# request -> x2; WRITE_SIZE as counted; both in KiB.  Outputs: gpurun_out/prof/<tag>_kernel_stats.csv, <tag>_traffic.json.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O; T=$1; export KPAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_stats /tmp/pc_FETCH_SIZE /tmp/pc_WRITE_SIZE
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_stats -- "$@" > $O/${T}_stats_run.log 2>&1 || { echo stats failed; tail -5 $O/${T}_stats_run.log; exit 1; }
cp $(find /tmp/pc_stats -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d /tmp/pc_$C -- "$@" > $O/${T}_pmc_$C.log 2>&1 || { echo pmc $C failed; tail -5 $O/${T}_pmc_$C.log; exit 1; }
  cp $(find /tmp/pc_$C -name "*counter_collection.csv" | head -1) $O/${T}_pmc_$C.csv
done
python3 - <<PY
import csv, json, os, collections
pats = os.environ["KPAT"].split("|")
def load(f, ctr):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr: agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
fe, wr = load("$O/${T}_pmc_FETCH_SIZE.csv", "FETCH_SIZE"), load("$O/${T}_pmc_WRITE_SIZE.csv", "WRITE_SIZE")
stats = {r["Name"]: r for r in csv.DictReader(open("$O/${T}_kernel_stats.csv"))}
out = {"command": "$*", "corrections": {"FETCH_SIZE": "KiB x 2 (64 B counted per 128-B request)", "WRITE_SIZE": "KiB x 1"}, "kernels": {}}
for k in sorted(fe):
    if not any(p in k for p in pats): continue
    f = sum(fe[k]) / len(fe[k]) * 1024 * 2.0; w = sum(wr[k]) / len(wr[k]) * 1024 if k in wr else None
    st = stats.get(k, {})
    us = float(st["AverageNs"]) / 1e3 if st else None
    out["kernels"][k] = {"launches": len(fe[k]), "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "avg_us": us,
                         "hbm_GBs": (f + (w or 0)) / us / 1e3 if us else None}
json.dump(out, open("$O/${T}_traffic.json", "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-60s %4d launches  fetch %8.1f MB  write %8.1f MB  %8.1f us  %6.0f GB/s" % (k[:60], v["launches"], v["fetch_bytes_per_launch"] / 1e6, (v["write_bytes_per_launch"] or 0) / 1e6, v["avg_us"] or 0, v["hbm_GBs"] or 0))
PY
