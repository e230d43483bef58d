#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export CEED_MI355X_ASSEMBLE=serial
for tag in config5 p6box32; do
  rm -rf /tmp/os_$tag
  if [ $tag = config5 ]; then A="--workload box --degree 6 --nr 64 --nth 64 --nz 64"; else A="--workload box --degree 6 --nr 32 --nth 32 --nz 32"; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/os_$tag -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline $A > /tmp/os_$tag.log 2>&1
  echo "== $tag serial"
  python3 - "$(find /tmp/os_$tag -name '*kernel_stats.csv' | head -1)" /tmp/os_$tag.log <<'PY'
import csv, json, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_fused_pencil" in r["Name"] or "k_assemble" in r["Name"]:
        print("  %-72s calls %3s  avg %8.1f us" % (r["Name"].split("(")[0][:72], r["Calls"], float(r["AverageNs"]) / 1000))
for l in open(sys.argv[2]):
    if l.startswith('{"metric"'):
        d = json.loads(l); print("  bench: %.2f GDoF/s, %.4f ms per apply" % (d["value"] / 1000, d["ms_per_step"]))
PY
done
