#!/bin/bash
# rocprofv3 kernel time of the fused + assemble kernels for the other models / shapes (GPU box)
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  tag=$1; shift; rm -rf /tmp/os_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/os_$tag -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" > /tmp/os_$tag.log 2>&1
  echo "== $tag: bench.py --steps 30 --warmup 3 --no-cpu-baseline $*"
  python3 - "$(find /tmp/os_$tag -name '*kernel_stats.csv' | head -1)" /tmp/os_$tag.log <<'PY'
import csv, json, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_fused_pencil" in r["Name"] or "k_assemble" in r["Name"]:
        print("  %-72s calls %3s  avg %8.1f us" % (r["Name"].split("(")[0][:72], r["Calls"], float(r["AverageNs"]) / 1000))
for l in open(sys.argv[2]):
    if l.startswith('{"metric"'):
        d = json.loads(l); print("  bench: %.2f GDoF/s, %.4f ms per apply" % (d["value"] / 1000, d["ms_per_step"]))
PY
}
run hyperSS --problem hyperSS
run linElas --problem linElas
run p6box32 --workload box --degree 6 --nr 32 --nth 32 --nz 32
run p2box96 --workload box --degree 2 --nr 96 --nth 96 --nz 96
run config5_whole --workload box --degree 6 --nr 64 --nth 64 --nz 64
run p1box128 --workload box --degree 1 --nr 128 --nth 128 --nz 128
run p3box64 --workload box --degree 3 --nr 64 --nth 64 --nz 64
