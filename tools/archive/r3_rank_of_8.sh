#!/bin/bash
# Rank-local regime of the N = 8 strong-scaling job on ONE GPU: emulated rank 3 of 8, the library's
# default one-call apply + exchange, true-size RCCL self-exchange; bench line + kernel trace + per-apply timeline.
#   usage: r3_rank_of_8.sh <tag> [ENV=..]      -> gpurun_out/r3/<tag>_{cyl,box}.json, _timeline.{txt,json}, _rccl_kernel.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
tag=$1; shift
for w in cyl box; do
  if [ $w = cyl ]; then A="--emulate-rank 3 --of 8"; else A="--emulate-rank 3 --of 8 --workload box --degree 6 --nr 64 --nth 64 --nz 64"; fi
  env "$@" timeout -k 10 280 python3 $R/bench.py $A --steps 100 --warmup 10 --no-cpu-baseline > $O/${tag}_$w.json 2> $O/${tag}_$w.err || { echo "$w bench failed"; tail -5 $O/${tag}_$w.err; exit 1; }
  rm -rf /tmp/kt_$w
  env "$@" timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$w -- python3 $R/bench.py $A --steps 50 --warmup 5 --no-cpu-baseline --cold-idle-s 0 > $O/${tag}_${w}_traced.json 2> $O/${tag}_${w}_traced.err || { echo "$w trace failed"; tail -5 $O/${tag}_${w}_traced.err; exit 1; }
  T=$(find /tmp/kt_$w -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/apply_timeline.py $T --per-apply ${PER:-1} --last 40 --json $O/${tag}_${w}_timeline.json > $O/${tag}_${w}_timeline.txt 2>&1 || { echo "timeline failed"; tail -3 $O/${tag}_${w}_timeline.txt; }
  (head -1 $T | cut -d, -f8,12-19; grep -m1 rccl $T | cut -d, -f8,12-19) > $O/${tag}_${w}_rccl_kernel.txt
  python3 - <<PY
import json
d=[json.loads(l) for l in open("$O/${tag}_$w.json") if l.startswith("{")][-1]; e=d["emulated_rank"]
print("$tag $w: %.1f us per apply incl. exchange (cold %.1f), %.2f GDoF/s per rank, exchange alone %.1f us, neighbours %s" % (e["us_per_apply_incl_exchange"], 1e3*(d.get("ms_per_step_cold") or 0), d["value"]/1e3, d["config"]["halo_exchange_us_alone"], e["neighbour_dofs"]))
PY
  cat $O/${tag}_${w}_timeline.txt
done
