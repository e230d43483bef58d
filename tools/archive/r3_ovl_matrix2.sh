#!/bin/bash
# Forms of CeedXOperatorApplyWithHalo on the emulated rank 3 of 8: OVL_MODE (0 whole apply then exchange, 1 split on one
# stream, 2 split on two streams) x COMM_INLINE (RCCL group on the producing stream / on the communicator's stream).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {
  tag=$1; shift; A=$1; shift
  env "$@" timeout -k 10 200 python3 $R/bench.py --emulate-rank 3 --of 8 $A --steps 100 --warmup 10 --no-cpu-baseline --cold-idle-s 0 > $O/n_$tag.json 2> $O/n_$tag.err || { echo "$tag failed"; tail -3 $O/n_$tag.err; return 1; }
  python3 - <<PY
import json
d=[json.loads(l) for l in open("$O/n_$tag.json") if l.startswith("{")][-1]; e=d["emulated_rank"]
print("%-28s %7.1f us/apply  %6.2f GDoF/s  exchange alone %.1f us" % ("$tag", e["us_per_apply_incl_exchange"], d["value"]/1e3, d["config"]["halo_exchange_us_alone"]))
PY
}
trace() {
  tag=$1; shift; A=$1; shift
  rm -rf /tmp/kt_$tag
  env "$@" timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -- python3 $R/bench.py --emulate-rank 3 --of 8 $A --steps 50 --warmup 5 --no-cpu-baseline --cold-idle-s 0 > $O/t_$tag.json 2> $O/t_$tag.err || { echo "$tag trace failed"; tail -3 $O/t_$tag.err; return 1; }
  T=$(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/apply_timeline.py $T --per-apply $PER --last 40 --json $O/t_${tag}_timeline.json > $O/t_${tag}_timeline.txt 2>&1
  cat $O/t_${tag}_timeline.txt
  grep -m1 rccl $T | cut -d, -f8,12-19 > $O/t_${tag}_rccl_kernel_resources.txt; cat $O/t_${tag}_rccl_kernel_resources.txt
}
CYL=""; BOX="--workload box --degree 6 --nr 64 --nth 64 --nz 64"
for w in cyl box; do
  if [ $w = cyl ]; then A="$CYL"; else A="$BOX"; fi
  for m in 0 1 2; do for i in 1 0; do
    run ${w}_mode${m}_inline${i} "$A" CEED_MI355X_OVL_MODE=$m CEED_MI355X_COMM_INLINE=$i
  done; done
done
PER=1 trace cyl_mode0_inline1 "$CYL"
PER=2 trace cyl_mode2_inline1 "$CYL" CEED_MI355X_OVL_MODE=2
PER=1 trace box_mode0_inline1 "$BOX"
