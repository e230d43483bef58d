#!/bin/bash
# One-call split-phase apply (CeedXOperatorApplyWithHalo) on the emulated rank 3 of 8: launch-shape matrix, one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {  # tag workload-args env...
  tag=$1; shift; A=$1; shift
  env "$@" timeout -k 10 200 python3 $R/bench.py --emulate-rank 3 --of 8 $A --steps 100 --warmup 10 --no-cpu-baseline --cold-idle-s 0 > $O/m_$tag.json 2> $O/m_$tag.err || { echo "$tag failed"; tail -3 $O/m_$tag.err; return 1; }
  python3 - <<PY
import json
d=[json.loads(l) for l in open("$O/m_$tag.json") if l.startswith("{")][-1]; e=d["emulated_rank"]
print("%-28s %7.1f us/apply  %6.2f GDoF/s  exchange alone %.1f us" % ("$tag", e["us_per_apply_incl_exchange"], d["value"]/1e3, d["config"]["halo_exchange_us_alone"]))
PY
}
trace() {  # tag workload-args env...
  tag=$1; shift; A=$1; shift
  rm -rf /tmp/kt_$tag
  env "$@" timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -- python3 $R/bench.py --emulate-rank 3 --of 8 $A --steps 50 --warmup 5 --no-cpu-baseline --cold-idle-s 0 > $O/t_$tag.json 2> $O/t_$tag.err || { echo "$tag trace failed"; tail -3 $O/t_$tag.err; return 1; }
  python3 $R/tools/apply_timeline.py $(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1) --per-apply 2 --last 40 --json $O/t_${tag}_timeline.json > $O/t_${tag}_timeline.txt 2>&1
  cat $O/t_${tag}_timeline.txt
}
CYL=""; BOX="--workload box --degree 6 --nr 64 --nth 64 --nz 64"
run cyl_seq      "$CYL" CEED_MI355X_OVL_MODE=1 || exit 1
run cyl_g1_0     "$CYL" CEED_MI355X_OVL_G0=1 CEED_MI355X_OVL_G1=0
run cyl_g0_0     "$CYL" CEED_MI355X_OVL_G0=0 CEED_MI355X_OVL_G1=0
run cyl_g1_1     "$CYL" CEED_MI355X_OVL_G0=1 CEED_MI355X_OVL_G1=1
run cyl_g1_2     "$CYL" CEED_MI355X_OVL_G0=1 CEED_MI355X_OVL_G1=2
run cyl_g1_3     "$CYL" CEED_MI355X_OVL_G0=1 CEED_MI355X_OVL_G1=3
run cyl_nooverlap "$CYL --no-overlap"
run box_seq      "$BOX" CEED_MI355X_OVL_MODE=1
run box_g1_0     "$BOX" CEED_MI355X_OVL_G0=1 CEED_MI355X_OVL_G1=0
run box_g1_4     "$BOX" CEED_MI355X_OVL_G0=1 CEED_MI355X_OVL_G1=4
run box_nooverlap "$BOX --no-overlap"
trace cyl_g1_0   "$CYL" CEED_MI355X_OVL_G0=1 CEED_MI355X_OVL_G1=0
trace cyl_seq    "$CYL" CEED_MI355X_OVL_MODE=1
trace box_g1_0   "$BOX" CEED_MI355X_OVL_G0=1 CEED_MI355X_OVL_G1=0
