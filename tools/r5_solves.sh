#!/bin/bash
# Round-5 solver records on ONE box: config 3 (pinned load), the 44 928-hex reference cylinder and the 99 000-hex hyperFS cylinder
# with the V-cycle's form chosen by measurement (--auto) against the four explicit forms; then the kernel shares of the chosen form.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
one() { tag=$1; shift; timeout -k 10 $TMO python3 $R/examples/solve_config3.py "$@" > $O/solve_$tag.json 2> $O/solve_$tag.err; echo "$tag rc=$? $(python3 -c "import json; d=json.load(open('$O/solve_$tag.json')); print(d['snes_its'], d['ksp_its'], round(d['snes_solve_s'],3), d['converged'], d['vcycle_graph'], d['fused_epilogue'], d.get('vcycle_tuning'))")"; }
TMO=120 one c3_auto --coarse amg --auto
TMO=120 one c3_graph_fused --coarse amg --graph
TMO=120 one c3_graph_two_pass --coarse amg --graph --no-fuse
BIG="--coarse amg --cylinder 10,110,90 --problem hyperFS --translate 0,-0.02,0.05"
TMO=600 one cyl99000_auto $BIG --auto
TMO=600 one cyl99000_graph_two_pass $BIG --graph --no-fuse
TMO=600 one cyl99000_eager_fused $BIG
TMO=600 one cyl99000_eager_two_pass $BIG --no-fuse
TMO=600 one cyl99000_auto_again $BIG --auto
TMO=400 one cyl44928_auto --coarse amg --auto --mesh $R/tests/golden/mesh_cylinder8_44928e_2ss_us.npz --translate 0,-0.02,0.05
if [ "$1" = stats ]; then
  rm -rf /tmp/solve_stats
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/solve_stats -- python3 $R/examples/solve_config3.py $BIG --auto > $O/solve99k_stats_run.log 2>&1
  cp $(find /tmp/solve_stats -name "*kernel_stats.csv" | head -1) $O/solve99k_kernel_stats.csv; head -14 $O/solve99k_kernel_stats.csv | cut -c1-160
fi
