#!/bin/bash
# Round-5 solver records on ONE box: config 3 (pinned load) and the 99 000-hex hyperFS cylinder, the apply's consumers fused
# (default) against the two-pass form (--no-fuse), default / variant / default-again; then the kernel shares of the fused solve.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
one() { tag=$1; shift; timeout -k 10 $TMO python3 $R/examples/solve_config3.py "$@" > $O/solve_$tag.json 2> $O/solve_$tag.err; echo "$tag rc=$? $(python3 -c "import json; d=json.load(open('$O/solve_$tag.json')); print(d['snes_its'], d['ksp_its'], round(d['snes_solve_s'],3), d['converged'])")"; }
TMO=120 one c3_fused --coarse amg --graph
TMO=120 one c3_nofuse --coarse amg --graph --no-fuse
TMO=120 one c3_fused_again --coarse amg --graph
TMO=600 one cyl99000_fused --coarse amg --graph --cylinder 10,110,90 --problem hyperFS --translate 0,-0.02,0.05
TMO=600 one cyl99000_nofuse --coarse amg --graph --cylinder 10,110,90 --problem hyperFS --translate 0,-0.02,0.05 --no-fuse
TMO=600 one cyl99000_fused_again --coarse amg --graph --cylinder 10,110,90 --problem hyperFS --translate 0,-0.02,0.05
if [ "$1" = stats ]; then
  rm -rf /tmp/solve_stats
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/solve_stats -- python3 $R/examples/solve_config3.py --coarse amg --graph --cylinder 10,110,90 --problem hyperFS --translate 0,-0.02,0.05 > $O/solve99k_stats_run.log 2>&1
  cp $(find /tmp/solve_stats -name "*kernel_stats.csv" | head -1) $O/solve99k_kernel_stats.csv; head -14 $O/solve99k_kernel_stats.csv | cut -c1-160
fi
