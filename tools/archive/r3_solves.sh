#!/bin/bash
# Round-3 solver records: config 3 at the README's load, config 3 at the pinned load (counts must not
# move with the pattern-only Galerkin products), and the big cylinders with the aggregation hierarchy (item 6).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
one() { tag=$1; shift; timeout -k 10 $TMO python3 $R/examples/solve_config3.py "$@" > $O/solve_$tag.json 2> $O/solve_$tag.err; echo "$tag rc=$?"; tail -c 900 $O/solve_$tag.json; echo; }
TMO=120 one c3_pinned_amg --coarse amg --graph
TMO=200 one c3_readme_load_hyperSS --coarse amg --graph --translate 0,-0.5,1
TMO=200 one c3_readme_load_hyperFS --coarse amg --graph --translate 0,-0.5,1 --problem hyperFS
TMO=200 one c3_readme_load_hyperSS_40inc --coarse amg --graph --translate 0,-0.5,1 --increments 40
TMO=200 one c3_readme_load_hyperFS_40inc --coarse amg --graph --translate 0,-0.5,1 --problem hyperFS --increments 40
TMO=400 one cyl44928_p4_amg --coarse amg --graph --mesh $R/tests/golden/mesh_cylinder8_44928e_2ss_us.npz --translate 0,-0.02,0.05
TMO=600 one cyl99000_p4_hyperFS_amg --coarse amg --graph --cylinder 10,110,90 --problem hyperFS --translate 0,-0.02,0.05
