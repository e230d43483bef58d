#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
M=tests/golden/mesh_cylinder8_44928e_2ss_us.npz
timeout -k 10 300 python -u examples/solve_config3.py --mesh $M --degree 2 --coarse assembled --graph --increments 2 --verbose 2>/dev/null | grep -v "      ksp" | head -30 | cut -c1-160
echo ===
timeout -k 10 300 python -u examples/solve_config3.py --mesh $M --degree 2 --coarse assembled --graph --increments 10 --translate 0,-0.02,0.05 2>/dev/null | tail -1 | cut -c1-700
