#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
for v in "40 100" "30 100" "20 100" "20 50" "15 30" "10 30" "10 15" "6 10"; do set -- $v
  python examples/solve_config3.py --coarse assembled --graph --coarse-cheb-its $1 --coarse-cheb-ratio $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cheb its $1 ratio $2: solve', round(d['snes_solve_s'],3), 's  newton', d['snes_its'], 'ksp', d['ksp_its'], 'converged', d['converged'])"
done
