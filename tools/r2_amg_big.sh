#!/bin/bash
# the reference's larger unstructured cylinder (44 928 hexes): full solves with the Chebyshev and the aggregation coarse solver
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
M=tests/golden/mesh_cylinder8_44928e_2ss_us.npz
for deg in 2 4; do for c in assembled amg; do
  timeout -k 10 500 python -u examples/solve_config3.py --mesh $M --degree $deg --coarse $c --graph --increments 2 --verbose 2> $O/big_${deg}_$c.err | grep "^{\|^AggregationAMG" > $O/big_${deg}_$c.out || { tail -5 $O/big_${deg}_$c.err; exit 1; }
  python - <<PY
import json
lines = open("$O/big_${deg}_$c.out").read().strip().splitlines()
for l in lines:
    if l.startswith("AggregationAMG"): print("  ", l[:300])
d = json.loads(lines[-1])
print("degree $deg", {k: d[k] for k in ("coarse_solver", "converged", "snes_its", "ksp_its", "global_dofs_per_level", "setup_s", "snes_solve_s")})
PY
done; done
