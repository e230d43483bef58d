#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
M=tests/golden/mesh_cylinder8_44928e_2ss_us.npz
for deg in 2 4; do for c in amg assembled; do
  timeout -k 10 500 python -u examples/solve_config3.py --mesh $M --degree $deg --coarse $c --graph --increments 10 --translate 0,-0.02,0.05 2> $O/big_${deg}_$c.err | tail -1 > $O/big_${deg}_$c.json || { tail -5 $O/big_${deg}_$c.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/big_${deg}_$c.json").read())
print("degree $deg", {k: d[k] for k in ("coarse_solver", "converged", "snes_its", "ksp_its", "global_dofs_per_level", "setup_s", "snes_solve_s")})
PY
done; done
