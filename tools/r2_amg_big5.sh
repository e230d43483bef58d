#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
M=tests/golden/mesh_cylinder8_44928e_2ss_us.npz
for deg in 2 4; do for cyc in 1 2 3; do
  [ $deg = 4 ] && [ $cyc = 3 ] && continue
  timeout -k 10 500 python -u examples/solve_config3.py --mesh $M --degree $deg --coarse amg --graph --increments 10 --translate 0,-0.02,0.05 --amg-coarse-cycles $cyc 2>/dev/null | tail -1 > $O/big_${deg}_amg_c$cyc.json
  python - <<PY
import json
d = json.loads(open("$O/big_${deg}_amg_c$cyc.json").read())
print("degree $deg cycles $cyc", {k: d[k] for k in ("converged", "snes_its", "ksp_its", "setup_s", "snes_solve_s")})
PY
done; done
timeout -k 10 300 python -u examples/solve_config3.py --coarse amg --graph 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config 3', {k: d[k] for k in ('converged','snes_its','ksp_its','setup_s','snes_solve_s')})"
