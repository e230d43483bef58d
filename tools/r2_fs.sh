#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
for c in assembled amg; do
  timeout -k 10 300 python -u examples/solve_config3.py --problem hyperFS --coarse $c --graph > $O/config3fs_$c.json 2> $O/config3fs_$c.err || { tail -5 $O/config3fs_$c.err; exit 1; }
  python - <<PY
import json; d = json.loads(open("$O/config3fs_$c.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("problem", "coarse_solver", "converged", "snes_its", "ksp_its", "snes_solve_s", "max_abs_displacement")})
PY
done
timeout -k 10 300 python -u examples/solve_config3.py --problem linElas --coarse amg --graph --increments 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('problem','coarse_solver','converged','snes_its','ksp_its','snes_solve_s')})"
