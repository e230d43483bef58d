#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
timeout -k 10 300 python -u -m pytest tests/test_amg.py -m gpu -x -q --timeout 250 -k "products or hierarchy" > $O/amg_tests2.log 2>&1 || { tail -5 $O/amg_tests2.log; exit 1; }
tail -1 $O/amg_tests2.log
for cfg in "3 10" "2 10" "2 5" "4 10" "3 20" "3 5"; do
  set -- $cfg
  timeout -k 10 200 python -u examples/solve_config3.py --coarse amg --graph --amg-smooth-its $1 --amg-smooth-ratio $2 > $O/sweep.json 2> $O/sweep.err || { tail -5 $O/sweep.err; exit 1; }
  python - <<PY
import json; d = json.loads(open("$O/sweep.json").read().strip().splitlines()[-1])
print("nu $1 ratio $2:", {k: d[k] for k in ("converged", "snes_its", "ksp_its", "snes_solve_s")})
PY
done
