#!/bin/bash
# GPU check of the aggregation coarse solve: its tests, then config 3 with the Chebyshev and with the aggregation coarse solver.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
timeout -k 10 600 python -u -m pytest tests/test_amg.py -m gpu -x -q --timeout 500 > $O/amg_tests.log 2>&1; rc=$?
tail -5 $O/amg_tests.log
[ $rc -eq 0 ] || exit $rc
for c in assembled amg; do
  timeout -k 10 300 python -u examples/solve_config3.py --coarse $c --graph > $O/config3_$c.json 2> $O/config3_$c.err || { tail -5 $O/config3_$c.err; exit 1; }
  python - <<PY
import json; d = json.loads(open("$O/config3_$c.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("coarse_solver", "converged", "snes_its", "ksp_its", "coarse_cg_its", "jacobian_applies", "coarse_spmv", "setup_s", "snes_solve_s", "max_abs_displacement")})
PY
done
