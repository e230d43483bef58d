#!/bin/bash
# solver tests on the device, then config 3 timings (Chebyshev and aggregation coarse solve) and the kernel profile of the latter
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
timeout -k 10 900 python -u -m pytest tests/test_amg.py tests/test_solver.py -m gpu -x -q --timeout 600 > $O/amg_tests3.log 2>&1 || { tail -15 $O/amg_tests3.log; exit 1; }
tail -1 $O/amg_tests3.log
for c in assembled amg amg; do
  timeout -k 10 300 python -u examples/solve_config3.py --coarse $c --graph > $O/config3_$c.json 2> $O/config3_$c.err || { tail -5 $O/config3_$c.err; exit 1; }
  python - <<PY
import json; d = json.loads(open("$O/config3_$c.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("coarse_solver", "converged", "snes_its", "ksp_its", "jacobian_applies", "coarse_spmv", "setup_s", "snes_solve_s")})
PY
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_amg
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_amg -- python3 $R/examples/solve_config3.py --coarse amg --graph > $O/config3_amg_prof.json 2> $O/config3_amg_prof.err || { tail -5 $O/config3_amg_prof.err; exit 1; }
cp $(find /tmp/prof_amg -name "*kernel_stats.csv" | head -1) $O/config3_amg_kernel_stats.csv
head -24 $O/config3_amg_kernel_stats.csv | cut -c1-60,100-260
