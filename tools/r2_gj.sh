#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
timeout -k 10 600 python -u -m pytest tests/test_amg.py -m gpu -x -q --timeout 400 > $O/gj_tests.log 2>&1 || { tail -25 $O/gj_tests.log; exit 1; }
tail -1 $O/gj_tests.log
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_gj
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gj -- python3 $R/examples/solve_config3.py --coarse amg --graph > $O/gj.json 2> $O/gj.err || { tail -5 $O/gj.err; exit 1; }
python3 - "$(find /tmp/prof_gj -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_gj" in r["Name"] or "k_symm" in r["Name"]: print("  %-40s calls %4s  avg %8.1f us" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1000))
PY
for i in 1 2; do python3 $R/examples/solve_config3.py --coarse amg --graph 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('unprofiled', {k: d[k] for k in ('converged','snes_its','ksp_its','snes_solve_s')})"; done
