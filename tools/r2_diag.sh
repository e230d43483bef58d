#!/bin/bash
# the sum-factorised diagonal kernel: parity tests that cover get_diag on every model / level, then its time against the direct kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
cd /tmp && export TMPDIR=/tmp
for mode in sf direct; do
  rm -rf /tmp/prof_diag_$mode
  if [ $mode = direct ]; then export CEED_MI355X_DIAG=direct; else unset CEED_MI355X_DIAG; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_diag_$mode -- python3 $R/examples/solve_config3.py --coarse amg --graph > $O/diag_$mode.json 2> $O/diag_$mode.err || { tail -5 $O/diag_$mode.err; exit 1; }
  echo "== $mode"; python3 - "$(find /tmp/prof_diag_$mode -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_diag" in r["Name"]: print("  %-40s calls %3s  avg %8.1f us" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1000))
PY
  tail -1 $O/diag_$mode.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('converged','snes_its','ksp_its','snes_solve_s')})"
done
unset CEED_MI355X_DIAG
python3 $R/examples/solve_config3.py --coarse amg --graph 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('unprofiled', {k: d[k] for k in ('converged','snes_its','ksp_its','snes_solve_s')})"
