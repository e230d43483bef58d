#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_solve; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_solve -- python3 $R/examples/solve_config3.py --coarse assembled --graph > $O/solve_trace.json 2> $O/solve_trace.err
python3 -c "
import json; d=json.loads(open('$O/solve_trace.json').read()); print('solve s', d['snes_solve_s'], 'ksp', d['ksp_its'], 'jac applies', d['jacobian_applies'])"
f=$(find /tmp/kt_solve -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time s", tot / 1e9)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("  %-70s calls %7s  avg %8.1f us  total %6.3f s" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e9))
PY
