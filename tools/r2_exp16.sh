#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R && timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 600 -k "extension_free or config3_full or unstructured" > $O/pytest_new.log 2>&1; rc=$?; echo "pytest new rc $rc"; tail -8 $O/pytest_new.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for v in "mesh44928 --workload mesh" "cyl45100 --nz 41"; do
  set -- $v; n=$1; shift
  rm -rf /tmp/kt_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$n -- python3 $R/bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline > $O/exp16_$n.json 2> $O/exp16_$n.err
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  echo "$n: $(tail -1 $O/exp16_$n.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4), "GDoF/s", round(d["value"]/1e3,2), "dofs", d["config"]["global_dofs"])') $(grep -E 'k_assemble|k_fused_pencil<5, 5, 6' $f | awk -F, '{print $1, $(NF-4)/1000}' | sed -e 's/cps:://g; s/(.*)//' | tr '\n' ' ')"
done
