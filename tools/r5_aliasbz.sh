#!/bin/bash
# Round 5, VERDICT r4 item 3: upper bound of the LDS-occupancy lever at Q = 7 -- the timing-only variant `aliasbz` (BZ aliased onto BX:
# a 6 Q^3 slab, 8 instead of 6 waves per CU; WRONG results) against the default on one box, default / variant / default-again.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "import sys, json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('%.4f ms  %.1f GDoF/s  %s' % (d['ms_per_step'], d['value'] / 1e3, d['config'].get('kernel', '')[:60]))"; }
for w in "--workload box --degree 6 --nr 32 --nth 32 --nz 32" "--workload box --degree 6 --nr 64 --nth 64 --nz 64 --steps 20"; do
  echo "== $w"
  for v in default aliasbz default aliasbz default; do
    if [ $v = default ]; then echo -n "$v: "; run $w; else echo -n "$v: "; CEEDPETSCSOLID_MI355X_LIB=$R/tools/variants/$v/libceed_mi355x.so run $w; fi
  done
done
