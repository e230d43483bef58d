#!/bin/bash
# Round 5: upper bound of the occupancy lever at Q = 5 (config 4's kernel): timing-only variants (WRONG results) on one box --
#   aliasbz5   BZ aliased onto BX (12.5 instead of 18.5 KB of LDS per wave) AND the registers held to 168: 12 waves per CU instead of 8
#   aliasbz5w2 the smaller slab alone (still 8 waves per CU by registers): control
#   minw3      the register limit alone (LDS still caps at 8 waves per CU): what the squeeze costs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "import sys, json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('%.4f ms  %.1f GDoF/s  spread %.2f %%' % (d['ms_per_step'], d['value'] / 1e3, d['timed_blocks']['spread_pct']))"; }
for w in "" "--problem hyperSS"; do
  echo "== config 4 ${w:-hyperFS}"
  for v in default aliasbz5 aliasbz5w2 minw3 default aliasbz5 default; do
    if [ $v = default ]; then echo -n "$v: "; run $w; else echo -n "$v: "; CEEDPETSCSOLID_MI355X_LIB=$R/tools/variants/$v/libceed_mi355x.so run $w; fi
  done
done
