#!/bin/bash
# usage: tools/ab_stats.sh name [ENV=VAL ...] -- rocprofv3 kernel time of the fused + assemble kernels for one configuration
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; v=$1; shift
rm -rf /tmp/ab_$v
env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > /tmp/ab_$v.log 2>&1 || { echo "$v failed"; tail -3 /tmp/ab_$v.log; }
f=$(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1)
echo "$v $(grep 'k_fused_pencil<5, 5, 6' $f | awk -F, '{print "fused_us", $(NF-4)/1000}') $(grep 'k_assemble' $f | awk -F, '{print "assemble_us", $(NF-4)/1000}') $(tail -1 /tmp/ab_$v.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4))')"
