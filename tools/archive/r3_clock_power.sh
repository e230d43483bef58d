#!/bin/bash
# Shader clock and socket power while the metric loop runs (rocm-smi sampled beside a long bench.py run), idle for comparison.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
smp() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'; echo; }
{
echo "idle:"; smp
for p in hyperFS hyperSS linElas; do
  python3 $R/bench.py --problem $p --steps 12000 --no-cpu-baseline --cold-idle-s 0 > $O/clk_$p.json 2> $O/clk_$p.err &
  pid=$!
  sleep 14
  for i in 1 2 3; do echo "$p under load:"; smp; sleep 1; done
  wait $pid
  python3 -c "
import json; d=json.loads(open('$O/clk_$p.json').read()); print('$p ms_per_step', d['ms_per_step'])"
done
} 2>&1 | tee $O/clock_power.txt
