#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in 0 1 2 4 7; do for w in 4 8; do
  rm -rf /tmp/kt_$w
  CEED_MI355X_GATED_KDBG=$k CEED_MI355X_GATED_DEBUG=16 CEED_MI355X_ASM_WAVES=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python3 $R/bench.py --steps 20 --warmup 3 --prewarm-ms 50 --no-cpu-baseline > $O/exp7.log 2>&1
  f=$(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1)
  echo "kdbg=$k waves=$w: $(grep -E 'k_assemble_gated|k_assemble_tail' $f | awk -F, '{print $1, $(NF-4)/1000}' | tr '\n' ' ')"
done; done
