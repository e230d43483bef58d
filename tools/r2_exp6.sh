#!/bin/bash
# gated kernel's own throughput: run it AFTER the fused kernel (debug 16) at several wave counts
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 4 8 16; do
  rm -rf /tmp/kt_$w
  CEED_MI355X_GATED_DEBUG=16 CEED_MI355X_ASM_WAVES=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/exp6_$w.log 2>&1
  f=$(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1)
  echo "serial-after waves=$w: $(grep -E 'k_assemble_gated|k_fused_pencil<5, 5, 6|k_assemble_tail' $f | awk -F, '{print $1, $(NF-4)/1000}' | tr '\n' ' ')"
done
for w in 4 8; do
  rm -rf /tmp/ktg_$w
  CEED_MI355X_ASM_WAVES=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktg_$w -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/exp6g_$w.log 2>&1
  f=$(find /tmp/ktg_$w -name "*kernel_stats.csv" | head -1)
  echo "beside waves=$w: $(grep -E 'k_assemble_gated|k_fused_pencil<5, 5, 6|k_assemble_tail' $f | awk -F, '{print $1, $(NF-4)/1000}' | tr '\n' ' ')"
  cp $(find /tmp/ktg_$w -name "*kernel_trace.csv" | head -1) $O/kernel_trace_gated_w$w.csv
done
