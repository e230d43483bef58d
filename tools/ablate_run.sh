#!/bin/bash
# timing-only ablation builds (tools/variants/lib_*.so, built with -DCPS_ABLATE_*): fused-kernel time per variant
L=ceedpetscsolid_amd/csrc/libceed_mi355x.so; cp $L /tmp/orig.so
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in orig noqd noqf nopass nostore noqfpass onlymem; do
  if [ $v = orig ]; then cp /tmp/orig.so $R/$L; else cp $R/tools/variants/lib_$v.so $R/$L; fi
  rm -rf /tmp/ab_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > /tmp/ab_$v.log 2>&1 || { echo "$v failed"; tail -3 /tmp/ab_$v.log; }
  f=$(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1)
  echo "$v $(grep 'k_fused_pencil<5, 5, 6' $f | awk -F, '{print "fused_us", $(NF-4)/1000}') $(grep 'k_assemble' $f | awk -F, '{print "assemble_us", $(NF-4)/1000}')"
done
cp /tmp/orig.so $R/$L
