#!/bin/bash
# lockstep cost: the pencil kernel as 256-thread workgroups of four waves with two barriers per group (no merge yet)
L=ceedpetscsolid_amd/csrc/libceed_mi355x.so; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; cp $R/$L /tmp/orig.so
cd /tmp && export TMPDIR=/tmp
for v in orig maxilp iterilp memclause bias0 orig maxilp iterilp memclause bias0; do
  if [ $v = orig ]; then cp /tmp/orig.so $R/$L; else cp $R/tools/variants/lib_$v.so $R/$L; fi
  rm -rf /tmp/ab_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > /tmp/ab_$v.log 2>&1 || { echo "$v failed"; tail -3 /tmp/ab_$v.log; }
  f=$(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1)
  echo "$v $(tail -1 /tmp/ab_$v.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4))') $(grep 'k_fused_pencil<5, 5, 6' $f | awk -F, '{print "fused_us", $(NF-4)/1000}') $(grep 'k_assemble' $f | awk -F, '{print "assemble_us", $(NF-4)/1000}')"
done
cp /tmp/orig.so $R/$L
