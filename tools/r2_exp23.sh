#!/bin/bash
# stored-state runs padded to whole 128-byte lines (timing-only variant lib_qpad.so) -- kernel time and FETCH_SIZE
L=ceedpetscsolid_amd/csrc/libceed_mi355x.so; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cp $R/$L /tmp/orig.so
cd /tmp && export TMPDIR=/tmp; export CEED_MI355X_QPAD_ALLOC=1
for v in orig lex orig lex; do
  if [ $v = orig ]; then cp /tmp/orig.so $R/$L; else cp $R/tools/variants/lib_$v.so $R/$L; fi
  rm -rf /tmp/ab_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /tmp/ab_$v.log 2>&1 || { echo "$v failed"; tail -3 /tmp/ab_$v.log; }
  f=$(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1)
  echo "$v $(tail -1 /tmp/ab_$v.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4))') $(grep 'k_fused_pencil<5, 5, 6' $f | awk -F, '{print "fused_us", $(NF-4)/1000}') $(grep 'k_assemble' $f | awk -F, '{print "assemble_us", $(NF-4)/1000}')"
done
for v in orig lex; do
  if [ $v = orig ]; then cp /tmp/orig.so $R/$L; else cp $R/tools/variants/lib_$v.so $R/$L; fi
  rm -rf /tmp/pf_$v
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$v -- python3 $R/bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline > /tmp/pf_$v.log 2>&1
  f=$(find /tmp/pf_$v -name "*counter_collection.csv" | head -1)
  python3 - $f $v <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if ("k_fused_pencil<5, 5, 6" in r["Kernel_Name"] or "k_assemble" in r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE": agg[r["Kernel_Name"][:24]].append(float(r["Counter_Value"]))
print(sys.argv[2], "FETCH_SIZE (KB counted, x2 for bytes):", {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
done
cp /tmp/orig.so $R/$L
