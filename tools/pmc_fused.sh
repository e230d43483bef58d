#!/bin/bash
# PMC passes over bench.py for the fused kernel (run on the GPU box): tools/pmc_fused.sh <tag> [env...]
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift     # BENCH_ARGS: extra bench.py arguments (workload)
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
  i=$((i+1))
  env "$@" timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$i -- python3 $R/bench.py $BENCH_ARGS --steps 5 --warmup 2 --blocks 1 --no-clock-probe --prewarm-ms 0 --cold-idle-s 0 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_${tag}_$i.log; }
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc_${tag}_pass$i.csv
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_${tag}_pass*.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_fused" in k: agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    m={n:sum(v)/len(v) for n,v in c.items()}
    print(k, "VGPR/regs n/a")
    for n in sorted(m): print("   %-24s %.4g" % (n, m[n]))
    wc=m.get("SQ_WAVE_CYCLES",0)
    if wc:
        for n in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_SCA","SQ_INST_CYCLES_VMEM","SQ_INST_CYCLES_SALU"):
            if n in m: print("   frac of wave-cycles %-22s %.3f" % (n, m[n]/wc))
    bc=m.get("SQ_BUSY_CYCLES",0)
    if bc and "SQ_LDS_IDX_ACTIVE" in m: print("   LDS_IDX_ACTIVE/BUSY_CYCLES %.3f  conflict share %.3f" % (m["SQ_LDS_IDX_ACTIVE"]/bc, m.get("SQ_LDS_BANK_CONFLICT",0)/max(m["SQ_LDS_IDX_ACTIVE"],1)))
PY
