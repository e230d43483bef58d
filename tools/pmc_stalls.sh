#!/bin/bash
# Stall-side SQ counters of the fused kernel (config 4 by default): average in-flight LDS / VMEM / SMEM instructions and their
# latencies (INST_LEVEL_x / INSTS_x), LDS FIFO back-pressure, instruction fetch.   usage: pmc_stalls.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
         "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_LEVEL_WAVES" \
         "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/st_${tag}_$i
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/st_${tag}_$i -- python3 $R/bench.py "$@" --steps 5 --warmup 2 --prewarm-ms 0 --cold-idle-s 0 --no-cpu-baseline > $O/st_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/st_${tag}_$i.log; }
  f=$(find /tmp/st_${tag}_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/st_${tag}_pass$i.csv
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/st_${tag}_pass*.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_fused" in k and ", 6," in k or "k_fused" in k and ", 17," in k: agg[k[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    m={n:sum(v)/len(v) for n,v in c.items()}
    print(k)
    for n in sorted(m): print("   %-30s %.5g" % (n, m[n]))
    def ratio(a,b,txt):
        if a in m and b in m and m[b]: print("   >> %-52s %.1f" % (txt, m[a]/m[b]))
    ratio("SQ_INST_LEVEL_LDS","SQ_INSTS_LDS","mean cycles an LDS instruction is in flight")
    if "SQ_INST_LEVEL_VMEM" in m: print("   >> %-52s %.1f" % ("mean cycles a VMEM instruction is in flight", m["SQ_INST_LEVEL_VMEM"]/max(m.get("SQ_INSTS_VMEM_RD",0)+m.get("SQ_INSTS_VMEM_WR",0),1)))
    ratio("SQ_INST_LEVEL_SMEM","SQ_INSTS_SMEM","mean cycles an SMEM instruction is in flight")
    ratio("SQ_IFETCH_LEVEL","SQ_IFETCH","mean cycles an instruction fetch is in flight")
    ratio("SQ_INST_LEVEL_LDS","SQ_BUSY_CYCLES","mean LDS instructions in flight per SQ (busy cycles)")
    ratio("SQ_LDS_CMD_FIFO_FULL","SQ_BUSY_CYCLES","LDS command FIFO full / busy cycles")
    ratio("SQ_LDS_DATA_FIFO_FULL","SQ_BUSY_CYCLES","LDS data FIFO full / busy cycles")
    ratio("SQ_LDS_ADDR_CONFLICT","SQ_BUSY_CYCLES","LDS address conflict cycles / busy cycles")
PY
