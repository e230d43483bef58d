#!/bin/bash
# Three SQ counter passes (rocprofv3 --pmc, counters only -- no trace domain beside them) over ONE python command, summarised per kernel
# whose name contains <pattern>:   tools/pmc_cmd.sh <tag> <kernel pattern> python3 <script> [args ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; export KPAT=$2; shift 2
O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
  i=$((i+1))
  rm -rf /tmp/pmcc_${tag}_$i
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcc_${tag}_$i -- "$@" > $O/${tag}_sq_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/${tag}_sq_$i.log; }
  f=$(find /tmp/pmcc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_sq_pass$i.csv
done
python3 - <<PY > $O/${tag}_sq_summary.txt
import csv,glob,collections,os
pat=os.environ["KPAT"].split("|")
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/${tag}_sq_pass*.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if any(p in k for p in pat): agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in sorted(agg.items()):
    m={n:sum(v)/len(v) for n,v in c.items()}
    print(k)
    for n in sorted(m): print("   %-24s %.4g" % (n, m[n]))
    wc=m.get("SQ_WAVE_CYCLES",0)
    if wc:
        for n in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_SCA","SQ_INST_CYCLES_VMEM","SQ_INST_CYCLES_SALU"):
            if n in m: print("   frac of wave-cycles %-22s %.3f" % (n, m[n]/wc))
    bc=m.get("SQ_BUSY_CYCLES",0)
    if bc and "SQ_LDS_IDX_ACTIVE" in m: print("   LDS_IDX_ACTIVE/BUSY_CYCLES %.3f  conflict share %.3f" % (m["SQ_LDS_IDX_ACTIVE"]/bc, m.get("SQ_LDS_BANK_CONFLICT",0)/max(m["SQ_LDS_IDX_ACTIVE"],1)))
PY
cat $O/${tag}_sq_summary.txt | grep -E "^void|^cps|frac of|conflict" 
