#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for S in 3; do
rm -rf /tmp/ptrace
CEED_MI355X_ASSEMBLE=pipelined CEED_MI355X_PIPE_CHAINS=1 CEED_MI355X_PIPE_SEGMENTS=$S timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptrace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/ptrace.log 2>&1 || { tail -5 $O/ptrace.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/ptrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_fused_pencil<5, 5, 6" in r["Kernel_Name"] or "k_assemble" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-6 * $S:]          # the last three applies
t0 = int(rows[0]["Start_Timestamp"])
print("S=$S: kernel, queue, start us, end us, duration us")
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(("fused " if "fused" in r["Kernel_Name"] else "asm   "), r.get("Queue_Id", "?"), f"{s:9.1f} {e:9.1f} {e - s:8.1f}")
PY
done
