#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.2f GDoF/s %.4f ms" % (sys.argv[1], d["value"]/1000, d["ms_per_step"]))'
A5="--workload box --degree 6 --nr 64 --nth 64 --nz 64"
for rep in 1 2; do
python3 $R/variants/old_tree/bench.py --steps 30 --warmup 3 --no-cpu-baseline $A5 2>/dev/null | python3 -c "$fmt" "config5 tree of 11:40"
CEED_MI355X_ASSEMBLE=serial python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline $A5 2>/dev/null | python3 -c "$fmt" "config5 serial now"
CEED_MI355X_ASSEMBLE=serial CEEDPETSCSOLID_MI355X_LIB=$R/variants/libceed_mi355x_nostagger.so python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline $A5 2>/dev/null | python3 -c "$fmt" "config5 serial now, no stagger"
done
python3 $R/variants/old_tree/bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "$fmt" "config4 tree of 11:40"
CEED_MI355X_ASSEMBLE=serial python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "$fmt" "config4 serial now"
