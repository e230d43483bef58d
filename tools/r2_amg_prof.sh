#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_amg
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_amg -- python3 $R/examples/solve_config3.py --coarse amg --graph > $O/config3_amg_prof.json 2> $O/config3_amg_prof.err || { tail -5 $O/config3_amg_prof.err; exit 1; }
cp $(find /tmp/prof_amg -name "*kernel_stats.csv" | head -1) $O/config3_amg_kernel_stats.csv
head -30 $O/config3_amg_kernel_stats.csv | cut -c1-220
tail -1 $O/config3_amg_prof.json | cut -c1-600
