#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R
for d in 2 0; do
CEED_MI355X_GATED_DEBUG=$d CEED_MI355X_ASM_SPINS=2000 timeout -k 5 30 python -u tools/r2_diag2.py tiny > $O/diag3_$d.log 2>&1; rc=$?; echo "debug=$d rc $rc"; tail -2 $O/diag3_$d.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
done
CEED_MI355X_ASM_SPINS=5000 timeout -k 5 240 python -u tools/r2_diag_gated.py > $O/diag1.log 2>&1; echo "diag rc $?"; tail -12 $O/diag1.log
