#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R
timeout -k 5 120 tools/microbench/xcd_handoff > $O/xcd_handoff.txt 2>&1; echo "handoff rc $?"; cat $O/xcd_handoff.txt
CEED_MI355X_ASM_SPINS=5000 timeout -k 5 240 python -u tools/r2_diag_gated.py > $O/diag1.log 2>&1; echo "diag rc $?"; tail -12 $O/diag1.log
