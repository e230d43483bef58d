#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R
CEED_MI355X_ASSEMBLE=serial timeout -k 5 60 python -u tools/r2_diag2.py tiny > $O/diag2_serial.log 2>&1; echo "serial rc $?"; tail -4 $O/diag2_serial.log
CEED_MI355X_ASM_SPINS=2000 timeout -k 5 60 python -u tools/r2_diag2.py tiny > $O/diag2_gated.log 2>&1; echo "gated tiny rc $?"; tail -8 $O/diag2_gated.log
CEED_MI355X_ASM_SPINS=2000 AMD_LOG_LEVEL=3 timeout -k 5 40 python -u tools/r2_diag2.py tiny > $O/diag2_gated_log.log 2>&1; echo "gated tiny logged rc $?"; tail -25 $O/diag2_gated_log.log | cut -c1-300
