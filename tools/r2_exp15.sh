#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R
run() { n=$1; shift; env "$@" timeout -k 5 60 python -u tools/r2_diag2.py mid > $O/diag7_$n.log 2>&1; echo "$n rc $?"; grep -E "jacobian 2|residual norm" $O/diag7_$n.log | cut -c1-160; }
run gated_nogatedkernel CEED_MI355X_ASSEMBLE=gated CEED_MI355X_GATED_DEBUG=1
run gated CEED_MI355X_ASSEMBLE=gated
run default_nodirect CEED_MI355X_DIRECT=0
run default_static_queue CEED_MI355X_SCHED=static
run default X=1
