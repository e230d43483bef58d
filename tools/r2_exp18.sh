#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R
timeout -k 5 120 tools/microbench/graph_memset_repro > $O/graph_memset_repro.txt 2>&1; echo "repro rc $?"; cat $O/graph_memset_repro.txt
timeout -k 5 120 python -u tools/graph_replay_check.py > $O/graph_replay_kernel_fill.txt 2>&1; echo "replay check (fill kernel) rc $?"; tail -4 $O/graph_replay_kernel_fill.txt
CEED_MI355X_GRAPH_MEMSET=1 timeout -k 5 120 python -u tools/graph_replay_check.py > $O/graph_replay_memset_nodes.txt 2>&1; echo "replay check (memset nodes) rc $?"; tail -4 $O/graph_replay_memset_nodes.txt
