#!/bin/bash
# repeated A/B of the shell E-vector order under the pipelined default and the serial form (variants/ built out of tree with -DCPS_SHELL_LEX)
R=$GRAFT_REPO_ROOT; cd /tmp
run() { python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$TAG %.4f ms' % d['ms_per_step'])"; }
for rep in 1 2 3 4; do
TAG="pipelined face-major " run
TAG="pipelined lexicograph" CEEDPETSCSOLID_MI355X_LIB=$R/variants/libceed_mi355x_lex.so run
done
for rep in 1 2; do
TAG="serial    face-major " CEED_MI355X_ASSEMBLE=serial run
TAG="serial    lexicograph" CEED_MI355X_ASSEMBLE=serial CEEDPETSCSOLID_MI355X_LIB=$R/variants/libceed_mi355x_lex.so run
done
