#!/bin/bash
# Re-tune the pipelined transpose for the cleaned kernel (config 4, same box): segments x rounds of the last segment.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
b() { tag=$1; shift; env "$@" python3 $R/bench.py --no-cpu-baseline --cold-idle-s 0 > $O/ps_$tag.json 2> $O/ps_$tag.err || { echo "$tag failed"; return; }
  python3 -c "
import json; d=json.loads(open('$O/ps_$tag.json').read()); print('%-22s %8.4f ms  frac %.3f  %s' % ('$tag', d['ms_per_step'], d['roofline']['frac'], d['config']['assembly']))"; }
b default
b serial CEED_MI355X_ASSEMBLE=serial
for s in 2 3 4 5; do for l in 2 4 6; do b seg${s}_last${l} CEED_MI355X_PIPE_SEGMENTS=$s CEED_MI355X_PIPE_LAST=$l; done; done
b default_again
