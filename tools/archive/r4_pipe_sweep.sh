#!/bin/bash
# Round 4: segments / last-segment rounds of the pipelined transpose on the round's final kernel (config 4), one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { env "$@" timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --cold-idle-s 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], d['config']['assembly'][:48])"; }
echo "default            $(run A=1)"
for seg in 2 3 4 5; do for last in 2 4 6; do echo "segments $seg last $last  $(run CEED_MI355X_PIPE_SEGMENTS=$seg CEED_MI355X_PIPE_LAST=$last)"; done; done
echo "serial             $(run CEED_MI355X_ASSEMBLE=serial)"
echo "default again      $(run A=1)"
