#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R && timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest_gpu_final.log 2>&1; rc=$?; echo "pytest gpu rc $rc"; tail -4 $O/pytest_gpu_final.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"; python3 -c "
import json; d=json.loads(open('$O/bench_final.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
