#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R && timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 -k "gated_assembly or all_operators or bitwise or split" > $O/pytest_sched.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/pytest_sched.log
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
for nz in 90 4; do for v in "static X=1" "dynamic CEED_MI355X_SCHED=dynamic"; do
  set -- $v; n=$1; shift
  env "$@" timeout -k 10 200 python3 $R/bench.py --nz $nz --steps 100 --warmup 5 --no-cpu-baseline > $O/exp21_${n}_$nz.json 2> $O/exp21_${n}_$nz.err
  echo "nz=$nz $n: $(tail -1 $O/exp21_${n}_$nz.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4), "kernel_us", round(d["roofline"]["kernel_avg_us"],1))')"
done; done
cd $R
for v in "static X=1" "dynamic CEED_MI355X_SCHED=dynamic"; do set -- $v; n=$1; shift
  env "$@" python examples/solve_config3.py --coarse assembled --graph 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3 $n', d['snes_solve_s'], d['snes_its'], d['ksp_its'])"
done
