#!/bin/bash
# A/B of the shell E-vector order (face-major = built default, lexicographic = variants/libceed_mi355x_lex.so), serial form
R=$GRAFT_REPO_ROOT; cd /tmp
run() { tag=$1; shift; python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag: %.2f GDoF/s %.4f ms | %s' % (d['value']/1000, d['ms_per_step'], d['config']['assembly'][:24]))"; }
for form in serial default; do
  if [ $form = serial ]; then export CEED_MI355X_ASSEMBLE=serial; else unset CEED_MI355X_ASSEMBLE; fi
  for lib in face lex; do
    if [ $lib = lex ]; then export CEEDPETSCSOLID_MI355X_LIB=$R/variants/libceed_mi355x_lex.so; else unset CEEDPETSCSOLID_MI355X_LIB; fi
    run "config5 $form $lib" --workload box --degree 6 --nr 64 --nth 64 --nz 64
    run "p6box32 $form $lib" --workload box --degree 6 --nr 32 --nth 32 --nz 32
    run "config4 $form $lib"
    run "p3box64 $form $lib" --workload box --degree 3 --nr 64 --nth 64 --nz 64
  done
done
