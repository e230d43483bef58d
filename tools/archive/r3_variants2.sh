#!/bin/bash
# Variant libraries (tools/variants/<name>/*.so, tools/mkvariant.sh) against the default on one box, a wider set of workloads than
# r3_variants.sh: config 4, 13 200 hexes, hyperSS, config 5's 32^3 block at p = 6, p = 2 and p = 3 cylinders, the emulated rank of 8.
#   usage: r3_variants2.sh [-w "c4 nz12 ..."] variant ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O /tmp/dflt_libs; cd /tmp; export TMPDIR=/tmp
W="c4 nz12 ss box6 p2 em"
if [ "$1" = "-w" ]; then W=$2; shift 2; fi
cp $R/ceedpetscsolid_amd/csrc/*.so /tmp/dflt_libs/
args() { case $1 in
  c4) echo "";; nz12) echo "--nz 12 --steps 100";; ss) echo "--problem hyperSS";;
  box6) echo "--workload box --nr 32 --nth 32 --nz 32 --degree 6";; p2) echo "--degree 2";; p3) echo "--degree 3";;
  lin) echo "--problem linElas";; em) echo "--emulate-rank 3 --of 8 --steps 100 --warmup 10";; esac; }
run() { name=$1
  for t in $W; do
    python3 $R/bench.py $(args $t) --no-cpu-baseline --cold-idle-s 0 > $O/v2_${name}_$t.json 2> $O/v2_${name}_$t.err || { echo "$name $t failed"; tail -2 $O/v2_${name}_$t.err; }
  done
  python3 -c "
import json
out=[]
for t in '$W'.split():
    try:
        d=[json.loads(l) for l in open('$O/v2_${name}_'+t+'.json') if l.startswith('{')][-1]
        out.append('%s %.4f' % (t, d['emulated_rank']['us_per_apply_incl_exchange']/1e3 if t=='em' else d['ms_per_step']))
    except Exception as e: out.append(t+' ERR')
print('%-14s' % '$name', '  '.join(out))"
}
run default
for v in "$@"; do
  cp $R/tools/variants/$v/*.so $R/ceedpetscsolid_amd/csrc/
  (cd $R && python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-60)
  run $v
done
cp /tmp/dflt_libs/*.so $R/ceedpetscsolid_amd/csrc/
run default_again
