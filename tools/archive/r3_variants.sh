#!/bin/bash
# Variant libraries (tools/variants/<name>/*.so, built out of tree with EXTRA_HIPFLAGS) against the default, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O /tmp/dflt_libs; cd /tmp; export TMPDIR=/tmp
cp $R/ceedpetscsolid_amd/csrc/*.so /tmp/dflt_libs/
run() { name=$1
  python3 $R/bench.py --no-cpu-baseline --cold-idle-s 0 > $O/var_${name}_c4.json 2> $O/var_${name}_c4.err || { echo "$name c4 failed"; tail -2 $O/var_${name}_c4.err; }
  python3 $R/bench.py --nz 12 --no-cpu-baseline --cold-idle-s 0 --steps 100 > $O/var_${name}_nz12.json 2> $O/var_${name}_nz12.err
  python3 $R/bench.py --problem hyperSS --no-cpu-baseline --cold-idle-s 0 > $O/var_${name}_ss.json 2> $O/var_${name}_ss.err
  python3 -c "
import json
for t in ('c4','nz12','ss'):
    try:
        d=json.loads(open('$O/var_${name}_'+t+'.json').read()); print('%-10s %-5s %8.4f ms  %7.2f GDoF/s  frac %.3f' % ('$name', t, d['ms_per_step'], d['value']/1e3, d['roofline']['frac']))
    except Exception as e: print('$name', t, 'ERR', e)"
}
run default
for v in "$@"; do
  cp $R/tools/variants/$v/*.so $R/ceedpetscsolid_amd/csrc/
  (cd $R && python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1)
  run $v
done
cp /tmp/dflt_libs/*.so $R/ceedpetscsolid_amd/csrc/
run default_again
