#!/bin/bash
# Round 4: variant libraries (tools/variants/<name>/*.so, tools/mkvariant.sh) against the default on ONE box.
#   usage: r4_variants.sh [-w "c4 nz12 ..."] [-s] variant ...      (-s: also the serial-form kernel times from rocprofv3 --stats)
# Default first and last; per workload ms per step of the default (pipelined) form; with -s the fused / assemble kernel times of
# the serial form (CEED_MI355X_ASSEMBLE=serial) on config 4.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O /tmp/dflt_libs; cd /tmp; export TMPDIR=/tmp
W="c4 nz12 ss box6"; STATS=0
while [ "${1:0:1}" = "-" ]; do case $1 in -w) W=$2; shift 2;; -s) STATS=1; shift;; esac; done
cp $R/ceedpetscsolid_amd/csrc/*.so /tmp/dflt_libs/
args() { case $1 in
  c4) echo "";; nz12) echo "--nz 12 --steps 100";; ss) echo "--problem hyperSS";;
  box6) echo "--workload box --nr 32 --nth 32 --nz 32 --degree 6";; p2) echo "--degree 2";; p3) echo "--degree 3";; mesh) echo "--workload mesh";;
  lin) echo "--problem linElas";; scr) echo "--scramble all";; scro) echo "--scramble order";; scrm) echo "--scramble all --reorder";; scrom) echo "--scramble order --reorder";; em) echo "--emulate-rank 3 --of 8 --steps 100 --warmup 10";; box64) echo "--workload box --nr 64 --nth 64 --nz 64 --degree 6 --steps 20";; box5) echo "--workload box --nr 36 --nth 36 --nz 36 --degree 5";; box6ss) echo "--workload box --nr 32 --nth 32 --nz 32 --degree 6 --problem hyperSS";; esac; }
run() { name=$1
  for t in $W; do
    timeout -k 10 300 python3 $R/bench.py $(args $t) --no-cpu-baseline --cold-idle-s 0 > $O/v_${name}_$t.json 2> $O/v_${name}_$t.err || { echo "$name $t failed"; tail -2 $O/v_${name}_$t.err; }
  done
  python3 -c "
import json
out=[]
for t in '$W'.split():
    try:
        d=[json.loads(l) for l in open('$O/v_${name}_'+t+'.json') if l.startswith('{')][-1]
        out.append('%s %.4f' % (t, d['emulated_rank']['us_per_apply_incl_exchange']/1e3 if t=='em' else d['ms_per_step']))
    except Exception as e: out.append(t+' ERR')
print('%-16s' % '$name', '  '.join(out))"
  if [ $STATS = 1 ]; then
    rm -rf /tmp/st_$name
    CEED_MI355X_ASSEMBLE=serial timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --cold-idle-s 0 > /tmp/st_$name.log 2>&1 || { echo "$name stats failed"; tail -3 /tmp/st_$name.log; }
    f=$(find /tmp/st_$name -name "*kernel_stats.csv" | head -1)
    echo "   serial form: $(grep 'k_fused_pencil<5, 5, 6' $f | awk -F, '{print "fused_us", $(NF-4)/1000}') $(grep 'k_assemble' $f | awk -F, '{print "assemble_us", $(NF-4)/1000}') $(tail -1 /tmp/st_$name.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", round(d["ms_per_step"],4))')"
    cp $f $O/stats_$name.csv 2>/dev/null
  fi
}
run default
for v in "$@"; do
  cp $R/tools/variants/$v/*.so $R/ceedpetscsolid_amd/csrc/
  run $v
done
cp /tmp/dflt_libs/*.so $R/ceedpetscsolid_amd/csrc/
run default_again
