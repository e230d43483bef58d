#!/bin/bash
# Round-3 GPU batch B: whole GPU suite, derived-state A/B (config 4, config 5), reserved-CU two-stream split-phase apply.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q --timeout 600 > $O/gputest5.log 2>&1; rc=$?; tail -8 $O/gputest5.log
[ $rc -le 1 ] || exit 1
cd /tmp; export TMPDIR=/tmp
b() { tag=$1; shift; env "$@" python3 $R/bench.py $A --no-cpu-baseline --cold-idle-s 0 > $O/ab_$tag.json 2> $O/ab_$tag.err || { echo "$tag failed"; tail -3 $O/ab_$tag.err; return; }
  python3 -c "
import json; d=[json.loads(l) for l in open('$O/ab_$tag.json') if l.startswith('{')][-1]; print('%-34s %8.2f GDoF/s %8.4f ms frac %.3f  %s' % ('$tag', d['value']/1e3, d['ms_per_step'], d['roofline']['frac'], d['config']['kernel']))"; }
for rep in 1 2; do
A=""; b c4_derived_$rep; b c4_plain_$rep CEED_MI355X_DERIVED=0
A="--workload box --degree 6 --nr 32 --nth 32 --nz 32"; b c5blk_derived_$rep; b c5blk_plain_$rep CEED_MI355X_DERIVED=0
done
A="--workload box --degree 6 --nr 64 --nth 64 --nz 64 --steps 20"; b c5whole_derived; b c5whole_plain CEED_MI355X_DERIVED=0; b c5whole_plain_general CEED_MI355X_DERIVED=0 CEED_MI355X_AFFINE=0
A="--problem hyperSS"; b c4_hyperSS
A="--problem linElas"; b c4_linElas
# kernel time of the fused kernel with and without the derived state (same box, rocprofv3)
for v in derived plain; do
  if [ $v = plain ]; then E="CEED_MI355X_DERIVED=0"; else E="X=1"; fi
  rm -rf /tmp/ks_$v; env $E CEED_MI355X_ASSEMBLE=serial timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -- python3 $R/bench.py --steps 50 --warmup 5 --cold-idle-s 0 --no-cpu-baseline > $O/ks_$v.log 2>&1
  echo "serial form, $v:"; grep -E "k_fused_pencil|k_assemble" $(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | head -3
done
# reserved CUs for RCCL beside the two-stream split-phase apply (emulated rank 3 of 8)
e() { tag=$1; shift; env "$@" timeout -k 10 200 python3 $R/bench.py --emulate-rank 3 --of 8 $A --steps 100 --warmup 10 --no-cpu-baseline --cold-idle-s 0 > $O/rs_$tag.json 2> $O/rs_$tag.err || { echo "$tag failed"; tail -3 $O/rs_$tag.err; return; }
  python3 -c "
import json; d=[json.loads(l) for l in open('$O/rs_$tag.json') if l.startswith('{')][-1]; e=d['emulated_rank']; print('%-34s %7.1f us/apply %6.2f GDoF/s exchange alone %.1f us' % ('$tag', e['us_per_apply_incl_exchange'], d['value']/1e3, d['config']['halo_exchange_us_alone']))"; }
A=""
e cyl_default
e cyl_mode2 CEED_MI355X_OVL_MODE=2
e cyl_mode2_reserve8 CEED_MI355X_OVL_MODE=2 CEED_MI355X_RESERVE_CUS=8
e cyl_mode2_reserve16 CEED_MI355X_OVL_MODE=2 CEED_MI355X_RESERVE_CUS=16
e cyl_mode2_reserve8_ownstream CEED_MI355X_OVL_MODE=2 CEED_MI355X_RESERVE_CUS=8 CEED_MI355X_COMM_INLINE=0
e cyl_mode2_reserve8_g2 CEED_MI355X_OVL_MODE=2 CEED_MI355X_RESERVE_CUS=8 CEED_MI355X_OVL_G1=2
A="--workload box --degree 6 --nr 64 --nth 64 --nz 64"
e box_default
e box_mode2_reserve16 CEED_MI355X_OVL_MODE=2 CEED_MI355X_RESERVE_CUS=16
rm -rf /tmp/kt_rs; CEED_MI355X_OVL_MODE=2 CEED_MI355X_RESERVE_CUS=8 timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_rs -- python3 $R/bench.py --emulate-rank 3 --of 8 --steps 50 --warmup 5 --no-cpu-baseline --cold-idle-s 0 > $O/rs_trace.json 2> $O/rs_trace.err
python3 $R/tools/apply_timeline.py $(find /tmp/kt_rs -name "*kernel_trace.csv" | head -1) --per-apply 2 --last 40 --json $O/rs_trace_timeline.json > $O/rs_trace_timeline.txt 2>&1; cat $O/rs_trace_timeline.txt
