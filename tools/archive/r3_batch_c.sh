#!/bin/bash
# Round-3 GPU batch C: whole GPU suite, NCCL_GRAPH_MIXING_SUPPORT=0 on the emulated rank, 2-rank gloo rehearsal of the
# replicated-coarse AMG solve on the device, per-level apply times, smoke, the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q --timeout 600 > $O/gputest6.log 2>&1; rc=$?; tail -6 $O/gputest6.log
[ $rc -le 1 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
e() { tag=$1; shift; env "$@" timeout -k 10 200 python3 $R/bench.py --emulate-rank 3 --of 8 $A --steps 100 --warmup 10 --no-cpu-baseline --cold-idle-s 0 > $O/gm_$tag.json 2> $O/gm_$tag.err || { echo "$tag failed"; tail -3 $O/gm_$tag.err; return; }
  python3 -c "
import json; d=[json.loads(l) for l in open('$O/gm_$tag.json') if l.startswith('{')][-1]; e=d['emulated_rank']; print('%-34s %7.1f us/apply %6.2f GDoF/s exchange alone %.1f us' % ('$tag', e['us_per_apply_incl_exchange'], d['value']/1e3, d['config']['halo_exchange_us_alone']))"; }
A=""; e cyl_default; e cyl_nomix NCCL_GRAPH_MIXING_SUPPORT=0; e cyl_default_2; e cyl_nomix_2 NCCL_GRAPH_MIXING_SUPPORT=0
A="--workload box --degree 6 --nr 64 --nth 64 --nz 64"; e box_default; e box_nomix NCCL_GRAPH_MIXING_SUPPORT=0
rm -rf /tmp/kt_nm; NCCL_GRAPH_MIXING_SUPPORT=0 timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_nm -- python3 $R/bench.py --emulate-rank 3 --of 8 --steps 50 --warmup 5 --no-cpu-baseline --cold-idle-s 0 > $O/gm_trace.json 2> $O/gm_trace.err
python3 $R/tools/apply_timeline.py $(find /tmp/kt_nm -name "*kernel_trace.csv" | head -1) --per-apply 1 --last 40 > $O/gm_trace_timeline.txt 2>&1; cat $O/gm_trace_timeline.txt
# multi-rank AMG on the device: two gloo ranks sharing the GPU (rehearsal), against the single-rank run
SOLVE_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/examples/solve_config3.py --coarse amg > $O/solve_c3_2rank_gloo_amg.json 2> $O/solve_c3_2rank_gloo_amg.err; echo "2-rank rc=$?"
timeout -k 10 120 python3 $R/examples/solve_config3.py --coarse amg > $O/solve_c3_1rank_amg_nograph.json 2> $O/solve_c3_1rank_amg_nograph.err
for f in solve_c3_2rank_gloo_amg solve_c3_1rank_amg_nograph; do python3 -c "
import json; d=[json.loads(l) for l in open('$O/$f.json') if l.startswith('{')][-1]; print('$f', {k:d[k] for k in ('ranks','converged','snes_its','ksp_its','snes_solve_s','final_residual_norm')})"; done
python3 $R/tools/level_apply_times.py --cylinder 10,110,90 --degree 4 --problem hyperFS > $O/levels_config4.json 2> $O/levels_config4.txt; cat $O/levels_config4.txt | grep "^#"
python3 $R/tools/level_apply_times.py --mesh $R/tests/golden/mesh_cylinder8_5580e_4ss_us.npz --degree 4 --problem hyperSS > $O/levels_config3.json 2> $O/levels_config3.txt; grep "^#" $O/levels_config3.txt
python3 $R/tools/level_apply_times.py --box 32,32,32 --degree 6 --problem hyperFS > $O/levels_config5.json 2> $O/levels_config5.txt; grep "^#" $O/levels_config5.txt
python3 $R/bench.py > $O/bench_final.json 2> $O/bench_final.err; python3 -c "
import json; d=json.loads(open('$O/bench_final.json').read()); print('bench', d['value'], d['ms_per_step'], d['ms_per_step_cold'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])"
