#!/bin/bash
# Round-3 GPU batch A: CU-mask microbenchmark, AMG / solver GPU tests after the dense SpGEMM, solver timing records, the
# committed profile sets of config 4 (r03) and of config 5's one-GPU block (r03_config5), affine A/B, whole 64^3 box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$R/tools/microbench/cu_mask > $O/cu_mask.txt 2>&1; cat $O/cu_mask.txt
cd $R && python -m pytest tests/test_amg.py tests/test_solver.py -m gpu -q --timeout 600 > $O/gputest_amg_solver.log 2>&1; tail -3 $O/gputest_amg_solver.log; cd /tmp
one() { tag=$1; shift; timeout -k 10 $TMO python3 $R/examples/solve_config3.py "$@" > $O/solve_$tag.json 2> $O/solve_$tag.err; echo "$tag rc=$?"; python3 -c "
import json; d=[json.loads(l) for l in open('$O/solve_$tag.json') if l.startswith('{')][-1]; print({k:d[k] for k in ('problem','converged','load_increments','snes_its','ksp_its','snes_solve_s','final_residual_norm')})"; }
TMO=120 one c3_pinned_amg_b --coarse amg --graph
TMO=120 one c3_readme_load_linElas --coarse amg --graph --translate 0,-0.5,1 --problem linElas --E 1e6
TMO=400 one cyl44928_p4_amg_b --coarse amg --graph --mesh $R/tests/golden/mesh_cylinder8_44928e_2ss_us.npz --translate 0,-0.02,0.05
C=$(cd $R && cat .commit_id 2>/dev/null || echo unknown)
bash $R/tools/refresh_profiles.sh r03 $C "k_fused_pencil<5, 5, 6" || exit 1
bash $R/tools/refresh_profiles.sh r03_config5 $C "k_fused_pencil<7, 7, 6" --workload box --degree 6 --nr 32 --nth 32 --nz 32 || exit 1
CEED_MI355X_AFFINE=0 python3 $R/bench.py --workload box --degree 6 --nr 32 --nth 32 --nz 32 --no-cpu-baseline --cold-idle-s 0 > $O/config5_block_general_geo.json 2>$O/config5_block_general_geo.err
python3 $R/bench.py --workload box --degree 6 --nr 32 --nth 32 --nz 32 --no-cpu-baseline --cold-idle-s 0 > $O/config5_block_affine.json 2>$O/config5_block_affine.err
python3 $R/bench.py --workload box --degree 6 --nr 64 --nth 64 --nz 64 --no-cpu-baseline --cold-idle-s 0 --steps 20 > $O/config5_whole_affine.json 2>$O/config5_whole_affine.err
CEED_MI355X_AFFINE=0 python3 $R/bench.py --workload box --degree 6 --nr 64 --nth 64 --nz 64 --no-cpu-baseline --cold-idle-s 0 --steps 20 > $O/config5_whole_general_geo.json 2>$O/config5_whole_general_geo.err
for f in config5_block_general_geo config5_block_affine config5_whole_general_geo config5_whole_affine; do python3 -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['value']/1e3,2),'GDoF/s', round(d['ms_per_step'],4),'ms frac', round(d['roofline']['frac'],3), d['config']['kernel'])"; done
