#!/bin/bash
# Round 4: the records of the final build, one GPU call each part.   usage: r4_final_records.sh <part: a|b> <commit>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
part=$1; commit=${2:-unknown}
if [ "$part" = a ]; then
  # (the capture_fork_repro reproducer deliberately crashes a GPU child: cause recorded in profiles/r04_rccl_capture_probe.txt; run by hand only when the HIP runtime changes -- ADVICE r4)
  python3 tools/rccl_capture_probe.py > $O/rccl_capture_probe_final.txt 2>&1; head -3 $O/rccl_capture_probe_final.txt | cut -c1-300
  bash tools/refresh_profiles.sh r04 $commit "k_fused_pencil<5, 5, 6" 2>&1 | tail -12
fi
if [ "$part" = b ]; then
  cd /tmp; export TMPDIR=/tmp
  python3 $R/bench.py --workload box --degree 6 --nr 32 --nth 32 --nz 32 --no-cpu-baseline > $O/r04_config5_bench.json 2> $O/r04_config5_bench.err || tail -3 $O/r04_config5_bench.err
  python3 $R/bench.py --workload box --degree 6 --nr 64 --nth 64 --nz 64 --steps 20 --no-cpu-baseline > $O/r04_config5_whole_box_bench.json 2> $O/r04_config5_whole_box_bench.err || tail -3 $O/r04_config5_whole_box_bench.err
  python3 $R/bench.py --workload mesh --no-cpu-baseline > $O/r04_mesh44928_bench.json 2> $O/r04_mesh44928_bench.err || tail -3 $O/r04_mesh44928_bench.err
  for pr in hyperSS linElas; do python3 $R/bench.py --problem $pr --no-cpu-baseline > $O/r04_${pr}_bench.json 2> $O/r04_${pr}_bench.err || tail -3 $O/r04_${pr}_bench.err; done
  python3 - <<PY
import json
for n in ("r04_config5_bench", "r04_config5_whole_box_bench", "r04_mesh44928_bench", "r04_hyperSS_bench", "r04_linElas_bench"):
    try:
        d = [json.loads(l) for l in open("$O/" + n + ".json") if l.startswith("{")][-1]
        print(n, "%.1f GDoF/s  %.4f ms  frac %.3f  %s" % (d["value"] / 1e3, d["ms_per_step"], d["roofline"]["frac"], d["config"]["assembly"][:40]))
    except Exception as e:
        print(n, "ERR", e)
PY
  cd $R
  bash tools/r3_rank_of_8.sh r04_rank_of_8 2>&1 | tail -40; cp gpurun_out/r3/r04_rank_of_8_* $O/ 2>/dev/null
  BENCH_DIST_BACKEND=gloo timeout -k 10 280 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_two_ranks_gloo_one_gpu.json 2> $O/r04_two_ranks_gloo_one_gpu.err; echo "2-rank gloo rehearsal rc=$?"; cut -c1-400 $O/r04_two_ranks_gloo_one_gpu.json
fi
