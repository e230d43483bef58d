R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
one() { tag=$1; shift; timeout -k 10 $TMO python3 $R/examples/solve_config3.py "$@" > $O/solve_$tag.json 2> $O/solve_$tag.err; echo "$tag rc=$?"; tail -c 600 $O/solve_$tag.json; echo; }
TMO=120 one c3_pinned_amg --coarse amg --graph
TMO=600 one cyl99000_p4_hyperFS_amg --coarse amg --graph --cylinder 10,110,90 --problem hyperFS --translate 0,-0.02,0.05
{ python3 $R/tools/level_apply_times.py --cylinder 10,110,90 --degree 4 --problem hyperFS; python3 $R/tools/level_apply_times.py --mesh $R/tests/golden/mesh_cylinder8_5580e_4ss_us.npz --degree 4 --problem hyperSS; python3 $R/tools/level_apply_times.py --box 32,32,32 --degree 6 --problem hyperFS; } 2>&1 | grep "^#" > $O/level_apply_times.txt; cat $O/level_apply_times.txt
for p in hyperSS linElas; do python3 $R/bench.py --problem $p --no-cpu-baseline --cold-idle-s 0 > $O/fin_$p.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/fin_$p.json').read()); print('$p', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
