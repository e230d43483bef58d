R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do line=""; for w in 0 7 6 5; do
  CEED_MI355X_PENCIL_WAVES=$w timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --cold-idle-s 0 > $O/pw.json 2> $O/pw.err
  line="$line  [waves/CU $w] $(python3 -c "
import json; d=[json.loads(l) for l in open('$O/pw.json') if l.startswith('{')][-1]; print('%.1f' % (1e3*d['ms_per_step']))")"
done; echo "$line"; done
