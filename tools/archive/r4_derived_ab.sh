R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
run() { w=$1; shift; env "$@" timeout -k 10 250 python3 $R/bench.py $w --no-cpu-baseline --cold-idle-s 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end='')"; }
for rep in 1 2; do
  echo "rep $rep: block derived $(run "--workload box --degree 6 --nr 32 --nth 32 --nz 32" A=1)  plain $(run "--workload box --degree 6 --nr 32 --nth 32 --nz 32" CEED_MI355X_DERIVED=0) | whole box derived $(run "--workload box --degree 6 --nr 64 --nth 64 --nz 64 --steps 20" A=1)  plain $(run "--workload box --degree 6 --nr 64 --nth 64 --nz 64 --steps 20" CEED_MI355X_DERIVED=0) | p=5 derived $(run "--workload box --degree 5 --nr 36 --nth 36 --nz 36" A=1) plain $(run "--workload box --degree 5 --nr 36 --nth 36 --nz 36" CEED_MI355X_DERIVED=0)"
done
