R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cp $R/ceedpetscsolid_amd/csrc/*.so /tmp/ && cp $R/tools/variants/ph1/*.so $R/ceedpetscsolid_amd/csrc/
CEED_MI355X_ASSEMBLE=serial python3 $R/bench.py --no-cpu-baseline --cold-idle-s 0 --phase-timing $O/phase_c4.txt > $O/phase_c4.json 2> $O/phase_c4.err || tail -5 $O/phase_c4.err
CEED_MI355X_ASSEMBLE=serial python3 $R/bench.py --nz 12 --no-cpu-baseline --cold-idle-s 0 --phase-timing $O/phase_nz12.txt > $O/phase_nz12.json 2> $O/phase_nz12.err || tail -5 $O/phase_nz12.err
CEED_MI355X_ASSEMBLE=serial python3 $R/bench.py --problem hyperSS --no-cpu-baseline --cold-idle-s 0 --phase-timing $O/phase_ss.txt > $O/phase_ss.json 2> $O/phase_ss.err || tail -5 $O/phase_ss.err
cat $O/phase_c4.txt $O/phase_nz12.txt $O/phase_ss.txt
cp /tmp/libceed_mi355x.so /tmp/libsolid_harness_mi355x.so $R/ceedpetscsolid_amd/csrc/
