#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O; cd $R
timeout -k 10 600 python -u -m pytest tests/test_amg.py tests/test_solver.py -m gpu -x -q --timeout 400 > $O/amg_tests5.log 2>&1 || { tail -15 $O/amg_tests5.log; exit 1; }
tail -1 $O/amg_tests5.log
timeout -k 10 300 python -u examples/solve_config3.py --coarse amg --graph 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config 3', {k: d[k] for k in ('converged','snes_its','ksp_its','setup_s','snes_solve_s')})"
M=tests/golden/mesh_cylinder8_44928e_2ss_us.npz
timeout -k 10 500 python -u examples/solve_config3.py --mesh $M --degree 2 --coarse amg --graph --increments 10 --translate 0,-0.02,0.05 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('44928e p2', {k: d[k] for k in ('converged','snes_its','ksp_its','setup_s','snes_solve_s')})"
