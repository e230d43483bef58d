"""Diagnostic only: per-phase share of a wave's lifetime in the fused kernel, from the
-DCPS_STAMPS build (tools/libceed_mi355x_stamps.so).  Never quote this build's run time."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem
lib = cd.CeedLib(os.environ["CEEDPETSCSOLID_MI355X_LIB"])
ceed = cd.Ceed(lib, "/gpu/hip/mi355x")
mesh = hollow_cylinder_mesh(10, 110, 90)
p = SolidProblem(ceed, mesh, 4, "hyperFS", bc_sides=[998, 999], multigrid="none")
n = p.lsize(); X, Y = ceed.vector(n), ceed.vector(n)
X.set_array(p.smooth_state(0.1)); p.form_residual(X, Y)
X.set_array(np.random.default_rng(0).uniform(-1, 1, n))
op = p.levels[p.fine].opJacob
for _ in range(3): p.apply_jacobian(p.fine, X, Y)
st = torch.zeros(mesh.nelem * 8, dtype=torch.int64, device="cuda")
lib.lib.CeedXOperatorSetStampBuffer(op.h, C.c_void_p(st.data_ptr()))
p.apply_jacobian(p.fine, X, Y); ceed.synchronize(); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(-1, 8)[:, :6].astype(np.float64)
s = s[s[:, 0] > 0]
d = np.diff(s, axis=1)
names = ["top: next offsets + gather->LDS", "interp (3 passes)", "grad+physics (all slots)", "grad^T", "interp^T+atomics issue"]
tot = s[:, 5] - s[:, 0]
print("waves", len(s), "median lifetime (s_memtime ticks)", np.median(tot))
for i, nm in enumerate(names):
    print(f"  {nm:28s} median {np.median(d[:, i]):9.0f}  share {100*np.median(d[:, i])/np.median(tot):5.1f} %")
print("kernel span ticks", s[:, 5].max() - s[:, 0].min())
