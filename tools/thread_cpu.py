"""Which threads burn CPU during the config 3 solve (cgroup quota throttling diagnosis)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import load_mesh_npz
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def snapshot():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            fields = f[f.rindex(")") + 2:].split()
            out[t] = (comm, int(fields[11]) + int(fields[12]))      # utime + stime in clock ticks
        except Exception:
            pass
    return out


L = cd.CeedLib(cd.PRODUCT_LIB); c = cd.Ceed(L, "/gpu/hip/mi355x")
mesh = load_mesh_npz(os.path.join(ROOT, "tests", "golden", "mesh_cylinder8_5580e_4ss_us.npz"))
p = SolidProblem(c, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
s = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}, coarse="chebyshev", graph=True)
a = snapshot(); t0 = time.time()
st = s.solve(10)
b = snapshot(); dt = time.time() - t0
tick = os.sysconf("SC_CLK_TCK")
use = sorted(((b[t][1] - a.get(t, (0, 0))[1]) / tick, b[t][0], t) for t in b)
print("solve", st.seconds, "wall", dt, "threads", len(b))
agg = {}
for u, comm, t in use:
    agg.setdefault(comm, [0, 0.0]); agg[comm][0] += 1; agg[comm][1] += u
for comm, (n, u) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {comm:20s} threads {n:3d}  cpu-seconds {u:8.2f}")
print({k: v for k, v in os.environ.items() if "THREADS" in k or "OMP" in k or "BLAS" in k})
