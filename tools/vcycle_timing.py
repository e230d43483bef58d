"""Wall-clock cost of the V-cycle's pieces on config 3 (5,580-hex cylinder, p=4; levels p=1,2,4)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import read_exodus, hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG

L = cd.CeedLib(cd.PRODUCT_LIB); c = cd.Ceed(L, "/gpu/hip/mi355x")
from ceedpetscsolid_amd.mesh import load_mesh_npz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mesh = load_mesh_npz(os.path.join(ROOT, "tests", "golden", "mesh_cylinder8_5580e_4ss_us.npz")) if "--structured" not in sys.argv \
    else hollow_cylinder_mesh(5, 62, 18, z0=-1.0, z1=1.0)
p = SolidProblem(c, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
s = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}, coarse="chebyshev")
s.bcv.set_array(s.bc_values(0.1)); s.residual(s.U, s.R); s.setup_preconditioner()
top = s.nlev - 1
print("levels", [(lv.degree, p.lsize(i)) for i, lv in enumerate(p.levels)], "nelem", mesh.nelem)


def bench(name, fn, reps=200, graph=False):
    fn(); c.synchronize()
    run = fn
    if graph:
        g = c.capture(fn); run = g.launch
    t0 = time.perf_counter()
    for _ in range(reps): run()
    c.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:45s} {dt * 1e6:9.1f} us" + ("  [graph]" if graph else ""), flush=True)


for lv in range(s.nlev):
    w = s.w[lv]
    bench(f"A(level {lv})", lambda: s.A(lv, w["x"], w["t"]))
    bench(f"axpby(level {lv})", lambda: s.axpby(w["z"], 1.0, w["x"], 0.5))
    bench(f"10 x axpby(level {lv})", lambda: [s.axpby(w["z"], 1.0, w["x"], 0.5) for _ in range(10)], 50)
    bench(f"10 x axpby(level {lv})", lambda: [s.axpby(w["z"], 1.0, w["x"], 0.5) for _ in range(10)], 50, True)
    bench(f"10 x A(level {lv})", lambda: [s.A(lv, w["x"], w["t"]) for _ in range(10)], 50)
    bench(f"10 x A(level {lv})", lambda: [s.A(lv, w["x"], w["t"]) for _ in range(10)], 50, True)
    bench(f"chebyshev(level {lv}, 3 its)", lambda: s.chebyshev(lv, w["b"], w["x"], 3, True), 100)
for lv in range(1, s.nlev):
    bench(f"restrict({lv}->{lv-1})", lambda: p.restrict(lv, s.w[lv]["z"], s.w[lv - 1]["b"]))
    bench(f"prolong({lv-1}->{lv})", lambda: p.prolong(lv, s.w[lv - 1]["x"], s.w[lv]["z"]))
bench("coarse chebyshev(40)", lambda: s.chebyshev(0, s.w[0]["b"], s.w[0]["x"], 40, True, 0.01), 20)
bench("coarse chebyshev(40)", lambda: s.chebyshev(0, s.w[0]["b"], s.w[0]["x"], 40, True, 0.01), 20, True)
bench("vcycle", lambda: s.vcycle(top, s.w[top]["b"], s.kz), 20)
bench("vcycle", lambda: s.vcycle(top, s.w[top]["b"], s.kz), 20, True)
x = s.w[top]["x"]
bench("dot (host-synchronised)", lambda: s.dot(x, x), 100)
for lv in range(s.nlev):
    bench(f"get_diag(level {lv})", lambda: p.get_diag(lv, s.w[lv]["dinv"]), 5)
bench("setup_preconditioner", s.setup_preconditioner, 3)
bench("residual", lambda: s.residual(s.U, s.R), 20)
bench("bc_values (host)", lambda: s.bc_values(0.3), 5)
s2 = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}, coarse="chebyshev", graph=True)
tv = [0.0, 0]; samples = []
orig = s2.precondition
def timed(r, z):
    c.synchronize(); t0 = time.perf_counter(); orig(r, z); c.synchronize(); dt = time.perf_counter() - t0; tv[0] += dt; tv[1] += 1; samples.append((s2.stats.newton_its, dt * 1e6))
s2.precondition = timed
st = s2.solve(10)
print("solve", st.converged, st.seconds, "newton", st.newton_its, "ksp", st.ksp_its, "V-cycle calls", tv[1], "mean V-cycle us", 1e6 * tv[0] / max(tv[1], 1))
print("per-call V-cycle us (newton step, us):", [(a, int(b)) for a, b in samples[:80]])
top2 = s2.nlev - 1
fn = lambda: s2.vcycle(top2, s2.w[top2]["b"], s2.kz)
fn(); g = c.capture(fn)
for gap_us in (0, 50, 200, 1000):
    c.synchronize(); tt = 0.0
    for _ in range(50):
        t0 = time.perf_counter(); g.launch(); c.synchronize(); tt += time.perf_counter() - t0
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < gap_us * 1e-6: pass
    print(f"post-solve state: V-cycle graph, synced each call, idle gap {gap_us:5d} us: {tt / 50 * 1e6:8.1f} us")
bench("post-solve state: vcycle back-to-back", fn, 20, True)
def cpustat():
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(f):
            return {l.split()[0]: int(l.split()[1]) for l in open(f)}
    return {}
import threading
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "threads", threading.active_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
a = cpustat()
s3 = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}, coarse="chebyshev", graph=True)
st = s3.solve(10)
b = cpustat()
print("solve again", st.seconds, {k: b[k] - a[k] for k in b})
print("nthreads in /proc/self/task:", len(os.listdir("/proc/self/task")))
