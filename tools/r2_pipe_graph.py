"""Full-size check (GPU): a recorded pipelined apply (config 4: 3 segments on 2 streams) replays bitwise like the eager
pipelined and the eager serial apply, on changing inputs."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem
lib = cd.CeedLib(cd.PRODUCT_LIB)
c1 = cd.Ceed(lib, "/gpu/hip/mi355x")
os.environ["CEED_MI355X_ASSEMBLE"] = "serial"; c2 = cd.Ceed(lib, "/gpu/hip/mi355x"); os.environ.pop("CEED_MI355X_ASSEMBLE")
mesh = hollow_cylinder_mesh(10, 110, 90)
ps = [SolidProblem(c, mesh, 4, "hyperFS", nu=0.3, E=1.0, bc_sides=[998, 999], multigrid="none") for c in (c1, c2)]
n = ps[0].lsize()
vs = [(c.vector(n), c.vector(n)) for c in (c1, c2)]
u0 = ps[0].smooth_state(0.1)
for (X, Y), p in zip(vs, ps):
    X.set_array(u0); p.form_residual(X, Y)
assert np.array_equal(vs[0][1].to_numpy(), vs[1][1].to_numpy()), "residual"
rng = np.random.default_rng(3)
Yg = c1.vector(n)
X1, Y1 = vs[0]
X1.set_array(rng.uniform(-1, 1, n)); X1.device_pointer(); ps[0].apply_jacobian(ps[0].fine, X1, Y1)
g = c1.capture(lambda: ps[0].apply_jacobian(ps[0].fine, X1, Yg))
print("launch info:", ps[0].levels[ps[0].fine].opJacob.launch_info(), ps[1].levels[ps[1].fine].opJacob.launch_info())
for it in range(5):
    x = rng.uniform(-1, 1, n)
    for (X, Y), p in zip(vs, ps):
        X.set_array(x); X.device_pointer(); p.apply_jacobian(p.fine, X, Y)
    Yg.set_value(-5.0); g.launch(); g.launch()
    a, b, cgr = vs[0][1].to_numpy(), vs[1][1].to_numpy(), Yg.to_numpy()
    assert np.array_equal(a, b), f"eager pipelined vs serial, apply {it}"
    assert np.array_equal(cgr, b), f"replayed pipelined vs serial, apply {it}"
X2, Y2 = vs[1]
Yg2 = c2.vector(n)
g2 = c2.capture(lambda: ps[1].apply_jacobian(ps[1].fine, X2, Yg2))
def eager(p, X, Y, c):
    for _ in range(20): p.apply_jacobian(p.fine, X, Y)
    c.synchronize(); t = time.perf_counter()
    for _ in range(100): p.apply_jacobian(p.fine, X, Y)
    c.synchronize(); return (time.perf_counter() - t) / 100 * 1e3
def replay(gr, c):
    for _ in range(20): gr.launch()
    c.synchronize(); t = time.perf_counter()
    for _ in range(100): gr.launch()
    c.synchronize(); return (time.perf_counter() - t) / 100 * 1e3
for rep in range(2):
    print("ms per apply: eager pipelined %.4f  eager serial %.4f  replayed pipelined %.4f  replayed serial %.4f" %
          (eager(ps[0], X1, Y1, c1), eager(ps[1], X2, Y2, c2), replay(g, c1), replay(g2, c2)))
print("ok: recorded pipelined apply == eager pipelined == eager serial, bitwise, 5 inputs")
