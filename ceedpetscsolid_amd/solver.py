"""Newton - CG - p-multigrid solve on top of the operator path (BASELINE config 3).

The reference wires this out of PETSc objects (elasticity.c:386-673): SNES Newton with the
critical-point line search (:596-601) over load increments (:637-673); KSPCG in the natural norm,
rtol 1e-10 (:504-507); PCMG multiplicative V-cycle with 3 Chebyshev(+Jacobi) smoothing steps per
level, eigenvalue bounds (0, 0.1, 0, 1.1) x the estimate (:539-552, :588-590); prolongation /
restriction MatShells (Prolong_Ceed / Restrict_Ceed); a coarse solve by GAMG on the FD-coloured
p=1 matrix (:457-483,:568-585).  PETSc is absent here, so this is this build's own control flow with
the same structure; every vector and operator operation runs on the device through the C ABI
(CeedOperatorApply, CeedOperatorLinearAssembleDiagonal, CeedX vector helpers).  The coarse level is
solved by Jacobi-preconditioned CG on the p=1 operator itself (no assembled matrix, no AMG); because
that makes the preconditioner mildly non-linear the outer Krylov method is the flexible
(Polak-Ribiere) CG, which coincides with CG for a fixed preconditioner.

Iteration counts are therefore this build's own, not PETSc's; the throughput figure reported is the
reference's "DoFs/Sec in SNES" = 1e-6 * Ugsz * (total KSP iterations) / (solve time)
(elasticity.c:755-764).
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import ceed as cd
from .solid import SolidProblem


# ---- boundary functions (src/boundary.c) -------------------------------------------------------
def bc_mms(X: np.ndarray, load: float) -> np.ndarray:
    """BCMMS, boundary.c:31-50."""
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([np.exp(2 * x) * np.sin(3 * y) * np.cos(4 * z), np.exp(3 * y) * np.sin(4 * z) * np.cos(2 * x),
                     np.exp(4 * z) * np.sin(2 * x) * np.cos(3 * y)], axis=1) / 1e8 * load


def bc_clamp(X: np.ndarray, load: float, translate=(0., 0., 0.), axis=(0., 0., 1.), angle_over_pi: float = 0.0) -> np.ndarray:
    """BCClamp, boundary.c:53-74: translation + rotation about `axis` by angle_over_pi*pi, scaled by the load
    fraction.  The x-component's (1-c) term is reproduced AS WRITTEN at boundary.c:69
    (`-ky*ky + kz*kz*x + ...`)."""
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    lx, ly, lz = (t * load for t in translate)
    kx, ky, kz = axis
    th = angle_over_pi * np.pi * load
    c, s = np.cos(th), np.sin(th)
    u0 = lx + s * (-kz * y + ky * z) + (1 - c) * (-ky * ky + kz * kz * x + kx * ky * y + kx * kz * z)
    u1 = ly + s * (kz * x + -kx * z) + (1 - c) * (kx * ky * x - (kx * kx + kz * kz) * y + ky * kz * z)
    u2 = lz + s * (-ky * x + kx * y) + (1 - c) * (kx * kz * x + ky * kz * y - (kx * kx + ky * ky) * z)
    return np.stack([u0, u1, u2], axis=1)


@dataclass
class SolveStats:
    newton_its: int = 0
    ksp_its: int = 0
    coarse_its: int = 0
    jacobian_applies: int = 0
    coarse_spmv: int = 0
    residual_evals: int = 0
    seconds: float = 0.0
    increments: int = 0
    converged: bool = True
    history: list = field(default_factory=list)


class Vec:
    """Thin helpers over CeedVector + the CeedX vector extensions (stand-ins for PETSc Vec calls)."""

    def __init__(self, ceed: cd.Ceed, n: int):
        self.v = ceed.vector(n).set_value(0.0)
        self.n = n


class NewtonPMG:
    def __init__(self, prob: SolidProblem, clamp: Optional[Dict[int, dict]] = None, mms: bool = False, forcing=None,
                 halo=None, rccl="auto", lead_elements: int = 0, smooth_its: int = 3, coarse_rtol: float = 1e-3, coarse_maxit: int = 200,
                 coarse: str = "cg", coarse_cheb_its: int = 40, coarse_cheb_ratio: float = 100.0, graph: bool = False,
                 amg_smooth_its: int = 3, amg_smooth_ratio: float = 10.0, amg_max_coarse_dofs: int = 1500, amg_coarse_cycles: int = 1,
                 ksp_rtol: float = 1e-10, snes_rtol: float = 1e-8, snes_maxit: int = 50, verbose: bool = False,
                 line_search: str = "cp", fuse_epilogue: bool = True):
        """``clamp``: {side_set_id: dict(translate=(..), axis=(..), angle_over_pi=..)} as
        -bc_clamp_<id>_translate / _rotate (cloptions.c:86-131); ids present in the problem's Dirichlet
        set but absent here are held at zero."""
        self.p, self.ceed, self.L = prob, prob.ceed, prob.ceed.L
        self.clamp, self.mms, self.halo = clamp or {}, mms, halo
        # "cp": critical-point secant search, up to three secant steps, a step outside (0, 10] ends the search (this build's default);
        # "cp-petsc": SNESLINESEARCHCP with PETSc's defaults (elasticity.c:596-601 sets the type and nothing else) as recalled from its
        # source, which is NOT in this image (unverified): ONE secant step from (0, 1), downhill slope enforced, direction switched below
        # steptol, lambda kept above maxstep, never a rejection; "full": lambda = 1 (SNESLINESEARCHBASIC)
        if line_search not in ("cp", "cp-petsc", "full"):
            raise ValueError(f"line_search {line_search!r}")
        self.line_search = line_search
        # fuse_epilogue: the smoother's Chebyshev step and the V-cycle's residual formed in the epilogue of the Jacobian apply
        # (CeedXOperatorApplyChebyshev / ApplyResidual: same bits as apply + update); one rank only (no interface sum in between)
        # "auto" (either switch): the forms give the same bits, so the first call of record_preconditioner TIMES the V-cycle in each
        # combination on the problem at hand and keeps the fastest (measured, 99 000 hexes at p = 4: eager + fused 6.1 ms, graph +
        # two passes 6.4-6.6, graph + fused 6.8-7.0, eager + two passes 6.5-6.7; 5 580 hexes: the replayed graph wins by 2x)
        self._auto_fuse = fuse_epilogue == "auto"
        self.fuse_epilogue = True if self._auto_fuse else bool(fuse_epilogue)
        self.tuning = None
        self.smooth_its, self.coarse_rtol, self.coarse_maxit = smooth_its, coarse_rtol, coarse_maxit
        # coarse solver: "cg" (Jacobi-PCG to coarse_rtol: accurate, but two host-synchronised dots per
        # iteration) or "chebyshev" (fixed polynomial over [emax/ratio, 1.1 emax]: no reductions, no host
        # sync, a fixed linear operator; the few lowest modes are left to the outer Krylov method)
        self.coarse, self.coarse_cheb_its, self.coarse_cheb_ratio = coarse, coarse_cheb_its, coarse_cheb_ratio
        # graph=True: record the V-cycle (a fixed sequence of ~150 small launches once the eigenvalue
        # estimates are known) into a hipGraph per Newton step and replay it per Krylov iteration;
        # needs the reduction-free Chebyshev coarse solver
        self._auto_graph = graph == "auto"
        self.graph = False if self._auto_graph else bool(graph)
        if self._auto_graph and coarse not in ("chebyshev", "assembled", "amg"):
            self._auto_graph = False               # (the CG coarse solve cannot be recorded: eager)
        if self.graph and coarse not in ("chebyshev", "assembled", "amg"):
            raise ValueError("graph=True needs coarse='chebyshev', 'assembled' or 'amg' (the CG coarse solve reads dot products on the host)")
        # coarse="assembled": the same Chebyshev polynomial, but on the ASSEMBLED p=1 matrix (assembly.py): a
        # coarse iteration is one SpMV on 81 entries per row instead of a matrix-free apply on the fine quadrature
        # coarse="amg": ONE cycle of a two-level smoothed-aggregation hierarchy on the assembled matrix (amg.py) -- what
        # the reference asks of PCGAMG under KSPPREONLY (elasticity.c:568-585): Chebyshev(amg_smooth_its) on
        # [emax / amg_smooth_ratio, 1.1 emax], the rigid-body-mode coarse correction with an exactly inverted
        # Galerkin matrix, Chebyshev again.  8 matrix products per cycle instead of 40, and 2.3x fewer outer iterations.
        self.asm = self.amg = None
        self.amg_smooth_its, self.amg_smooth_ratio = amg_smooth_its, amg_smooth_ratio
        if coarse in ("assembled", "amg") and len(prob.levels) > 1:
            from .assembly import AssembledLevel
            # several ranks: every rank assembles ITS OWN p = 1 matrix (its elements' sum; the operator is A = sum_r R_r^T A_r R_r, applied as
            # "local product, then the interface sum" like the matrix-free levels); with "amg" the first transfer of the aggregation
            # hierarchy is distributed over the ranks and everything under it is small and replicated (amg.py, round 5; rounds 3-4
            # all-gathered all element matrices and replicated the whole level)
            many = halo is not None and isinstance(halo, (list, tuple)) and halo[-1].world > 1
            self.asm = AssembledLevel(prob, 0)
            if coarse == "amg":
                from .amg import AggregationAMG
                self.amg = AggregationAMG(self.asm, verbose=verbose, max_coarse_dofs=amg_max_coarse_dofs,
                                          smooth_its=amg_smooth_its, smooth_ratio=amg_smooth_ratio, coarse_cycles=amg_coarse_cycles,
                                          dist_halo=halo[0] if many else None)
        self._pc_graph, self._pc_graph_io, self._pc_graph_counts, self._pc_warm = None, None, (0, 0), False
        self.ksp_rtol, self.snes_rtol, self.snes_maxit, self.verbose = ksp_rtol, snes_rtol, snes_maxit, verbose
        self.nlev = len(prob.levels)
        c = self.ceed
        # Several ranks (one element partition per rank, halo.py): `halo` is one HaloExchange per multigrid level.
        # Every vector then aliases a torch tensor (host memory for the CPU oracle, device memory otherwise; the
        # Ceed's stream must be torch's current stream, as in bench.py), so the interface sums of halo.add() act on
        # the operators' outputs in place: after each Jacobian / residual / transfer / diagonal, the L -> G sum of
        # matops.c:57.  Dots are weighted by ownership and all-reduced.
        self.halos = None
        if halo is not None and not isinstance(halo, (list, tuple)):
            if halo.world > 1 and len(prob.levels) > 1:
                raise ValueError("a multi-rank multigrid solve needs one HaloExchange per level")
            halo = [halo]
        if halo is not None and halo[-1].world > 1:
            if len(halo) != len(prob.levels):
                raise ValueError("one HaloExchange per multigrid level expected")
            self.halos = list(halo)
        self.halo = self.halos[-1] if self.halos else None
        # the library's exchange per level (halo.RcclHalo), or None: torch.distributed point-to-point + index ops
        self.rhalos = None
        if self.halos:
            if isinstance(rccl, (list, tuple)):
                self.rhalos = list(rccl)
            elif rccl == "auto":
                import torch.distributed as dist
                if dist.is_initialized() and dist.get_backend(self.halos[-1].group) == "nccl" and self.halos[-1].device.type == "cuda":
                    from .halo import RcclHalo
                    self.rhalos = [RcclHalo(self.ceed, h) for h in self.halos]
            if self.rhalos is not None and len(self.rhalos) != len(prob.levels):
                raise ValueError("one RcclHalo per multigrid level expected")
            if self.graph and self.rhalos is None:
                raise ValueError("graph=True on several ranks needs the library's exchange (torch.distributed calls cannot be recorded)")
            if self.rhalos is None:
                self._auto_graph = False
            if self.rhalos is not None and lead_elements > 0:      # split-phase Jacobians: exchange under the interior elements
                for lv, level in enumerate(prob.levels):
                    level.opJacob.set_overlap_split(int(lead_elements), self.halos[lv].interface_dof_mask())
        self._split = bool(self.rhalos) and lead_elements > 0
        self.w = [{k: self._vec(prob.lsize(lv), lv) for k in ("x", "b", "r", "d", "t", "dinv", "z")} for lv in range(self.nlev)]
        self.emax = [1.0] * self.nlev
        self._x0 = {}
        self._scal = None
        n = prob.lsize()
        top = self.nlev - 1
        self.U, self.R, self.dU, self.Xloc, self.bcv, self.Rtry, self.Utry = (self._vec(n, top) for _ in range(7))
        self.kp, self.kz, self.kAp = (self._vec(n, top) for _ in range(3))
        lvf = prob.levels[prob.fine]
        self.free = (lvf.mask == 0)
        self.weights = [None] * self.nlev
        if self.halos:
            for lv in range(self.nlev):
                wv = self._vec(prob.lsize(lv), lv)
                self._set(wv, self.halos[lv].owner_weight * (prob.levels[lv].mask == 0))
                self.weights[lv] = wv
                self._refresh_multiplicity(lv)
        # body force vector (opSetupForce output, setuplibceed.c:555-583): SNESSolve(snes, F, U) solves
        # residual(U) = load * F on the unconstrained dofs (elasticity.c:645-654)
        self.fv = None
        if forcing is not None:
            self.fv = self._vec(n, top)
            self._set(self.fv, np.asarray(forcing, dtype=np.float64) * self.free)
            self._halo_sum(top, self.fv)
        self.load = 1.0
        self.stats = SolveStats()
        self._bc_nodes = self._collect_bc_nodes()
        if self.amg is not None:
            # setup, not solve: aggregates, prolongation and the product patterns from the Jacobian of the undeformed state
            self.U.set_value(0.0)
            self.p.form_residual(self.U, self.R)
            self.asm.assemble()
            self.amg.build()

    # ---- vector helpers ---------------------------------------------------------------------
    def axpby(self, y, a, x, b):
        self.L.chk(self.L.lib.CeedXVectorAXPBY(y.h, C.c_double(a), x.h, C.c_double(b)))

    def waxpby(self, w, a, x, b, y):
        self.L.chk(self.L.lib.CeedXVectorWAXPBY(w.h, C.c_double(a), x.h, C.c_double(b), y.h))

    def copy(self, dst, src):
        self.axpby(dst, 1.0, src, 0.0)

    def pmult(self, w, x, y):
        self.L.chk(self.L.lib.CeedXVectorPointwiseMult(w.h, x.h, y.h))

    def dot(self, x, y, fine_weight=False, lv=None) -> float:
        """x . y; on several ranks each dof counts once (owner weights of level `lv`, default the fine level)."""
        r = C.c_double()
        lv = self.nlev - 1 if lv is None else lv
        wv = self.weights[lv].h if self.weights[lv] is not None else None
        if self.rhalos:      # the sum over the ranks on the device (ncclAllReduce on the Ceed's stream), ONE read at the end
            if self._scal is None:
                self._scal = self.ceed.vector(8 + 2 * 16)
            lib, chk = self.L.lib, self.L.chk
            chk(lib.CeedXVectorDotTo(x.h, y.h, wv, self._scal.h, 7))
            chk(lib.CeedXCommAllReduce(self.ceed.h, self._scal.h, 7, 1))
            return float(self._scal.to_numpy()[7])
        self.L.chk(self.L.lib.CeedXVectorDot(x.h, y.h, wv, C.byref(r)))
        v = r.value
        if self.halos:
            import torch, torch.distributed as dist
            t = torch.tensor([v], dtype=torch.float64, device="cpu" if self.halos[lv].device.type == "cpu" or self.halos[lv].stage_host else self.halos[lv].device)
            dist.all_reduce(t, group=self.halos[lv].group)
            v = float(t.item())
        return v

    # ---- vectors that alias torch tensors (several ranks only) ----------------------------------------
    def _vec(self, n, lv):
        c = self.ceed
        if not self.halos:
            return c.vector(n).set_value(0.0)
        import torch
        dev = self.halos[lv].device
        v = c.vector(n)
        v.t = torch.zeros(n, dtype=torch.float64, device=dev)
        if dev.type == "cuda":
            v.set_device_pointer(v.t.data_ptr())
        else:
            v.set_array(v.t.numpy(), copy=False)
        return v

    def _set(self, vec, arr):
        """vec := arr (host array), keeping the torch alias intact."""
        if hasattr(vec, "t"):
            import torch
            vec.t.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)))
            self._touched(vec)
        else:
            vec.set_array(arr)

    def _touched(self, vec):
        """The tensor behind a device vector was modified outside the Ceed: drop its host mirror."""
        if vec.t.device.type == "cuda":
            vec.set_device_pointer(vec.t.data_ptr())

    def _halo_sum(self, lv, vec):
        """Interface sum of an operator output at level lv (the DMLocalToGlobal(ADD_VALUES) of matops.c:57)."""
        if self.rhalos:               # the library's exchange, on the Ceed's stream: nothing to synchronise, nothing to re-wrap
            self.rhalos[lv].add(vec)
        elif self.halos:
            if vec.t.device.type == "cuda":
                self.ceed.synchronize()
            self.halos[lv].add(vec.t)
            self._touched(vec)

    def _refresh_multiplicity(self, lv):
        """multVec of misc.c:115-143 counted over ALL ranks: 1 / (halo-summed element multiplicity)."""
        level = self.p.levels[lv]
        m = self._vec(self.p.lsize(lv), lv)
        level.Erestrictu.multiplicity(m)
        self._halo_sum(lv, m)
        self._set(level.multinv, 1.0 / m.to_numpy())

    # ---- operators ---------------------------------------------------------------------------
    def A(self, lv, x, y):
        if lv == 0 and self.asm is not None:
            self.asm.apply(x, y)
            self.stats.coarse_spmv += 1
        elif self._split:             # ApplyLocalCeedOp + DMLocalToGlobal(ADD_VALUES) in one library call, exchange overlapped
            self.p.levels[lv].opJacob.apply_with_halo(x, y, self.rhalos[lv])
            self.stats.jacobian_applies += 1
            return
        else:
            self.p.apply_jacobian(lv, x, y)
            self.stats.jacobian_applies += 1
        self._halo_sum(lv, y)

    def _fused_op(self, lv):
        """The level's Jacobian operator if its consumers may be fused behind it: matrix-free level, no interface sum."""
        if not self.fuse_epilogue or self.halos or self.rhalos or (lv == 0 and self.asm is not None):
            return None
        return self.p.levels[lv].opJacob

    def level_residual(self, lv, b, x, z):
        """z = b - A x on level lv (t: the level's scratch)."""
        w, op = self.w[lv], self._fused_op(lv)
        if op is not None:
            self.L.chk(self.L.lib.CeedXOperatorApplyResidual(op.h, x.h, w["t"].h, b.h, z.h))
            self.stats.jacobian_applies += 1
        else:
            self.A(lv, x, w["t"])
            self.waxpby(z, 1.0, b, -1.0, w["t"])

    def _collect_bc_nodes(self):
        from .mesh import side_set_nodes
        lv = self.p.levels[self.p.fine]
        out = {}
        for sid in self.clamp:
            out[sid] = side_set_nodes(self.p.mesh, lv.dofmap, [sid])
        return out

    def bc_values(self, load: float) -> np.ndarray:
        """DMPlexInsertBoundaryValues at `time = loadIncrement` (matops.c:70-72)."""
        lv = self.p.levels[self.p.fine]
        v = np.zeros((lv.dofmap.nnodes, 3))
        if self.mms:
            nodes = np.nonzero(lv.mask.reshape(-1, 3)[:, 0])[0]
            v[nodes] = bc_mms(lv.dofmap.node_coords[nodes], load)
        for sid, par in self.clamp.items():
            nodes = self._bc_nodes[sid]
            v[nodes] = bc_clamp(lv.dofmap.node_coords[nodes], load, **par)
        return (v.reshape(-1) * (lv.mask != 0))

    def residual(self, U, R):
        """FormResidual_Ceed: Xloc = free part of U + boundary values; R = F(Xloc), constrained rows dropped."""
        self.copy(self.Xloc, U)
        self.axpby(self.Xloc, 1.0, self.bcv, 1.0)
        self.p.form_residual(self.Xloc, R)
        self._halo_sum(self.nlev - 1, R)
        if self.fv is not None:
            self.axpby(R, -self.load, self.fv, 1.0)
        self.stats.residual_evals += 1

    # ---- multigrid preconditioner ---------------------------------------------------------------
    def setup_preconditioner(self):
        """Per Newton step: diagonals (GetDiag_Ceed), Chebyshev eigenvalue estimates."""
        if self.asm is not None:
            self.asm.assemble()
            self.stats.jacobian_applies += self.asm.nd
            if self.amg is not None:
                self.amg.setup()                # Galerkin matrix of THIS Jacobian and its inverse
        for lv in range(self.nlev):
            w = self.w[lv]
            self.p.get_diag(lv, w["dinv"])
            self._halo_sum(lv, w["dinv"])
            # 1 / diagonal; constrained rows come out of the masked operator as zeros and stay zero (CeedVectorReciprocal
            # leaves zeros alone): residuals and corrections are zero there anyway.  No trip through the host.
            w["dinv"].reciprocal()
            if hasattr(w["dinv"], "t"):
                self._touched(w["dinv"])
            mask = self.p.levels[lv].mask != 0
            # The start vector of the eigenvalue estimate is drawn once per level; no BLAS on the host (a threaded BLAS
            # call leaves its worker pool spinning, which starves a CPU-quota'd process for ~0.1 s a call).
            if lv not in self._x0:
                if self.halos:   # shared nodes must get the same value on every rank: a hash of the coordinates
                    X = self.p.levels[lv].dofmap.node_coords
                    k = np.array([[12.9898, 78.233, 37.719], [93.989, 67.345, 24.113], [45.164, 11.135, 83.951]])
                    v = np.sin(X @ k.T) * 437.5453
                    x = (2.0 * (v - np.floor(v)) - 1.0).reshape(-1) * (~mask)
                else:
                    x = np.random.default_rng(1234 + lv).uniform(-1, 1, mask.size) * (~mask)
                x0 = self._vec(mask.size, lv)
                self._set(x0, x / np.sqrt(np.square(x).sum()))     # (any positive scale: the iteration renormalises)
                self._x0[lv] = x0
            # largest eigenvalue of D^-1 A: 10 steps of Jacobi-preconditioned CG on the noisy right-hand side and
            # the largest eigenvalue of its Lanczos tridiagonal -- what KSPChebyshevEstEig does (elasticity.c:546-549).
            # (A plain power iteration from the same vector was 2x low after 12 steps on the config-3 mesh.)
            alphas, betas = self._lanczos_host(lv, 10) if (self.halos and not self.rhalos) else self._lanczos_device(lv, 10)
            k = len(alphas)
            T = np.zeros((max(k, 1), max(k, 1)))
            for j in range(k):
                T[j, j] = 1.0 / alphas[j] + (betas[j - 1] / alphas[j - 1] if j else 0.0)
                if j + 1 < k:
                    T[j, j + 1] = T[j + 1, j] = np.sqrt(max(betas[j], 0.0)) / alphas[j]
            self.emax[lv] = float(np.linalg.eigvalsh(T).max()) if k else 1.0

    def _lanczos_host(self, lv, steps):
        """CG coefficients with every dot product read on the host (several ranks: the dots are all-reduced)."""
        w = self.w[lv]
        r, z, pv, Ap = w["r"], w["z"], w["d"], w["t"]
        self.copy(r, self._x0[lv])
        self.pmult(z, r, w["dinv"]); self.copy(pv, z)
        rz = self.dot(r, z, lv=lv)
        alphas, betas = [], []
        for _ in range(steps):
            self.A(lv, pv, Ap)
            pAp = self.dot(pv, Ap, lv=lv)
            if not (pAp > 0.0 and rz > 0.0):
                break
            alpha = rz / pAp
            self.axpby(r, -alpha, Ap, 1.0)
            self.pmult(z, r, w["dinv"])
            rz_new = self.dot(r, z, lv=lv)
            alphas.append(alpha); betas.append(rz_new / rz)
            self.axpby(pv, 1.0, z, rz_new / rz)
            rz = rz_new
        return alphas, betas

    def _lanczos_device(self, lv, steps):
        """The same recurrence with its scalars kept on the device (CeedXVectorDotTo / CeedXScalarDivide /
        CeedXVectorAXPBYScalars): one read of the 2 * steps coefficients at the end instead of 2 * steps + 1 host round
        trips.  Scalar slots: 0 / 3 rz of the even / odd steps, 1 pAp; 8 + 2 j alpha_j, 9 + 2 j beta_j."""
        w, lib, chk = self.w[lv], self.L.lib, self.L.chk
        r, z, pv, Ap = w["r"], w["z"], w["d"], w["t"]
        if self._scal is None:
            self._scal = self.ceed.vector(8 + 2 * 16)
        sc = self._scal
        sc.set_value(0.0)
        self.copy(r, self._x0[lv])
        self.pmult(z, r, w["dinv"]); self.copy(pv, z)
        one, neg = C.c_double(1.0), C.c_double(-1.0)
        wv = self.weights[lv].h if self.weights[lv] is not None else None      # several ranks: every dof counts once
        many = bool(self.rhalos)

        def dot_to(a, b, slot):
            chk(lib.CeedXVectorDotTo(a.h, b.h, wv, sc.h, slot))
            if many:                 # summed over the ranks where it lies: the scalar never leaves the device
                chk(lib.CeedXCommAllReduce(self.ceed.h, sc.h, slot, 1))
        dot_to(r, z, 0)
        for j in range(steps):
            rz, rz_new, ja, jb = (0, 3, 8 + 2 * j, 9 + 2 * j) if j % 2 == 0 else (3, 0, 8 + 2 * j, 9 + 2 * j)
            self.A(lv, pv, Ap)
            dot_to(pv, Ap, 1)
            chk(lib.CeedXScalarDivide(sc.h, ja, rz, 1, one))                           # alpha_j = rz / pAp (0 on breakdown)
            chk(lib.CeedXVectorAXPBYScalars(r.h, sc.h, ja, neg, Ap.h, -1, one))         # r -= alpha Ap
            self.pmult(z, r, w["dinv"])
            dot_to(r, z, rz_new)
            chk(lib.CeedXScalarDivide(sc.h, jb, rz_new, rz, one))                       # beta_j = rz_new / rz
            chk(lib.CeedXVectorAXPBYScalars(pv.h, sc.h, -1, one, z.h, jb, one))         # p = z + beta p
        v = sc.to_numpy()
        alphas, betas = [], []
        for j in range(steps):
            if not (v[8 + 2 * j] > 0.0) or not np.isfinite(v[9 + 2 * j]):
                break
            alphas.append(float(v[8 + 2 * j])); betas.append(float(v[9 + 2 * j]))
        return alphas, betas

    def chebyshev(self, lv, b, x, its, zero_guess, lmin_frac=0.1):
        """Chebyshev iteration on D^-1 A with bounds [0.1, 1.1] x emax (KSPChebyshevEstEigSet(0,0.1,0,1.1)).  As KSPCHEBYSHEV does, the
        residual is RECOMPUTED from the iterate in every step (r_k = b - A x_k), not carried by a recurrence (r -= A d): the operator is
        applied to x, and the step reads b, dinv, d, x and writes d, x -- no residual vector is read or written (round 5: 48 instead of
        56 B per dof in the step; the two are equal in exact arithmetic)."""
        w = self.w[lv]
        lmin, lmax = lmin_frac * self.emax[lv], 1.1 * self.emax[lv]
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        d, t = w["d"], w["t"]
        step = self.L.lib.CeedXVectorChebyshevStep
        fused = self.L.lib.CeedXOperatorApplyChebyshev
        op = self._fused_op(lv)

        def one(c1, c2, have_x):
            """d = c1 dinv (b - A x) + c2 d;  x (+)= d   (have_x False: x = 0, no apply)"""
            if not have_x:
                self.L.chk(step(x.h, d.h, None, b.h, None, w["dinv"].h, C.c_double(c1), C.c_double(c2), 1))
            elif op is not None:      # the apply and the step in one: A x is consumed where it is formed
                self.L.chk(fused(op.h, x.h, t.h, x.h, d.h, None, b.h, w["dinv"].h, C.c_double(c1), C.c_double(c2), 0))
                self.stats.jacobian_applies += 1
            else:
                self.A(lv, x, t)
                self.L.chk(step(x.h, d.h, None, b.h, t.h, w["dinv"].h, C.c_double(c1), C.c_double(c2), 0))
        one(1.0 / theta, 0.0, not zero_guess)
        for k in range(1, its):
            rho_new = 1.0 / (2.0 * sigma - rho)
            one(2.0 * rho_new / delta, rho_new * rho, True)
            rho = rho_new

    def coarse_solve(self, b, x):
        """Jacobi-preconditioned CG on the coarsest operator (stands in for GAMG on the FD-coloured matrix)."""
        w = self.w[0]
        r, d, t, z = w["r"], w["d"], w["t"], w["z"]
        x.set_value(0.0)
        self.copy(r, b)
        self.pmult(z, r, w["dinv"]); self.copy(d, z)
        rz = self.dot(r, z, lv=0)
        rz0 = rz
        if rz0 <= 0.0:
            return
        for it in range(self.coarse_maxit):
            self.A(0, d, t)
            alpha = rz / self.dot(d, t, lv=0)
            self.axpby(x, alpha, d, 1.0); self.axpby(r, -alpha, t, 1.0)
            self.pmult(z, r, w["dinv"])
            rz_new = self.dot(r, z, lv=0)
            self.stats.coarse_its += 1
            if rz_new <= self.coarse_rtol ** 2 * rz0:
                break
            self.axpby(d, 1.0, z, rz_new / rz)
            rz = rz_new

    def amg_cycle(self, b, x):
        """One V-cycle of the aggregation hierarchy under the assembled level (amg.py): the coarse solve."""
        w, amg = self.w[0], self.amg
        lf = 1.0 / self.amg_smooth_ratio
        self.chebyshev(0, b, x, self.amg_smooth_its, True, lf)
        self.A(0, x, w["t"])
        self.waxpby(w["z"], 1.0, b, -1.0, w["t"])                          # residual
        amg.restrict(w["z"])
        amg.solve_coarsest()
        amg.prolong(w["z"])
        self.axpby(x, 1.0, w["z"], 1.0)
        self.chebyshev(0, b, x, self.amg_smooth_its, False, lf)
        self.stats.coarse_its += 2 * self.amg_smooth_its

    def vcycle(self, lv, b, x):
        """PC_MG_MULTIPLICATIVE V-cycle, 3 smoothing steps down and up (elasticity.c:588-590)."""
        if lv == 0:
            if self.nlev == 1:
                self.chebyshev(0, b, x, self.smooth_its, True)
            elif self.amg is not None:
                self.amg_cycle(b, x)
            elif self.coarse in ("chebyshev", "assembled"):
                self.chebyshev(0, b, x, self.coarse_cheb_its, True, 1.0 / self.coarse_cheb_ratio)
                self.stats.coarse_its += self.coarse_cheb_its
            else:
                self.coarse_solve(b, x)
            return
        w, wc = self.w[lv], self.w[lv - 1]
        self.chebyshev(lv, b, x, self.smooth_its, True)
        self.level_residual(lv, b, x, w["z"])
        self.p.restrict(lv, w["z"], wc["b"])                                # Restrict_Ceed
        self._halo_sum(lv - 1, wc["b"])
        self.vcycle(lv - 1, wc["b"], wc["x"])
        if not self.halos and self.fuse_epilogue:   # one rank: the correction added in place by the prolongation's owners (same bits)
            self.p.prolong_add(lv, wc["x"], x)
        else:
            self.p.prolong(lv, wc["x"], w["z"])                             # Prolong_Ceed
            self._halo_sum(lv, w["z"])
            self.axpby(x, 1.0, w["z"], 1.0)
        self.chebyshev(lv, b, x, self.smooth_its, False)

    def _autotune_vcycle(self, r, z, reps=2):
        """Pick (fused consumers?, replayed graph?) by timing the V-cycle in every combination left open ("auto")."""
        import time
        top = self.nlev - 1
        keep = (self.stats.jacobian_applies, self.stats.coarse_its, self.stats.coarse_spmv)
        fuses = (True, False) if (self._auto_fuse and self._fused_op(top) is not None) else (self.fuse_epilogue,)
        graphs = (True, False) if self._auto_graph else (self.graph,)
        times = {}
        for f in fuses:
            self.fuse_epilogue = f
            self.vcycle(top, r, z)                 # first-use set-up of this form is neither recorded nor timed
            for g in graphs:
                run, gr = (lambda: self.vcycle(top, r, z)), None
                if g:
                    try:
                        gr = self.ceed.capture(run)
                    except cd.CeedError:           # a backend that cannot record (the CPU oracle): eager only
                        continue
                    run = gr.launch
                run(); self.ceed.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    run()
                self.ceed.synchronize()
                times[(f, g)] = (time.perf_counter() - t0) / reps
                if gr is not None:
                    gr.destroy()
        (self.fuse_epilogue, self.graph), _ = min(times.items(), key=lambda kv: kv[1])
        self._auto_fuse = self._auto_graph = False
        self._pc_warm = True
        self.stats.jacobian_applies, self.stats.coarse_its, self.stats.coarse_spmv = keep
        self.tuning = {"fused_epilogue": self.fuse_epilogue, "vcycle_graph": self.graph,
                       "vcycle_ms": {f"{'fused' if f else 'two_pass'}+{'graph' if g else 'eager'}": 1e3 * t for (f, g), t in times.items()}}
        if self.verbose:
            print("V-cycle forms (ms):", self.tuning["vcycle_ms"], "->", "fused" if self.fuse_epilogue else "two passes", "+", "graph" if self.graph else "eager")

    def record_preconditioner(self, r, z):
        """Capture vcycle(r -> z) for the current diagonals / eigenvalue bounds (call after setup_preconditioner)."""
        if self._pc_graph is not None:
            self._pc_graph.destroy()
            self._pc_graph = None
        if (self._auto_fuse or self._auto_graph) and self.nlev > 1:
            self._autotune_vcycle(r, z)
        if not self.graph or self.nlev == 1:
            return
        top = self.nlev - 1
        if not self._pc_warm:                      # first-use setup (CSR maps, scratch) must not be recorded
            self.vcycle(top, r, z)
            self._pc_warm = True
        j0, c0 = self.stats.jacobian_applies, self.stats.coarse_its
        self._pc_graph = self.ceed.capture(lambda: self.vcycle(top, r, z))
        self._pc_graph_counts = (self.stats.jacobian_applies - j0, self.stats.coarse_its - c0)
        self.stats.jacobian_applies, self.stats.coarse_its = j0, c0
        self._pc_graph_io = (r, z)

    def precondition(self, r, z):
        if self._pc_graph is not None and self._pc_graph_io[0] is r and self._pc_graph_io[1] is z and self._pc_graph.stale():
            # a vector the recording depends on was overwritten behind it (CeedXGraphIsStale; the check is LOCAL to this rank): the
            # recording is dropped and this rank runs the V-cycle eagerly -- the same launches and, on several ranks, the same RCCL
            # exchanges in the same order as its peers' replays, so nobody waits for a message that never comes
            self._pc_graph.destroy()
            self._pc_graph = None
        if self._pc_graph is not None and self._pc_graph_io[0] is r and self._pc_graph_io[1] is z:
            self._pc_graph.launch()
            self.stats.jacobian_applies += self._pc_graph_counts[0]
            self.stats.coarse_its += self._pc_graph_counts[1]
            return
        if self.nlev == 1:          # -multigrid none: Jacobi (elasticity.c:516-519)
            self.pmult(z, r, self.w[0]["dinv"])
        else:
            self.vcycle(self.nlev - 1, r, z)

    # ---- Krylov ---------------------------------------------------------------------------------
    def fcg(self, b, x, rtol):
        """Flexible preconditioned CG, natural-norm convergence test (KSP_NORM_NATURAL: sqrt(r'z))."""
        fine = self.nlev - 1
        r, z, p, Ap = self.w[fine]["b"], self.kz, self.kp, self.kAp
        x.set_value(0.0)
        self.copy(r, b)
        self.precondition(r, z)
        self.copy(p, z)
        rz = self.dot(r, z, True)
        rz0 = rz
        its = 0
        if rz0 <= 0.0:
            return 0
        for its in range(1, 500):
            self.A(fine, p, Ap)
            alpha = rz / self.dot(p, Ap, True)
            self.axpby(x, alpha, p, 1.0); self.axpby(r, -alpha, Ap, 1.0)
            r_zold = self.dot(r, z, True)                    # r_new . z_old, taken BEFORE the preconditioner overwrites z (no copy of z)
            self.precondition(r, z)
            rz_new = self.dot(r, z, True)
            if self.verbose:
                print(f"      ksp {its:3d}  natural norm {np.sqrt(abs(rz_new)):.3e}")
            if rz_new <= rtol ** 2 * rz0:
                break
            beta = (rz_new - r_zold) / rz                    # Polak-Ribiere: flexible
            self.axpby(p, 1.0, z, beta)
            rz = rz_new
        return its

    # ---- Newton with load increments ---------------------------------------------------------------
    def solve(self, num_increments: int = 10, stop_after: Optional[int] = None) -> SolveStats:
        """``stop_after``: run only the first so many of the ``num_increments`` load increments (load studies)."""
        st = self.stats
        t0 = time.perf_counter()
        self.U.set_value(0.0)
        for inc in range(1, (stop_after or num_increments) + 1):
            load = inc / num_increments
            self.load = load
            self._set(self.bcv, self.bc_values(load))
            self.residual(self.U, self.R)
            rnorm0 = np.sqrt(self.dot(self.R, self.R, True))
            rnorm = rnorm0
            if self.verbose:
                print(f"increment {inc}/{num_increments}: |R| = {rnorm0:.6e}")
            for it in range(self.snes_maxit):
                if rnorm <= self.snes_rtol * rnorm0 or rnorm < 1e-50:
                    break
                self.setup_preconditioner()
                self.record_preconditioner(self.w[self.nlev - 1]["b"], self.kz)
                self.axpby(self.Rtry, -1.0, self.R, 0.0)                 # rhs = -R
                k = self.fcg(self.Rtry, self.dU, self.ksp_rtol)
                st.ksp_its += k
                # critical-point line search (SNESLINESEARCHCP): secant on phi(l) = dU . R(U + l dU)
                lam, lam_old = 1.0, 0.0
                phi_old = self.dot(self.dU, self.R, True)
                for _ in range(0 if self.line_search == "full" else (1 if self.line_search == "cp-petsc" else 3)):
                    self.copy(self.Utry, self.U); self.axpby(self.Utry, lam, self.dU, 1.0)
                    self.residual(self.Utry, self.Rtry)
                    phi = self.dot(self.dU, self.Rtry, True)
                    if abs(phi) <= 1e-8 * abs(phi_old) or abs(phi - phi_old) < 1e-300:
                        break
                    if self.line_search == "cp-petsc":
                        # ONE secant step in the form of PETSc's SNESLineSearchApply_CP (max_its 1, linear order) AS RECALLED -- PETSc's
                        # source is not in this image, so this is unverified against it: with fty = Y . F(X - lambda Y)
                        # (Y = -dU, so fty = -phi) the slope s = d fty / d lambda is made negative ("always go downhill"), the update is
                        # lambda - fty / s, a result below steptol (1e-12) switches direction (lambda + fty / s), one above maxstep (1e8)
                        # or a NaN leaves lambda as it was; the step is never rejected.  (Round 4 clamped the plain secant result to
                        # [1e-12, 1e8] instead: a negative secant step then took lambda = 1e-12, i.e. no step.)
                        fty, fty_old = -phi, -phi_old
                        sl = (fty - fty_old) / (lam - lam_old)
                        if sl > 0.0:
                            sl = -sl
                        if sl != 0.0:
                            lam_up = lam - fty / sl
                            if lam_up < 1e-12:
                                lam_up = lam + fty / sl
                            if np.isfinite(lam_up) and lam_up <= 1e8:
                                lam = lam_up
                        break
                    lam_new = lam - phi * (lam - lam_old) / (phi - phi_old)
                    if not np.isfinite(lam_new) or abs(lam_new - lam) < 1e-8 or lam_new <= 0.0 or lam_new > 10.0:
                        break
                    lam_old, phi_old, lam = lam, phi, lam_new
                self.axpby(self.U, lam, self.dU, 1.0)
                self.residual(self.U, self.R)                              # also refreshes the stored state
                rnorm = np.sqrt(self.dot(self.R, self.R, True))
                st.newton_its += 1
                st.history.append((inc, it + 1, k, lam, rnorm))
                if self.verbose:
                    print(f"   newton {it + 1:2d}: ksp its {k:3d}  lambda {lam:.4f}  |R| = {rnorm:.6e}")
            # decided on the residual itself: a step that meets the tolerance in the LAST allowed iteration has converged
            if not (rnorm <= self.snes_rtol * rnorm0 or rnorm < 1e-50):
                st.converged = False
            st.increments = inc
            if not np.isfinite(rnorm) or not st.converged:
                st.converged = False
                break
        self.ceed.synchronize()
        st.seconds = time.perf_counter() - t0
        if self._pc_graph is not None:
            self._pc_graph.destroy()
            self._pc_graph = None
        return st
