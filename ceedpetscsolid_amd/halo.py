"""Interface-dof halo sum between element partitions: the multi-GPU exchange step.

Replaces the reference's ``DMLocalToGlobal(ADD_VALUES)`` + ``DMGlobalToLocal(INSERT_VALUES)``
pair around every operator apply (src/matops.c:33,57; PetscSF over MPI) by ONE neighbour
exchange: every rank keeps its closure dofs (owned + shared) in its L-vector, shared
entries are replicated and kept consistent, so after the local scatter-add each rank adds
its neighbours' partial sums on the shared entries and the next gather needs no
broadcast.  Transport is ``torch.distributed`` point-to-point
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests); payloads are
small (<= ~0.5 MB per neighbour at p=4, 99k elements per GPU), i.e. latency-bound.

Shared nodes are discovered from partition-independent topological keys
(``DofMap.node_keys``, built from global vertex ids), so no rank needs the global mesh.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .mesh import DofMap, HexMesh, boundary_nodes, key_bytes


@dataclass
class Neighbour:
    rank: int
    dof_idx: torch.Tensor      # int64 indices into the L-vector, same order on both sides
    send: torch.Tensor
    recv: torch.Tensor


class HaloExchange:
    def __init__(self, mesh: HexMesh, dm: DofMap, device="cpu", group=None, virtual=None):
        """`virtual` (emulation of ONE rank of a larger job on a single process, bench.py --emulate-rank): a dict
        {"rank": K, "world": N, "keys": {r: key bytes of rank r's nodes near the cut}} standing in for the all-gather."""
        self.group = group
        self.virtual = virtual
        if virtual is not None:
            self.rank, self.world = int(virtual["rank"]), int(virtual["world"])
        else:
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device(device)
        # gloo cannot move device tensors point-to-point: stage through the host (CPU tests and
        # single-GPU rehearsals only; the GPU node uses "nccl" = RCCL, device buffers end to end)
        self.stage_host = (virtual is None and dist.is_initialized() and dist.get_backend(group) == "gloo" and self.device.type == "cuda")
        self.dm = dm
        self.neigh: List[Neighbour] = []
        self.owner_weight = np.ones(dm.lsize)   # 1 on dofs this rank owns (lowest sharing rank)
        if self.world == 1:
            return
        cand = boundary_nodes(mesh, dm)
        kb = key_bytes(dm.node_keys[cand])
        if virtual is not None:
            gathered = [None] * self.world
            for r in range(self.world):
                okb = virtual["keys"].get(r)
                gathered[r] = {"keys": okb.tobytes(), "n": int(okb.size), "w": int(okb.dtype.itemsize)} if okb is not None and r != self.rank else {"keys": b"", "n": 0, "w": int(kb.dtype.itemsize)}
        else:
            mine = {"keys": kb.tobytes(), "n": int(cand.size), "w": int(kb.dtype.itemsize)}
            gathered: List[Optional[dict]] = [None] * self.world
            dist.all_gather_object(gathered, mine, group=group)
        order = np.argsort(kb, kind="stable")
        kb_sorted, cand_sorted = kb[order], cand[order]
        nc = dm.ncomp
        for r, other in enumerate(gathered):
            if r == self.rank or other["n"] == 0:
                continue
            okb = np.frombuffer(other["keys"], dtype=kb.dtype)
            common = np.intersect1d(kb_sorted, okb, assume_unique=True)   # sorted by key: same order on both ranks
            if common.size == 0:
                continue
            pos = np.searchsorted(kb_sorted, common)
            nodes = cand_sorted[pos].astype(np.int64)
            dofs = (nodes[:, None] * nc + np.arange(nc)[None, :]).ravel()
            idx = torch.from_numpy(dofs).to(self.device)
            bdev = torch.device("cpu") if self.stage_host else self.device
            buf = lambda: torch.zeros(dofs.size, dtype=torch.float64, device=bdev)
            self.neigh.append(Neighbour(r, idx, buf(), buf()))
            if r < self.rank:
                self.owner_weight[dofs] = 0.0

    @property
    def n_shared_dofs(self) -> int:
        return int(sum(n.dof_idx.numel() for n in self.neigh))

    def add(self, y: torch.Tensor):
        """y[shared] += sum over neighbours of their y[shared]  (in place; all ranks call)."""
        if not self.neigh:
            return
        ops = []
        for n in self.neigh:
            if self.stage_host:
                n.send.copy_(torch.index_select(y, 0, n.dof_idx))
            else:
                torch.index_select(y, 0, n.dof_idx, out=n.send)
        for n in self.neigh:
            ops.append(dist.P2POp(dist.isend, n.send, n.rank, group=self.group))
            ops.append(dist.P2POp(dist.irecv, n.recv, n.rank, group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for n in self.neigh:
            y.index_add_(0, n.dof_idx, n.recv.to(y.device) if self.stage_host else n.recv)

    # -- split form for communication overlap: start() after the interface nodes are complete
    # (CeedXOperatorApplyPhase 0), finish() after the interior work has been queued -----------------
    def start(self, y: torch.Tensor):
        self._works = []
        if not self.neigh:
            return
        ops = []
        for n in self.neigh:
            if self.stage_host:
                n.send.copy_(torch.index_select(y, 0, n.dof_idx))
            else:
                torch.index_select(y, 0, n.dof_idx, out=n.send)
        for n in self.neigh:
            ops.append(dist.P2POp(dist.isend, n.send, n.rank, group=self.group))
            ops.append(dist.P2POp(dist.irecv, n.recv, n.rank, group=self.group))
        self._works = dist.batch_isend_irecv(ops)

    def finish(self, y: torch.Tensor):
        for w in getattr(self, "_works", []):
            w.wait()
        for n in self.neigh:
            y.index_add_(0, n.dof_idx, n.recv.to(y.device) if self.stage_host else n.recv)
        self._works = []

    def interface_dof_mask(self) -> np.ndarray:
        """uint8 mask over the L-vector: 1 on every dof shared with another rank."""
        m = np.zeros(self.dm.lsize, dtype=np.uint8)
        for n in self.neigh:
            m[n.dof_idx.cpu().numpy()] = 1
        return m

    def global_count(self, local_mask_free: np.ndarray) -> int:
        """Number of distinct unconstrained dofs over all ranks (the reference's Ugsz)."""
        mine = float((local_mask_free * self.owner_weight).sum())
        if self.world == 1 or self.virtual is not None:   # (emulated rank: the dofs this rank OWNS)
            return int(round(mine))
        t = torch.tensor([mine], dtype=torch.float64, device="cpu" if self.stage_host else self.device)
        dist.all_reduce(t, group=self.group)
        return int(round(t.item()))

    def dot(self, a: torch.Tensor, b: torch.Tensor, weight: torch.Tensor) -> float:
        """Global dot product of two consistent L-layout vectors (owner-weighted)."""
        s = (a * b * weight).sum().reshape(1)
        if self.world > 1:
            dist.all_reduce(s, group=self.group)
        return float(s.item())


class RcclHalo:
    """The same exchange behind the C ABI (include/ceed.h, CeedXHalo*): RCCL sends / receives issued by the library on a
    stream of its own, hand-written pack and unpack-add kernels, no framework ops on the path.  Built from a
    HaloExchange's neighbour lists; the communicator is bootstrapped like any NCCL communicator (rank 0's unique id,
    distributed here with torch.distributed -- an MPI_Bcast in the reference's world)."""

    @staticmethod
    def unique_id(ceed, halo: "HaloExchange"):
        """Rank 0's communicator id on every rank (collective over the torch group; MPI_Bcast in the reference's world).
        Call it from the thread that owns the process group: the worker thread of checked_rccl_halo must not."""
        import ctypes as C
        ident = [None]
        if halo.rank == 0:
            try:
                buf = C.create_string_buffer(128)
                ceed.L.chk(ceed.L.lib.CeedXCommGetUniqueId(ceed.h, buf))
                ident = [buf.raw]
            except Exception as e:   # noqa: BLE001 -- the other ranks are waiting in the broadcast: tell them
                ident = [RuntimeError(f"CeedXCommGetUniqueId failed on rank 0: {e!r}")]
        dist.broadcast_object_list(ident, src=0, group=halo.group)
        if isinstance(ident[0], Exception):
            raise ident[0]            # on EVERY rank
        return ident[0]

    def __init__(self, ceed, halo: "HaloExchange", emulate_self: bool = False, ident: Optional[bytes] = None):
        """emulate_self (bench.py --emulate-rank): a ONE-rank communicator; every neighbour list is sent to and received
        from this rank itself -- the launches, buffer sizes and stream hand-overs of the real exchange on a single GPU (the
        sums it produces are those of a vector folded onto itself, not of the partitioned problem).
        ident: the communicator id from unique_id() (taken here, collectively, if not given)."""
        import ctypes as C
        self.ceed, self.L = ceed, ceed.L
        self.h = C.c_void_p()
        self.world = halo.world
        lib = self.L.lib
        if emulate_self:
            if not getattr(ceed, "_comm_ready", False):
                buf = C.create_string_buffer(128)
                self.L.chk(lib.CeedXCommGetUniqueId(ceed.h, buf))
                self.L.chk(lib.CeedXCommInit(ceed.h, C.c_int(1), C.c_int(0), C.c_char_p(buf.raw)))
                ceed._comm_ready = True
        elif halo.world > 1 and not getattr(ceed, "_comm_ready", False):
            if ident is None:
                ident = RcclHalo.unique_id(ceed, halo)
            self.L.chk(lib.CeedXCommInit(ceed.h, C.c_int(halo.world), C.c_int(halo.rank), C.c_char_p(ident)))
            ceed._comm_ready = True
        nn = len(halo.neigh)
        ranks = (C.c_int * max(nn, 1))(*[0 if emulate_self else n.rank for n in halo.neigh])
        counts = (C.c_int * max(nn, 1))(*[int(n.dof_idx.numel()) for n in halo.neigh])
        self._idx = [np.ascontiguousarray(n.dof_idx.cpu().numpy().astype(np.int32)) for n in halo.neigh]
        ptrs = (C.POINTER(C.c_int) * max(nn, 1))(*[a.ctypes.data_as(C.POINTER(C.c_int)) for a in self._idx])
        self.L.chk(lib.CeedXHaloCreate(ceed.h, C.c_int(nn), ranks, counts, ptrs, C.byref(self.h)))

    def start(self, y):
        self.L.chk(self.L.lib.CeedXHaloStart(self.h, y.h))

    def finish(self, y):
        self.L.chk(self.L.lib.CeedXHaloFinish(self.h, y.h))

    def add(self, y):
        self.start(y); self.finish(y)

    def destroy(self):
        if self.h:
            self.L.lib.CeedXHaloDestroy(__import__("ctypes").byref(self.h))
            self.h = None


class HaloBringUpError(RuntimeError):
    """The library's exchange could not be brought up on this job (raised on EVERY rank)."""


def checked_rccl_halo(ceed, halo: "HaloExchange", probe: np.ndarray, device, timeout_s: float = 120.0, strict: bool = True):
    """Bring the library's exchange (RcclHalo) up and CHECK it against the torch exchange of the same test vector before
    it is used.  Collective.  Returns (RcclHalo, note).

    The communicator id is broadcast HERE, on the calling thread (the one that owns the process group); only
    ncclCommInitRank and the first exchange -- the calls that can hang when a peer is missing -- run in a worker thread
    under a time limit, with the rank's device made current first (a new thread starts on device 0).
    * time-out: the worker is stuck inside a collective and nothing sound can be done beside it in this process -- the
      note is printed and the process EXITS with status 3 (a fresh launch is the retry).  Only the stuck rank exits by itself: the
      others are waiting in the collective that follows the bring-up and end when the LAUNCHER ends them -- which is required of it:
      torch.distributed.run (what bench.py --gpus N starts, and what the driver uses) tears the whole job down as soon as one rank
      exits non-zero, as mpirun does; a launcher that does not must put its own limit around the job.
    * exception or wrong sums on any rank: with strict (the default) HaloBringUpError on every rank; without, every rank
      gets (None, note) and may use the torch exchange -- the caller must then SAY so (bench.py: "halo_path")."""
    import sys
    import threading
    from . import ceed as cd
    box = {}
    n = probe.size
    dev = torch.device(device)
    cur_dev = torch.cuda.current_device() if dev.type == "cuda" else None
    try:
        ident = RcclHalo.unique_id(ceed, halo) if halo.world > 1 and not getattr(ceed, "_comm_ready", False) else None
    except RuntimeError as e:       # raised on every rank alike
        if strict:
            raise HaloBringUpError(str(e))
        return None, f"{e}; fell back to torch.distributed point-to-point"

    def bring_up():
        try:
            if dev.type == "cuda":       # a new thread starts on device 0: make the rank's device current
                torch.cuda.set_device(dev if dev.index is not None else cur_dev)
            h = RcclHalo(ceed, halo, ident=ident)
            t = torch.from_numpy(probe).to(dev)
            V = ceed.vector(n)
            V.set_device_pointer(t.data_ptr())
            h.add(V)
            ceed.synchronize()
            if t.is_cuda:
                torch.cuda.synchronize()
            V.take_array(cd.MEM_DEVICE)
            box["h"], box["got"] = h, t
        except Exception as e:   # noqa: BLE001
            box["err"] = repr(e)

    th = threading.Thread(target=bring_up, daemon=True)
    th.start(); th.join(timeout=timeout_s)
    if th.is_alive():
        print(f"[halo] CeedXHalo* bring-up timed out after {timeout_s:.0f} s on rank {halo.rank}: exiting (status 3); "
              f"launch the job again", file=sys.stderr, flush=True)
        import os
        os._exit(3)
    ref = torch.from_numpy(probe).to(dev)
    halo.add(ref)                      # the torch exchange of the same vector: every rank (no worker is alive beside it)
    note, good, h = None, 0.0, None
    if "err" in box:
        note = f"CeedXHalo* failed to initialise ({box['err']})"
    else:
        err = float((box["got"] - ref).abs().max().item()) / max(float(ref.abs().max().item()), 1e-300)
        if err < 1e-12:
            h, good = box["h"], 1.0
        else:
            note = f"CeedXHalo* sums differ from the torch exchange (rel {err:.2e})"
    if halo.world > 1:
        flag = torch.tensor([good], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=halo.group)
        good_all = flag.item() != 0.0
    else:
        good_all = good != 0.0
    if not good_all:
        note = (note or "CeedXHalo* failed on another rank")
        if strict:
            raise HaloBringUpError(note)
        return None, note + "; fell back to torch.distributed point-to-point"
    return h, "CeedXHalo* checked against the torch exchange on a test vector at start-up"


def virtual_world(rank: int, world: int, mesh: HexMesh, part_of, degree: int) -> dict:
    """Stand-in for the collectives of HaloExchange / interface_elements when ONE rank of a `world`-rank job is emulated
    on a single process: `part_of(r)` builds rank r's mesh; of every other rank only the elements that touch a vertex of
    `mesh` are numbered (the topological keys are partition independent, so a sub-mesh gives the same keys)."""
    from .mesh import build_dofmap, submesh
    mine = np.unique(mesh.gid())
    keys, verts = {}, []
    for r in range(world):
        if r == rank:
            continue
        other = part_of(r)
        og = other.gid()
        touch = np.isin(og, mine)
        if not touch.any():
            continue
        elems = np.flatnonzero(touch[other.cells].any(axis=1))
        sub = submesh(other, elems)
        keys[r] = np.sort(key_bytes(build_dofmap(sub, degree, locality_order=False).node_keys))
        verts.append(og[touch])
    return {"rank": rank, "world": world, "keys": keys, "shared_vertices": np.unique(np.concatenate(verts)) if verts else np.zeros(0, dtype=np.int64)}


def interface_elements(mesh: HexMesh, group=None, virtual=None) -> np.ndarray:
    """bool per element: True if the element has a vertex that another rank also holds (its nodes may
    need the halo sum).  Collective; uses global vertex ids only."""
    if virtual is not None:
        return np.isin(mesh.gid(), virtual["shared_vertices"])[mesh.cells].any(axis=1)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.zeros(mesh.nelem, dtype=bool)
    g = mesh.gid()
    # boundary vertices of the local mesh: on a face owned by a single local element
    gc = g[mesh.cells]
    faces = []
    for f in range(6):
        axis, side = f // 2, f % 2
        cs = [c for c in range(8) if ((c >> axis) & 1) == side]
        faces.append(gc[:, cs])
    allf = np.concatenate(faces, axis=0)
    _, inv, cnt = np.unique(np.sort(allf, axis=1), axis=0, return_inverse=True, return_counts=True)
    bverts = np.unique(allf[cnt[inv] == 1])
    gathered: List[Optional[np.ndarray]] = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, bverts, group=group)
    me = dist.get_rank(group)
    others = np.unique(np.concatenate([v for r, v in enumerate(gathered) if r != me] or [np.zeros(0, dtype=np.int64)]))
    shared = np.intersect1d(bverts, others)
    vmask = np.isin(g, shared)
    return vmask[mesh.cells].any(axis=1)


def slab_cylinder(rank: int, world: int, nr: int, nth: int, nz: int, height_per_rank: float = 10.0) -> HexMesh:
    """Weak-scaling workload: rank's z-slab of a hollow cylinder `world` slabs tall, each slab
    the 99k-element stand-in of BASELINE config 4 (R 0.5-1, height 10).  Vertex ids are
    global so interface nodes get identical topological keys on both sides; side sets
    998 / 999 exist only on the bottom / top rank."""
    from .mesh import hollow_cylinder_mesh
    z0 = -0.5 * height_per_rank * world + rank * height_per_rank
    m = hollow_cylinder_mesh(nr, nth, nz, z0=z0, z1=z0 + height_per_rank)
    k = np.arange(m.nvert) // (nth * (nr + 1))
    rest = np.arange(m.nvert) % (nth * (nr + 1))
    m.vertex_gid = (k + rank * nz) * (nth * (nr + 1)) + rest
    if rank != 0:
        m.side_sets.pop(998, None)
    if rank != world - 1:
        m.side_sets.pop(999, None)
    m.name = f"cylslab{rank}of{world}_{nr}x{nth}x{nz}"
    return m


def part_cylinder(rank: int, world: int, nr: int, nth: int, nz: int, height: float = 10.0) -> HexMesh:
    """STRONG-scaling workload (BASELINE config 4 as stated: ONE ~99k-element cylinder over the GPUs of the node,
    src/setupdm.c:57-64): rank's share of the nz element layers of one hollow cylinder (R 0.5-1, `height`), layers
    [rank nz / world, (rank+1) nz / world).  Global vertex ids; side sets 998 / 999 on the bottom / top rank only."""
    from .mesh import hollow_cylinder_mesh
    k0, k1 = rank * nz // world, (rank + 1) * nz // world
    if k1 <= k0:
        raise ValueError(f"{nz} element layers cannot be split over {world} ranks")
    dz = height / nz
    m = hollow_cylinder_mesh(nr, nth, k1 - k0, z0=-0.5 * height + k0 * dz, z1=-0.5 * height + k1 * dz)
    per_layer = nth * (nr + 1)
    m.vertex_gid = (np.arange(m.nvert) // per_layer + k0) * per_layer + np.arange(m.nvert) % per_layer
    if rank != 0:
        m.side_sets.pop(998, None)
    if rank != world - 1:
        m.side_sets.pop(999, None)
    m.name = f"cylpart{rank}of{world}_{nr}x{nth}x{nz}"
    return m


def block_grid(world: int):
    """Blocks per direction for `world` ranks: 8 -> 2x2x2 (BASELINE config 5), 4 -> 2x2x1, 2 -> 2x1x1, else slabs in z."""
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world, (1, 1, world))


def part_box(rank: int, world: int, nx: int, ny: int, nz: int) -> HexMesh:
    """STRONG-scaling box workload (BASELINE config 5: -dm_plex_box_faces 64,64,64 over 8 GPUs): rank's block of ONE
    nx x ny x nz box cut into block_grid(world) blocks (2x2x2 at 8 ranks: a 193^2-node interface per face at p = 6,
    against 385^2 per side for z-slabs).  Global vertex ids of the whole box; face sets 1 (z-) / 2 (z+) where the block
    touches them."""
    from .mesh import box_mesh
    bx, by, bz = block_grid(world)
    ix, iy, iz = rank % bx, (rank // bx) % by, rank // (bx * by)
    rng = lambda n, b, i: (i * n // b, (i + 1) * n // b)
    (x0, x1), (y0, y1), (z0, z1) = rng(nx, bx, ix), rng(ny, by, iy), rng(nz, bz, iz)
    h = 1.0 / nx
    m = box_mesh(x1 - x0, y1 - y0, z1 - z0, lo=(x0 * h, y0 * h, z0 * h), hi=(x1 * h, y1 * h, z1 * h))
    lx, ly = x1 - x0 + 1, y1 - y0 + 1
    v = np.arange(m.nvert)
    i, j, k = v % lx + x0, (v // lx) % ly + y0, v // (lx * ly) + z0
    m.vertex_gid = (k * (ny + 1) + j) * (nx + 1) + i
    if z0 != 0:
        m.side_sets.pop(1, None)
    if z1 != nz:
        m.side_sets.pop(2, None)
    for sid in [s for s in m.side_sets if s not in (1, 2)]:   # lateral faces: interior cuts are not boundaries (and are never clamped here)
        m.side_sets.pop(sid, None)
    m.name = f"boxblock{rank}of{world}_{nx}x{ny}x{nz}"
    return m


def slab_box(rank: int, world: int, nx: int, ny: int, nz: int) -> HexMesh:
    """Weak-scaling box workload (BASELINE config 5 shape: `-dm_plex_box_faces`): rank's z-slab of
    nx x ny x nz unit-spaced elements of a box `world` slabs tall; global vertex ids; face sets 1 (z-)
    and 2 (z+) only on the bottom / top rank."""
    from .mesh import box_mesh
    h = 1.0 / nx
    m = box_mesh(nx, ny, nz, lo=(0.0, 0.0, rank * nz * h), hi=(1.0, ny * h, (rank + 1) * nz * h))
    per_layer = (nx + 1) * (ny + 1)
    k = np.arange(m.nvert) // per_layer
    m.vertex_gid = (k + rank * nz) * per_layer + np.arange(m.nvert) % per_layer
    if rank != 0:
        m.side_sets.pop(1, None)
    if rank != world - 1:
        m.side_sets.pop(2, None)
    m.name = f"boxslab{rank}of{world}_{nx}x{ny}x{nz}"
    return m
