#!/usr/bin/env python3
"""bench.py -- MDoF/s of the matrix-free Jacobian apply (BASELINE.json metric).

One "step" = one y = J(u) x with the semantics of ApplyJacobian_Ceed -> ApplyLocalCeedOp
(reference src/matops.c:98-112,26-60): homogeneous Dirichlet rows/columns removed, gather ->
grad -> HyperFSdF -> grad^T -> scatter-add, plus (N > 1) the interface-dof halo sum that
replaces DMLocalToGlobal(ADD_VALUES).  Inputs are resident in HBM before the timed region.

Workload (config.workload): BASELINE config 4 -- hyperFS, degree 4, ~99k-element hollow
cylinder (R 0.5-1, height 10; structured stand-in for the absent cylinder8_99Ke_4ss_us.exo,
same geometry and side-set ids), clamped at both ends.  N > 1 is STRONG scaling, as the config is
stated: that ONE cylinder's element layers are partitioned over the N ranks (--scaling weak: one
such cylinder per rank, round 1's form); ranks exchange the interface dofs over RCCL through the
library's own CeedXHalo* (--halo torch: torch.distributed point-to-point).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 5
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A threaded host BLAS leaves its worker pool (one thread per core of the HOST, not of this job's CPU
# share) spinning for ~0.1 s after every call; under a cgroup CPU quota that throttles the whole
# process, including the thread that launches and waits for the device.  Nothing here needs host BLAS.
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ceedpetscsolid_amd import ceed as cd  # noqa: E402
from ceedpetscsolid_amd.halo import (HaloBringUpError, HaloExchange, RcclHalo, checked_rccl_halo, interface_elements, part_box, part_cylinder,  # noqa: E402
                                     slab_box, slab_cylinder, virtual_world)
from ceedpetscsolid_amd.mesh import reorder_elements_first  # noqa: E402
from ceedpetscsolid_amd.harness import SolidApp  # noqa: E402
from ceedpetscsolid_amd.solid import SolidProblem, smooth_displacement  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def coord_hash_vector(coords: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """Deterministic pseudo-random x in (-1, 1) from node coordinates: identical on every rank
    that shares a node, zero on constrained dofs."""
    k = np.array([[12.9898, 78.233, 37.719], [93.989, 67.345, 24.113], [45.164, 11.135, 83.951]])
    v = np.sin(coords @ k.T) * 43758.5453
    x = 2.0 * (v - np.floor(v)) - 1.0
    return (x.reshape(-1) * (mask == 0)).astype(np.float64)


def algorithmic_bytes(nelem: int, P: int, Q: int, lsize: int, state: bool) -> int:
    """per element 8*(10 [+9])*Q^3 q-point data + 4*P^3 offsets; per L-dof 8 (x) + 8 (y)."""
    return nelem * (8 * (19 if state else 10) * Q ** 3 + 4 * P ** 3) + 16 * lsize


def host_cores() -> int:
    """Host threads this process may really use: affinity, capped by the cgroup CPU quota and by the
    GPU box's per-GPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("BENCH_CPU_THREADS", "16"))))


def isa_kernel_key(kernel_name: str):
    """Name of the fused kernel's instantiation as tools/isa_guard.py lists it, from the operator's kernel name."""
    import re
    m = re.match(r"fused_grad<P=(\d+),Q=(\d+),([\w+]+)>", kernel_name)
    if not m:
        return None
    P, Q, qf = int(m.group(1)), int(m.group(2)), m.group(3)
    geo = 2 if "affine elements" in kernel_name else (3 if "swept elements" in kernel_name else (1 if "recomputed" in kernel_name else 0))
    return f"k_fused_pencil<P={P},Q={Q},{qf},geo={geo},eo={1 if 4 <= Q <= 7 else 0}>"


def isa_line(kernel_name: str, path: str):
    """The instantiation's row of an ISA summary (registers, LDS bytes, instruction counts: tools/isa_guard.py), or None."""
    want = isa_kernel_key(kernel_name)
    if not want or not os.path.exists(path):
        return None
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if f[0] == want:
            return f
    return None


BUILD_ISA = os.path.join(ROOT, "ceedpetscsolid_amd", "csrc", "build", "isa_summary.txt")


def isa_row_named(name: str, path: str):
    """A row of an ISA summary by its first column (tools/isa_guard.py lists k_assemble beside the pencil kernels), or None."""
    if not os.path.exists(path):
        return None
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if f[0] == name:
            return f
    return None


def traffic_from_profile(kernel_name: str, nelem: int, launch_info=None):
    """HBM bytes per operator apply from the newest committed PMC reduction (tools/collect_traffic.py ->
    profiles/rNN_traffic.json), if it was taken on this workload, this kernel AND this launch sequence; (bytes, provenance) or
    (None, reason).  The value is a constant of that profile, NOT a measurement of this run (PMC counters cannot be read from
    inside bench.py), so it is only reported while the profile still describes what runs:
      * the fused kernel's row of the ISA summary (registers, LDS, instruction counts) stored with the profile -- "kernel_isa" in
        the file, else the round's profiles/rNN_isa_summary.txt -- equals the row of THIS build (csrc/build/isa_summary.txt);
      * the same for k_assemble's row ("assemble_isa"; half of the measured bytes are its own), where the profile holds one;
      * the fused launches per apply of this run (CeedXOperatorGetLaunchInfo: segments of the pipelined form) equal the profile's
        per_kernel launches_per_apply -- a run under CEED_MI355X_PIPE_MB / _PIPE_SEGMENTS / _ASSEMBLE=serial moves other bytes.
    Otherwise: null, "stale: ...".  A profile of the right size but another kernel is skipped, not final."""
    import glob
    reasons = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("elements_per_gpu") != nelem:
            continue
        base = os.path.basename(path)
        if d.get("kernel") != kernel_name:
            reasons.append(f"profiles/{base} was taken on kernel '{d.get('kernel')}', this run's is '{kernel_name}'")
            continue
        now = isa_line(kernel_name, BUILD_ISA)
        then = d.get("kernel_isa") or isa_line(kernel_name, os.path.join(ROOT, "profiles", base.split("_")[0] + "_isa_summary.txt"))
        if now is None or then is None:
            return None, f"stale: cannot tell whether profiles/{base} @{d.get('commit')} describes this build (no ISA row: {'build' if now is None else 'profile'})"
        if list(now) != list(then):
            return None, (f"stale: profiles/{base} @{d.get('commit')} was taken on another build of this kernel (VGPR/SGPR/LDS/VALU/ds_read/ds_write "
                          f"then {then[1]}/{then[2]}/{then[3]}/{then[11]}/{then[12]}/{then[13]}, now {now[1]}/{now[2]}/{now[3]}/{now[11]}/{now[12]}/{now[13]}): re-run tools/refresh_profiles.sh")
        a_then, a_now = d.get("assemble_isa"), isa_row_named("k_assemble", BUILD_ISA)
        if a_then is not None and (a_now is None or list(a_now) != list(a_then)):
            return None, f"stale: profiles/{base} @{d.get('commit')} was taken on another build of k_assemble (its ISA row changed): re-run tools/refresh_profiles.sh"
        if launch_info is not None:
            seg_then = [v.get("launches_per_apply") for v in (d.get("per_kernel") or {}).values()]
            if seg_then and any(sg != launch_info["segments"] for sg in seg_then if sg is not None):
                return None, (f"stale: other assembly form -- profiles/{base} @{d.get('commit')} was taken with {seg_then[0]} fused launch(es) per apply, "
                              f"this run makes {launch_info['segments']}")
        guard = "same ISA rows (fused kernel" + (", k_assemble" if a_then is not None else "") + ") and launch sequence as this run"
        return d.get("hbm_bytes_per_apply"), f"profiles/{base} @{d.get('commit', 'round 1')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, calibrated on a known axpby; {guard})"
    if reasons:
        return None, "stale: " + "; ".join(reasons)
    return None, "no committed PMC profile of this workload"


def own_floor_bytes(nelem: int, P: int, Q: int, lsize: int, state: bool) -> int:
    """What the apply as BUILT must move at least: stored state (the geometric factors are recomputed from 168 B of map
    coefficients per element, not streamed), offsets, x and y once."""
    return nelem * (8 * (9 if state else 0) * Q ** 3 + 4 * P ** 3 + 168) + 16 * lsize


def cpu_baseline(args, nr, nth):
    """The oracle (CPU restatement of the reference's /cpu/self path) on the FULL workload -- the same mesh, degree,
    model and state, threaded over element chunks on the cores this job may use (>= 2 timed applies) -- plus a ONE-core
    figure on a thin z-slab of the same cylinder (a full-size one-core apply would take minutes).  elasticity.c:755-764
    is the metric: dofs / time of the operator apply."""
    import ctypes as C
    from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh
    path = os.path.join(ROOT, "oracle", "liboracle_ceed.so")
    if not os.path.exists(path):
        return None
    lib = cd.CeedLib(path)
    cores = host_cores()
    orc = cd.Ceed(lib, "/cpu/self/oracle")

    def rate(mesh, threads, seconds, min_applies, bc):
        lib.lib.OracleSetNumThreads(C.c_int(threads))
        p = SolidProblem(orc, mesh, args.degree, args.problem, nu=args.nu, E=args.E, bc_sides=bc, multigrid="none")
        n = p.lsize()
        X, Y = orc.vector(n), orc.vector(n)
        X.set_array(p.smooth_state(0.1)); p.form_residual(X, Y)
        X.set_array(coord_hash_vector(p.levels[p.fine].dofmap.node_coords, p.levels[p.fine].mask))
        p.apply_jacobian(p.fine, X, Y)  # warm-up
        t0 = time.perf_counter(); k = 0
        while True:
            p.apply_jacobian(p.fine, X, Y); k += 1
            el = time.perf_counter() - t0
            if (el > seconds and k >= min_applies) or k >= 5000:
                break
        out = (1e-6 * p.n_free() * k / el, k, el, mesh.nelem)
        p.destroy()
        return out

    full = hollow_cylinder_mesh(nr, nth, args.nz)
    v, k, el, ne = rate(full, cores, args.cpu_seconds, 2, [998, 999])
    nz1 = args.cpu_sample_layers
    slab = hollow_cylinder_mesh(nr, nth, nz1, z0=-5.0, z1=-5.0 + 10.0 * nz1 / args.nz)
    v1, k1, el1, ne1 = rate(slab, 1, 0.4 * args.cpu_seconds, 2, [998])
    return {"value": v, "unit": "MDoF/s", "cores": cores, "kind": "port", "sample": "full",
            "detail": f"the whole workload ({ne} elements, degree {args.degree} {args.problem}): {k} applies in {el:.1f} s, oracle with OpenMP element chunks on {cores} threads",
            "one_core": {"value": v1, "unit": "MDoF/s", "cores": 1,
                         "sample": f"{ne1}-element z-slab ({nr}x{nth}x{nz1}) of the same cylinder, {k1} applies in {el1:.1f} s"}}


def valu_issue(kernel_name: str, ngroups: int, seconds: float):
    """Share of the f64 vector pipe's issue slots the fused kernel's instruction stream needs: VALU instructions of one
    element group (static count from the build's disassembly, tools/isa_guard.py) x groups x 4 cycles (a wave64 VALU
    instruction occupies its SIMD's 16 lanes for 4 cycles) / (1024 SIMDs x 2.4 GHz) / time.  The kernel's binding limit
    (VERDICT r2, weak 3): 1.0 would be a vector pipe that never idles."""
    for path in (BUILD_ISA, os.path.join(ROOT, "profiles", "r04_isa_summary.txt")):
        f = isa_line(kernel_name, path)
        if f:
            valu = int(f[11])
            t_valu = valu * ngroups * 4 / (1024 * 2.4e9)
            return {"valu_instructions_per_group": valu, "groups": ngroups, "valu_issue_us": 1e6 * t_valu,
                    "valu_issue_frac": t_valu / seconds, "of": "the whole apply's device time (kernel_avg_us), k_assemble included", "source": os.path.relpath(path, ROOT),
                    "assumes": "4 cycles per wave64 VALU instruction, 1024 SIMDs, 2.4 GHz (spec clock; the clock measured inside the kernel under this load is 2.0 GHz and two waves per SIMD issue one f64 instruction per 4.4 cycles: profiles/r03_phase_timing.txt, profiles/r04_mfma_f64.txt); the count is the kernel's static one, prologue included"}
    return None


def self_launch(n: int) -> None:
    """`python bench.py --gpus N` run directly (no launcher in the environment): start the N ranks as a CHILD process --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
    arguments>` -- BEFORE anything in this process touches the GPU (a process that has initialised the GPU must not be
    replaced, and the parent needs none), relay rank 0's single JSON line and exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ... as a child process", file=sys.stderr, flush=True)
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in child.stdout.decode(errors="replace").splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    for ln in child.stdout.decode(errors="replace").splitlines():
        if ln not in lines[-1:]:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif child.returncode == 0:
        sys.exit("the ranks exited with status 0 but printed no JSON line")
    sys.exit(child.returncode)


def partition_text(args, world: int, overlap: bool, use_rccl: bool, emu: bool) -> str:
    if not (world > 1 or emu):
        return "single GPU"
    strong = args.scaling == "strong"
    if strong and args.workload == "box":
        from ceedpetscsolid_amd.halo import block_grid
        what = "blocks %dx%dx%d of ONE box" % block_grid(args.of if emu else world)
    else:
        what = "z-layers of ONE mesh" if strong else "one such mesh per GPU (z-slabs)"
    return (what + "; halo sum " + ("overlapped with interior elements" if overlap else "after the apply")
            + (" through CeedXHalo* (RCCL group of ncclSend/ncclRecv, pack / unpack-add kernels)" if use_rccl else " through torch.distributed P2P"))


def dry_run(args, world: int, rank: int) -> None:
    """--dry-run: the launch path of the N-rank job WITHOUT a device and without the operator -- the ranks rendezvous on
    gloo, partition the ONE mesh exactly as the real run does, find their neighbour lists, and time --steps interface sums
    of a test vector over torch.distributed (halo.HaloExchange on CPU tensors).  Rank 0 prints the line with "value": null
    and "dry_run": true: what a CPU test (tests/test_bench_launch.py) and a rehearsal before an 8-GPU run can check --
    that `bench.py --gpus N` starts, that every rank gets its share, and what the line says about itself."""
    from ceedpetscsolid_amd.mesh import build_dofmap, dirichlet_mask, side_set_nodes
    if world > 1:
        dist.init_process_group("gloo")
    strong = args.scaling == "strong"
    if args.workload == "cylinder":
        mesh = part_cylinder(rank, world, args.nr, args.nth, args.nz) if strong else slab_cylinder(rank, world, args.nr, args.nth, args.nz)
        bc = [s for s in (998, 999) if s in mesh.side_sets]
    elif args.workload == "box":
        mesh = part_box(rank, world, args.nr, args.nth, args.nz) if strong else slab_box(rank, world, args.nr, args.nth, args.nz)
        bc = [s for s in (1, 2) if s in mesh.side_sets]
    else:
        sys.exit("--dry-run: cylinder or box workload")
    lead = interface_elements(mesh)
    overlap = world > 1 and not args.no_overlap and lead.any() and not lead.all()
    if overlap:
        mesh = reorder_elements_first(mesh, lead)
    dofmap = build_dofmap(mesh, args.degree)
    mask = np.ascontiguousarray(dirichlet_mask(dofmap, side_set_nodes(mesh, dofmap, bc) if bc else np.zeros(0, dtype=np.int64)), dtype=np.uint8)
    n = dofmap.lsize
    halo = HaloExchange(mesh, dofmap, device=torch.device("cpu"))
    n_global = halo.global_count((mask == 0).astype(np.float64))
    yt = torch.from_numpy(coord_hash_vector(dofmap.node_coords, np.zeros(n, dtype=np.uint8)))
    y0 = yt.clone()
    for _ in range(args.warmup):
        yt.copy_(y0); halo.add(yt)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        halo.add(yt)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # every interface entry of the once-summed test vector = multiplicity over the ranks x the entry (the vector is rank-independent)
    yt.copy_(y0); halo.add(yt)
    ok = bool(torch.isfinite(yt).all())
    if world > 1:
        t = torch.tensor([elapsed, float(mesh.nelem)], dtype=torch.float64)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        ne = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ne, t[1:])
        elapsed, elems = float(t[0]), [int(v.item()) for v in ne]
    else:
        elems = [mesh.nelem]
    if rank == 0:
        Q = args.degree + 1
        print(json.dumps({
            "metric": "MDoF/s for matrix-free Jacobian apply, p=4 hyperFS hex, 1/2/4/8 GPU", "value": None, "unit": "MDoF/s",
            "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "halo_path": None if world == 1 else "torch",
            "rccl_ranks": None, "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"DRY RUN (no device, no operator: partition + gloo interface sums only) of {args.workload} "
                                   f"{args.nr}x{args.nth}x{args.nz}, degree {args.degree}, Q={Q}",
                       "global_dofs": n_global, "elements_per_rank": elems, "ldofs_rank0": n, "halo_dofs_rank0": halo.n_shared_dofs,
                       "partition": partition_text(args, world, overlap, False, False), "finite": ok}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--problem", default="hyperFS")
    ap.add_argument("--nr", type=int, default=10)
    ap.add_argument("--nth", type=int, default=110)
    ap.add_argument("--nz", type=int, default=90)
    ap.add_argument("--mesh", default=os.path.join(ROOT, "tests", "golden", "mesh_cylinder8_44928e_2ss_us.npz"),
                    help="--workload mesh: an unstructured HEX8 mesh fixture (the reference's cylinder8_44928e_2ss_us.exo, CUBIT order)")
    ap.add_argument("--workload", default="cylinder", choices=["cylinder", "box", "mesh"],
                    help="cylinder: BASELINE config 4 (default, the metric's config); box: config 5 shape, "
                         "nr x nth x nz elements per GPU (e.g. --workload box --nr 32 --nth 32 --nz 32 --degree 6)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = ONE mesh of the stated size partitioned over the ranks, as BASELINE configs 4 / 5 are "
                         "stated (setupdm.c:57-64: DMPlexDistribute of one mesh); weak = one such mesh PER rank (round 1's form)")
    ap.add_argument("--halo", default="rccl", choices=["rccl", "torch"],
                    help="N > 1 on the nccl backend: rccl = the library's own exchange (CeedXHalo*: pack kernel, ncclSend/ncclRecv group on "
                         "its stream, unpack-add kernel); torch = torch.distributed point-to-point + index ops (also the gloo rehearsal)")
    ap.add_argument("--no-strict-halo", action="store_true",
                    help="N > 1 with --halo rccl: if the library's exchange cannot be brought up, time the torch exchange instead of "
                         "exiting non-zero (the line then says \"halo_path\": \"torch-fallback\"); strict is the default")
    ap.add_argument("--nu", type=float, default=0.3)
    ap.add_argument("--E", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: do not hide the halo exchange under the interior elements")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-sample-layers", type=int, default=3)
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed applies for this long before --warmup: the first ~20 ms after an idle period run at a "
                         "transient clock (per-dispatch trace in profiles/r02_dispatch_series.txt); reported as prewarm_ms")
    ap.add_argument("--emulate-rank", type=int, default=-1, metavar="K",
                    help="ONE GPU, no torch.distributed: run rank K's share of the --of N strong-scaling job -- its real sub-mesh, the "
                         "split-phase apply exactly as at N > 1, and the library's RCCL exchange on a one-rank communicator whose "
                         "neighbour lists have the TRUE sizes, sent to this rank itself (everything one GPU can tell about the N-GPU regime)")
    ap.add_argument("--of", type=int, default=8, metavar="N")
    ap.add_argument("--cold-idle-s", type=float, default=1.0,
                    help="idle time before the COLD timing of --steps applies (reported as ms_per_step_cold; 0: skipped)")
    ap.add_argument("--per-apply-events", action="store_true",
                    help="kernel_avg_us from a pair of hipEvents around EVERY apply (round 1-3's form) instead of one pair around the timed steps")
    ap.add_argument("--graph", action="store_true",
                    help="record one step (apply + exchange) into a hipGraph after the first applies and time its replays "
                         "(CeedXGraph*; the exchange is recorded in its in-order form)")
    ap.add_argument("--phase-timing", default=None, metavar="FILE",
                    help="diagnostic, needs a library built with -DCPS_PHASE_TIMING=<k> (tools/mkvariant.sh): after the timed loop one more "
                         "apply with the waves' phase time stamps collected; mean shader cycles per phase of the fused kernel -> FILE")
    ap.add_argument("--calibrate-traffic", action="store_true",
                    help="also launch k_axpby over a 1 GiB vector (known byte count) for PMC calibration")
    ap.add_argument("--refine-layers", type=int, default=0, metavar="NZ",
                    help="--workload mesh: re-make the swept mesh with its cross-section refined 2 x 2 and NZ uniform layers (mesh.refine_swept_mesh); "
                         "the reference's cylinder8_44928e with NZ = 53 is 99 216 hexes on a REAL unstructured (CUBIT-paved) cross-section -- the "
                         "closest thing to the absent cylinder8_99Ke_4ss_us.exo")
    ap.add_argument("--scramble", default="none", choices=["none", "order", "all"],
                    help="one GPU: the SAME mesh as an unstructured generator might hand it over (mesh.scramble_mesh) -- order: elements and "
                         "vertices in random order; all: also every element's local axes relabelled by a random rotation (no common sweep "
                         "direction: the general geometry path).  The stand-in for the absent cylinder8_99Ke_4ss_us.exo at its worst")
    ap.add_argument("--reorder", action="store_true",
                    help="sort the elements along a Morton curve through their centroids before numbering the nodes (mesh.reorder_elements_locality): "
                         "what the mesh layer can do for a badly ordered mesh (use with --scramble or --workload mesh)")
    ap.add_argument("--blocks", type=int, default=5,
                    help="timed blocks of --steps applies each (every block between a barrier + synchronize on both sides, one hipEvent pair "
                         "each): ms_per_step and value are the MEDIAN block's, the spread over the blocks is reported beside them")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip the untimed block of applies with the shader-clock probe beside it (profiling runs: the trace then ends with the timed block)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no device, no operator: the N ranks rendezvous on gloo, partition the mesh as the real run does and time the "
                         "interface sums of a test vector over torch.distributed; the line says \"dry_run\": true and \"value\": null")
    args = ap.parse_args()
    # `python bench.py --gpus N` without a launcher: start the ranks ourselves, as a child, before any GPU call here
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"--gpus {args.gpus} but the launcher started {os.environ.get('WORLD_SIZE', '1')} ranks")
    if args.dry_run:
        return dry_run(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))

    # ONE JSON line on stdout: libraries that print there on their own (RCCL's version banner at communicator start-up) go to
    # stderr until the line is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ngpu_visible = torch.cuda.device_count()
    local_rank = local_rank % max(1, ngpu_visible)   # rehearsal: several ranks may share the one visible GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one node (the contract of --gpus N): RCCL's socket bootstrap -- torch's communicator and the library's own -- over the loopback
        # interface, which always exists and always resolves (the container's hostname may not); the data goes over xGMI either way
        if os.environ.get("LOCAL_WORLD_SIZE", str(world)) == str(world):   # (a launcher that spans nodes keeps RCCL's own choice)
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")   # "gloo": single-GPU rehearsal of the N > 1 path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    lib = cd.CeedLib(cd.PRODUCT_LIB)          # fails loudly if the HIP library is missing
    ceed = cd.Ceed(lib, "/gpu/hip/mi355x")
    stream = torch.cuda.current_stream()
    ceed.set_stream(stream.cuda_stream)

    # ---- workload ---------------------------------------------------------
    strong = args.scaling == "strong"
    emu = args.emulate_rank >= 0
    vw = None
    if emu:
        if world != 1 or args.workload == "mesh" or not 0 <= args.emulate_rank < args.of:
            sys.exit("--emulate-rank K --of N runs on one GPU without torch.distributed (cylinder or box workload)")
        part = (lambda r: part_cylinder(r, args.of, args.nr, args.nth, args.nz)) if args.workload == "cylinder" else \
               (lambda r: part_box(r, args.of, args.nr, args.nth, args.nz))
        mesh = part(args.emulate_rank)
        vw = virtual_world(args.emulate_rank, args.of, mesh, part, args.degree)
        bc = [s for s in ((998, 999) if args.workload == "cylinder" else (1, 2)) if s in mesh.side_sets]
    elif args.workload == "cylinder":
        mesh = part_cylinder(rank, world, args.nr, args.nth, args.nz) if strong else slab_cylinder(rank, world, args.nr, args.nth, args.nz)
        bc = [s for s in (998, 999) if s in mesh.side_sets]
    elif args.workload == "mesh":   # one unstructured mesh, z-slab partition of its elements over the ranks (strong scaling)
        from ceedpetscsolid_amd.mesh import load_mesh_npz, partition_slabs, submesh
        mesh = load_mesh_npz(args.mesh)
        if args.refine_layers > 0:
            from ceedpetscsolid_amd.mesh import refine_swept_mesh
            mesh = refine_swept_mesh(mesh, args.refine_layers)
        if world > 1:
            mesh = submesh(mesh, partition_slabs(mesh, world)[rank])
        bc = [s for s in (998, 999) if s in mesh.side_sets and len(mesh.side_sets[s])]
    else:
        mesh = part_box(rank, world, args.nr, args.nth, args.nz) if strong else slab_box(rank, world, args.nr, args.nth, args.nz)
        bc = [s for s in (1, 2) if s in mesh.side_sets]
    if args.scramble != "none":
        if world != 1 or emu:
            sys.exit("--scramble runs on one GPU")
        from ceedpetscsolid_amd.mesh import scramble_mesh
        mesh = scramble_mesh(mesh, 20261005, order=True, orient=args.scramble == "all")
    if args.reorder:
        from ceedpetscsolid_amd.mesh import reorder_elements_locality
        mesh = reorder_elements_locality(mesh)
    lead = interface_elements(mesh, virtual=vw)        # collective; all False on one rank
    overlap = (world > 1 or emu) and not args.no_overlap and lead.any() and not lead.all()
    if overlap:
        mesh = reorder_elements_first(mesh, lead)      # interface-touching elements lead (split-phase apply)
    # host side = the C++ harness (csrc/solid_harness.cpp): SetupLibceedFineLevel / SetupLibceedLevel /
    # ApplyJacobian_Ceed restated over include/ceed.h
    prob = SolidApp(ceed, mesh, args.degree, args.problem, nu=args.nu, E=args.E, bc_sides=bc, multigrid="none")
    dofmap, mask = prob.dofmaps[prob.fine], prob.masks[prob.fine]
    n = prob.lsize()
    halo = HaloExchange(mesh, dofmap, device=dev, virtual=vw)
    # the exchange itself: behind the C ABI over RCCL on a GPU node; the torch path on gloo (single-GPU rehearsal) or on request
    use_rccl = world > 1 and args.halo == "rccl" and dist.get_backend() == "nccl"
    chalo, halo_note = None, None
    halo_path = None if world == 1 and not emu else ("rccl" if (use_rccl or emu) else "torch")
    if world > 1 and os.environ.get("BENCH_TEST_BRING_UP_FAILURE"):   # (rehearsal of the failure record on a box without RCCL peers)
        use_rccl = True
    if use_rccl:   # brought up with a time limit and checked against the torch exchange; a failure ends the run (exit 4) unless --no-strict-halo
        try:
            if os.environ.get("BENCH_TEST_BRING_UP_FAILURE"):
                raise HaloBringUpError("rehearsal: BENCH_TEST_BRING_UP_FAILURE is set")
            chalo, halo_note = checked_rccl_halo(ceed, halo, coord_hash_vector(dofmap.node_coords, np.zeros(n, dtype=np.uint8)), dev,
                                                 strict=not args.no_strict_halo)
        except HaloBringUpError as e:
            print(f"[bench] rank {rank}: {e}; not timing a fallback (--no-strict-halo would)", file=sys.stderr, flush=True)
            if rank == 0:   # the record says WHY: a line with no value, the reason, and the communicator as far as it got
                try:
                    ranks = ceed.comm_size()[0] or None
                except Exception:   # noqa: BLE001
                    ranks = None
                sys.stdout.flush()
                os.dup2(stdout_fd, 1)
                print(json.dumps({"metric": "MDoF/s for matrix-free Jacobian apply, p=4 hyperFS hex, 1/2/4/8 GPU", "value": None, "unit": "MDoF/s",
                                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "rccl_ranks": ranks,
                                  "halo_path": f"failed: {e}", "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling,
                                  "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                                  "config": {"workload": f"{args.workload} {args.nr}x{args.nth}x{args.nz}, degree {args.degree}: not run (the library's RCCL "
                                                         "exchange could not be brought up; --no-strict-halo times the torch exchange instead)"}}), flush=True)
            dist.destroy_process_group()
            sys.exit(4)
        use_rccl = chalo is not None
        if not use_rccl:
            halo_path = "torch-fallback"
    if emu:   # one-rank communicator, every neighbour list exchanged with this rank itself
        chalo, halo_note, use_rccl = RcclHalo(ceed, halo, emulate_self=True), "emulated rank: self-exchange of the true neighbour lists", True
    free = (mask == 0).astype(np.float64)
    n_global = halo.global_count(free)

    # state u (stores gradu through the residual), then the Jacobian input x
    if args.workload == "mesh":   # (the reference cylinders: radius 1, height 2.x -- any fixed box works, the field only has to be rank-independent)
        xt = torch.from_numpy(smooth_displacement(dofmap.node_coords, 0.1, origin=(-2.0, -2.0, -2.0), span=(4.0, 4.0, 8.0))).to(dev)
    else:
        xt = torch.from_numpy(smooth_displacement(dofmap.node_coords, 0.1, origin=(-1.0, -1.0, 0.0), span=(2.0, 2.0, 10.0))).to(dev)
    yt = torch.zeros(n, dtype=torch.float64, device=dev)
    X, Y = ceed.vector(n), ceed.vector(n)
    X.set_device_pointer(xt.data_ptr()); Y.set_device_pointer(yt.data_ptr())   # matops.c:40-41, -memtype device
    prob.form_residual(X, Y)
    xt.copy_(torch.from_numpy(coord_hash_vector(dofmap.node_coords, mask)).to(dev))
    op = prob.opJacob[prob.fine]
    if overlap:
        op.set_overlap_split(int(lead.sum()), halo.interface_dof_mask())

    def step():
        if overlap and chalo:   # ONE library call: interface elements, the RCCL exchange started, interior elements beside it, arrivals added
            op.apply_with_halo(X, Y, chalo)
        elif overlap:           # torch exchange (gloo rehearsal): the same split, driven from here
            op.apply_phase(X, Y, 0)
            halo.start(yt)
            op.apply_phase(X, Y, 1)
            halo.finish(yt)
        else:
            prob.apply_jacobian(prob.fine, X, Y)   # ApplyJacobian_Ceed: k_fused_pencil + k_assemble on `stream`
            chalo.add(Y) if chalo else halo.add(yt)   # interface sum (no-op at N = 1)

    # cold figure (reported beside the warm one): the first --steps applies after a 1 s idle, before any pre-warm -- the
    # clock / power transient after idle makes them ~8 % slower (profiles/r02_dispatch_series.txt)
    ms_cold = None
    if args.cold_idle_s > 0:
        step(); torch.cuda.synchronize()        # one apply first: transpose maps, flags and streams are set up by it
        time.sleep(args.cold_idle_s)
        if world > 1:
            dist.barrier()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms_cold = 1e3 * (time.perf_counter() - tc) / args.steps
    if args.calibrate_traffic:   # known traffic for tools/collect_traffic.py: reads 2 GiB, writes 1 GiB
        import ctypes as C
        ncal = 2 ** 27
        va, vb = ceed.vector(ncal).set_value(1.0), ceed.vector(ncal).set_value(2.0)
        for _ in range(3):
            lib.chk(lib.lib.CeedXVectorAXPBY(va.h, C.c_double(0.5), vb.h, C.c_double(0.25)))
        torch.cuda.synchronize()
    if args.graph:   # one step recorded; every later step() is a replay
        step(); torch.cuda.synchronize()
        recorded = ceed.capture(step)
        step = recorded.launch
    # pre-warm (untimed, reported): back-to-back applies until the clock / power state has settled
    prewarm_steps = 0
    if args.prewarm_ms > 0:
        torch.cuda.synchronize()
        tp = time.perf_counter()
        while 1e3 * (time.perf_counter() - tp) < args.prewarm_ms:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            prewarm_steps += 10
    for _ in range(args.warmup):
        step()
    # Device time of the timed region: ONE pair of hipEvents on the operator's stream (= torch's current stream, ceed.set_stream
    # above; the side stream of a pipelined apply forks from it and joins it) around the K steps.  --per-apply-events brackets
    # every apply instead (CeedXOperatorSetTiming: two event records per apply, which cost the small launches of the N-GPU
    # regime ~10 % -- profiles/r03_ab_experiments.txt item 16).
    if args.per_apply_events:
        op.set_timing(True)
    # --blocks timed blocks of EXACTLY --steps applies, each between a barrier + synchronize on both sides and inside one hipEvent
    # pair; the line's ms_per_step / value are the MEDIAN block's (max over the ranks per block), the spread says how noisy the box is
    nblk = max(1, args.blocks)
    blk_wall, blk_dev = [], []
    for _b in range(nblk):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        blk_wall.append(time.perf_counter() - t0)
        blk_dev.append(ev0.elapsed_time(ev1))
    if world > 1:
        t = torch.tensor(blk_wall, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blk_wall = [float(v) for v in t.tolist()]
    order = sorted(range(nblk), key=lambda i: blk_wall[i])
    imed = order[(nblk - 1) // 2]
    elapsed = blk_wall[imed]
    if args.per_apply_events:
        kernel_ms, launches = op.get_timing()
        kernel_ms, launches = kernel_ms / nblk, launches // nblk
        op.set_timing(False)
    else:
        kernel_ms, launches = blk_dev[imed], args.steps
    # the shader clock WHILE the applies run: one more (untimed) block is queued and a one-wave probe on a stream of its own counts
    # shader cycles against the 100 MHz counter for part of it -- a slow box (clock) is then distinguishable from a slow build
    clock_ghz = None
    if not args.no_clock_probe:
        try:
            for _ in range(args.steps):
                step()
            clock_ghz = ceed.clock_probe(int(max(200, min(2000, 0.5e3 * blk_dev[imed]))))
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001  (informational)
            clock_ghz = None
            print(f"[bench] clock probe failed: {e}", file=sys.stderr)
    if args.phase_timing:
        pbuf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
        os.environ["CEED_MI355X_PHASE_BUF"] = hex(pbuf.data_ptr())
        step(); torch.cuda.synchronize()
        del os.environ["CEED_MI355X_PHASE_BUF"]
        tb = pbuf.cpu().numpy().reshape(-1, 32)
        tb = tb[tb[:, 0] > 0]
        if len(tb) == 0:
            sys.exit("--phase-timing: no time stamps came back -- the library in ceedpetscsolid_amd/csrc was not built with -DCPS_PHASE_TIMING=<k> "
                     "(tools/mkvariant.sh ph1 \"-DCPS_PHASE_TIMING=1\", then copy tools/variants/ph1/*.so over it as tools/r3_phase_timing.sh does)")
        names = ["requests+gather", "F1", "F2", "F3", "F4", "F5", "geo->LDS"] + [f"physics {r}" for r in range(9)] + ["B1", "B2", "B3", "x+B4", "B5", "final"]
        with open(args.phase_timing, "w") as f:
            f.write(f"# {len(tb)} waves, shader-clock cycles per phase of one group (mean, median, p90)\n")
            prev = 0
            for i in range(1, 23):
                if not (tb[:, i] > 0).all():
                    continue
                d = tb[:, i] - tb[:, prev]
                f.write("%-16s %8.0f %8.0f %8.0f\n" % (names[prev], d.mean(), np.median(d), np.percentile(d, 90)))
                prev = i
            d = tb[:, prev] - tb[:, 0]
            f.write("%-16s %8.0f %8.0f %8.0f\n" % ("whole group", d.mean(), np.median(d), np.percentile(d, 90)))
            if (tb[:, 25] > tb[:, 24]).all():   # the same interval on the constant 100 MHz counter: the shader clock during it
                f.write("shader clock     %8.3f GHz (cycles of the group / its time on the 100 MHz wall clock, mean over the waves)\n"
                        % float((d / ((tb[:, 25] - tb[:, 24]) * 10.0)).mean()))
    # the exchange alone (outside the timed region): mean of 20 back-to-back interface sums
    halo_us = None
    if world > 1 or emu:
        ykeep = yt.clone()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        th = time.perf_counter()
        for _ in range(20):
            chalo.add(Y) if chalo else halo.add(yt)
        torch.cuda.synchronize()
        halo_us = 1e6 * (time.perf_counter() - th) / 20
        yt.copy_(ykeep)
    # sanity on the result of the last step (cheap, outside the timed region)
    ynorm = float(torch.linalg.vector_norm(yt).item())
    assert np.isfinite(ynorm) and ynorm > 0.0

    if rank == 0:
        P, Q = args.degree + 1, args.degree + 1
        abytes = algorithmic_bytes(mesh.nelem, P, Q, n, args.problem != "linElas")
        avg_s = kernel_ms * 1e-3 / max(args.steps, 1)   # per APPLY (a split-phase apply is two timed launch pairs)
        if avg_s <= 0.0:   # --graph with --per-apply-events: the replays carry no hipEvents; the step time stands in
            avg_s = elapsed / args.steps
        li = op.launch_info()
        assembly_form = ("pipelined: %d segments on %d streams, last segment %d elements" % (li["segments"], li["streams"], li["last_segment_elements"])
                         if li["segments"] > 1 else os.environ.get("CEED_MI355X_ASSEMBLE", "serial (launch too small to pipeline, split-phase apply, or switched off)"))
        achieved = abytes / avg_s / 1e9
        traffic, traffic_src = traffic_from_profile(op.kernel_name, mesh.nelem, li)
        out = {
            "metric": "MDoF/s for matrix-free Jacobian apply, p=4 hyperFS hex, 1/2/4/8 GPU",
            "value": 1e-6 * n_global * args.steps / elapsed,
            "unit": "MDoF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "rccl_ranks": (ceed.comm_size()[0] or None),   # ranks of the library's communicator READ BACK from RCCL (ncclCommCount); null: none was created
            "halo_path": halo_path,   # N > 1: "rccl" = the library's CeedXHalo* (pack kernel, RCCL group, unpack-add); "torch" on request / on gloo; "torch-fallback" only with --no-strict-halo
            "prewarm_ms": args.prewarm_ms, "prewarm_steps": prewarm_steps,
            "launch": "hipGraph replay of one recorded step" if args.graph else "direct",
            "ms_per_step": 1e3 * elapsed / args.steps,     # the MEDIAN of the timed blocks (below)
            "timed_blocks": {"blocks": nblk, "steps_per_block": args.steps, "ms_per_step": [1e3 * w / args.steps for w in blk_wall],
                             "device_ms_per_step": [d / args.steps for d in blk_dev],
                             "min": 1e3 * min(blk_wall) / args.steps, "median": 1e3 * elapsed / args.steps, "max": 1e3 * max(blk_wall) / args.steps,
                             "spread_pct": 100.0 * (max(blk_wall) - min(blk_wall)) / elapsed,
                             "note": "each block: barrier + synchronize, --steps applies inside one hipEvent pair, synchronize + barrier; wall time, max over the ranks"},
            "shader_clock_GHz_under_load": clock_ghz,   # CeedXClockProbe during one more untimed block (2.4 GHz spec; ~2.0 under this kernel's f64 load)
            "ms_per_step_cold": ms_cold,   # the same loop right after a 1 s idle, before the pre-warm (rank 0's clock; not max-reduced)
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"config 4: {args.problem}, hollow cylinder {args.nr}x{args.nth}x{args.nz} = "
                                    f"{mesh.nelem} hex per GPU (stand-in for cylinder8_99Ke_4ss_us.exo), degree {args.degree}, "
                                    f"Q={Q}, clamped ends, Jacobian apply y=J(u)x") if args.workload == "cylinder" else
                                   (f"unstructured reference mesh {os.path.basename(args.mesh)}" + (f", cross-section refined 2 x 2, {args.refine_layers} layers" if args.refine_layers > 0 else "") + f": {args.problem}, {mesh.nelem} hex on this rank, "
                                    f"degree {args.degree}, Q={Q}, side sets {bc} clamped, Jacobian apply y=J(u)x") if args.workload == "mesh" else
                                   (f"config 5 shape: {args.problem}, box {args.nr}x{args.nth}x{args.nz} = {mesh.nelem} hex per GPU, "
                                    f"degree {args.degree}, Q={Q}, z faces clamped, Jacobian apply y=J(u)x"),
                       "reorder": "elements sorted along a Morton curve through their centroids" if args.reorder else None,
                       "scramble": None if args.scramble == "none" else ("elements and vertices in random order" + (", every element's local axes rotated at random" if args.scramble == "all" else "")),
                       "global_dofs": n_global, "elements_per_gpu": mesh.nelem, "ldofs_per_gpu": n,
                       "halo_dofs_rank0": halo.n_shared_dofs, "kernel": op.kernel_name,
                       "assembly": assembly_form, "schedule": os.environ.get("CEED_MI355X_SCHED", "static"),
                       "partition": partition_text(args, world, overlap, use_rccl, emu),
                       "halo_exchange_us_alone": halo_us, "halo_note": halo_note, "multi_gpu_measured": (None if world == 1 else "this run")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # `achieved` prices the REFERENCE formulation's bytes (qdata and gradu streamed) over the
                         # measured time: an effective rate.  The kernel as built recomputes qdata (own_floor_bytes) and makes
                         # an E-vector round trip (the difference between traffic and own_floor_bytes).
                         "own_floor_bytes": own_floor_bytes(mesh.nelem, P, Q, n, args.problem != "linElas"),
                         "measured_traffic_GBs": (traffic / avg_s / 1e9) if traffic else None,
                         "algorithmic_bytes_per_launch": abytes, "kernel_avg_us": avg_s * 1e6,
                         "kernel_launches_timed": launches,
                         "kernel_avg_us_from": "a pair of hipEvents around every apply" if args.per_apply_events else "one pair of hipEvents on the operator's stream around the timed steps / steps",

                         "launches_per_apply": li["segments"] + li["assemble_launches"],
                         "kernels": ("k_fused_pencil (gather..physics..shell E-vector, interior nodes straight to y) + k_assemble (deterministic per-node sum of the shared nodes): "
                                     + ("the two launches of one CeedOperatorApply, timed together with hipEvents on their stream" if li["segments"] <= 1 else
                                        "one CeedOperatorApply = %d segments of consecutive elements, each a fused launch followed by the k_assemble of the rows it completes, alternating between %d streams so that the rows of a segment are summed beside the next fused kernel; kernel_avg_us is the hipEvent time of the WHOLE apply on the operator's stream (fork to join), not a sum of per-kernel durations, which overlap (profiles/README.md)" % (li["segments"], li["streams"]))),
                         "peak_measured_copy_GBs": 6290.0,
                         "valu_issue": valu_issue(op.kernel_name, (mesh.nelem + (2 if Q == 5 else (1 if Q >= 6 else (4 if Q >= 3 else 8))) - 1) // (2 if Q == 5 else (1 if Q >= 6 else (4 if Q >= 3 else 8))), avg_s)},
        }
        if emu:
            out["emulated_rank"] = {"rank": args.emulate_rank, "of": args.of, "lead_elements": int(lead.sum()),
                                    "neighbour_dofs": [int(nb.dof_idx.numel()) for nb in halo.neigh],
                                    "us_per_apply_incl_exchange": 1e6 * elapsed / args.steps,
                                    "note": "value = dofs this rank OWNS x steps / time: one rank's share of the N-rank job, NOT the job's rate"}
        if not args.no_cpu_baseline and world == 1 and args.workload == "cylinder" and not emu:
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.nr, args.nth)
            except Exception as e:  # the baseline is informational; never hide the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
