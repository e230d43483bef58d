"""`python bench.py --gpus N`, run DIRECTLY, starts its own ranks: the parent spawns
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child before it touches any GPU, relays rank 0's one
JSON line and the exit code.  Checked here without a device through --dry-run (gloo rendezvous, the real partition and
neighbour lists, interface sums over torch.distributed; no operator, no oracle)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run_bench(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)


def test_gpus_2_run_directly_starts_two_ranks_and_prints_one_line():
    p = run_bench("--gpus", "2", "--dry-run", "--nr", "2", "--nth", "8", "--nz", "8", "--degree", "2", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE line on stdout: everything else went to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dry_run"] is True and d["value"] is None
    assert d["config"]["partition"].startswith("z-layers of ONE mesh")
    assert d["config"]["elements_per_rank"] == [64, 64]      # the ONE 2 x 8 x 8 cylinder, four z-layers each
    assert d["config"]["global_dofs"] == 3 * (5 * 16 * 17 - 2 * 5 * 16)   # 5 x 16 x 17 nodes at p = 2, both end rings clamped
    assert d["config"]["halo_dofs_rank0"] == 3 * 5 * 16      # one interface ring of 5 x 16 nodes
    assert "torch.distributed.run" in p.stderr.decode()      # the parent said what it started


def test_box_blocks_of_four_ranks():
    p = run_bench("--gpus", "4", "--dry-run", "--workload", "box", "--nr", "4", "--nth", "4", "--nz", "2", "--degree", "2", "--steps", "2", "--warmup", "0")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 4 and d["config"]["partition"].startswith("blocks 2x2x1 of ONE box")
    assert d["config"]["elements_per_rank"] == [8, 8, 8, 8]


def test_a_launcher_with_another_rank_count_is_refused():
    p = run_bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0 and b"--gpus 2 but the launcher started 3 ranks" in p.stderr


def test_the_child_exit_code_is_relayed():
    p = run_bench("--gpus", "2", "--dry-run", "--workload", "mesh")    # refused by every rank (exit 1)
    assert p.returncode != 0 and p.stdout.decode().strip() == ""
