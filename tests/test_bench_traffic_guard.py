"""bench.py reports `roofline.traffic` -- a constant of a committed PMC profile, not a measurement of the run -- only while the profile
still describes what runs: the fused kernel's and k_assemble's rows of the ISA summary and the
launch sequence must equal the running build's; a profile of another kernel is skipped, not final."""
import json
import os

import bench

KERNEL = "fused_grad<P=5,Q=5,HyperFSdF>/pencil [swept elements: 2 x 2 dXdx recomputed per point]"
ROW = ["k_fused_pencil<P=5,Q=5,HyperFSdF,geo=3,eo=1>", "184", "92", "18496", "0", "0", "0", "0", "0", "0", "0", "2242", "280", "215"]
ASM = ["k_assemble", "44", "46", "0", "0", "0", "0", "0", "0", "0", "0", "120", "0", "0"]


def _setup(tmp_path, monkeypatch, profiles):
    (tmp_path / "profiles").mkdir()
    for name, d in profiles.items():
        (tmp_path / "profiles" / name).write_text(json.dumps(d))
    isa = tmp_path / "isa_summary.txt"
    isa.write_text("kernel\tvgpr\n" + "\t".join(ROW) + "\n" + "\t".join(ASM) + "\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "BUILD_ISA", str(isa))


def _profile(**kw):
    d = {"kernel": KERNEL, "elements_per_gpu": 99000, "commit": "abc", "kernel_isa": ROW, "assemble_isa": ASM, "hbm_bytes_per_apply": 2.08e9,
         "per_kernel": {"k_fused_pencil<5, 5, 6": {"launches_per_apply": 2}, "k_assemble(": {"launches_per_apply": 2}}}
    d.update(kw)
    return d


def test_traffic_is_reported_while_the_profile_describes_the_run(tmp_path, monkeypatch):
    _setup(tmp_path, monkeypatch, {"r05_traffic.json": _profile()})
    v, src = bench.traffic_from_profile(KERNEL, 99000, {"segments": 2})
    assert v == 2.08e9 and "k_assemble" in src and "launch sequence" in src
    assert bench.traffic_from_profile(KERNEL, 12345, {"segments": 2})[0] is None          # another workload: no profile


def test_stale_profiles_are_refused(tmp_path, monkeypatch):
    other_asm = list(ASM); other_asm[11] = "131"
    other_row = list(ROW); other_row[1] = "190"
    _setup(tmp_path, monkeypatch, {"r05_traffic.json": _profile()})
    v, why = bench.traffic_from_profile(KERNEL, 99000, {"segments": 3})                   # CEED_MI355X_PIPE_MB=90, _ASSEMBLE=serial, ...
    assert v is None and "other assembly form" in why
    for bad, word in ((_profile(assemble_isa=other_asm), "k_assemble"), (_profile(kernel_isa=other_row), "another build of this kernel")):
        (tmp_path / "profiles" / "r05_traffic.json").write_text(json.dumps(bad))
        v, why = bench.traffic_from_profile(KERNEL, 99000, {"segments": 2})
        assert v is None and why.startswith("stale") and word in why, why


def test_a_profile_of_another_kernel_is_skipped_not_final(tmp_path, monkeypatch):
    _setup(tmp_path, monkeypatch, {"r06_traffic.json": _profile(kernel="fused_grad<P=5,Q=5,HyperSSdF>/pencil [qdata read]"),
                                  "r05_traffic.json": _profile(hbm_bytes_per_apply=1.0e9)})
    v, src = bench.traffic_from_profile(KERNEL, 99000, {"segments": 2})
    assert v == 1.0e9 and "r05_traffic.json" in src
    os.remove(tmp_path / "profiles" / "r05_traffic.json")
    v, why = bench.traffic_from_profile(KERNEL, 99000, {"segments": 2})
    assert v is None and "r06_traffic.json" in why
