#!/usr/bin/env python3
"""isa_guard.py -- check the ISA-level assumptions of the pencil kernel after a build (runs on the CPU).

The pencil kernel (csrc/kernel_fused_pencil.hpp) hands data from one pass to the next through LDS with
nothing but the in-order LDS queue of ONE wave between them, and it relies on the compiler keeping every
volatile LDS access a separate ds_read_b64 / ds_write_b64 (the two-address forms run at half the byte
rate on gfx950), on a spill-free register allocation at two waves per SIMD, and on there being no
workgroup barrier.  None of this is promised by the language, so the build checks the code object:

  * no ds_read2_b64 / ds_write2_b64 / ds_read_b128 / ds_write_b128 in a pencil kernel
  * no s_barrier in a pencil kernel
  * no scratch: private_segment_fixed_size == 0, no scratch_* / buffer_* ... offen instructions,
    no VGPR or SGPR spills
  * at most 256 VGPRs (two waves per SIMD) for the kernels of the metric configurations

usage: isa_guard.py build/fused_q5p0.o [build/fused_q7p0.o ...] [--summary out.txt]
Exit status 1 on a violation in a guarded kernel (k_fused_pencil<P,Q,...>; Q = 8 -- degree 7, one wave per SIMD by design -- is held to
the same LDS / barrier / scratch rules with a 512-register limit).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"      # --llvm
ARCH = "gfx950"                      # --arch
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
BAD_LDS = ("ds_read2_b64", "ds_write2_b64", "ds_read_b128", "ds_write_b128", "ds_read2st64_b64", "ds_write2st64_b64")


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def code_object(obj, tmp):
    fat = os.path.join(tmp, os.path.basename(obj) + ".fatbin")
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    run(f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat)
    run(f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}")
    return co


def kernel_meta(co):
    """{mangled name: {vgpr, sgpr, lds, scratch, vspill, sspill}} from the code object's metadata note."""
    txt = run(f"{LLVM}/llvm-readelf", "--notes", co)
    keys = {"vgpr_count": "vgpr", "sgpr_count": "sgpr", "group_segment_fixed_size": "lds",
            "private_segment_fixed_size": "scratch", "vgpr_spill_count": "vspill", "sgpr_spill_count": "sspill"}
    out = {}
    for block in re.split(r"(?m)^  - (?=\.agpr_count)", txt)[1:]:
        d = {}
        for m in re.finditer(r"(?m)^    \.(\w+):\s+(\S+)\s*$", block):   # the kernel's own keys: four spaces of indent
            if m.group(1) == "name":
                d["name"] = m.group(2)
            elif m.group(1) in keys:
                d[keys[m.group(1)]] = int(m.group(2))
        if "name" in d:
            out[d["name"]] = d
    return out


QF_NAMES = {2: "LinElas", 3: "HyperSSF", 4: "HyperSSdF", 5: "HyperFSF", 6: "HyperFSdF", 17: "HyperFSdF+derived"}


def other_name(mangled):
    """Kernels of kernels_misc.hip that ride along in the summary WITHOUT being guarded: the restriction transpose (half of the apply's
    measured traffic: bench.py reports a PMC profile only while this row is unchanged too), its epilogue form, the transfer kernels."""
    if re.search(r"\d+k_assembleE", mangled):
        return "k_assemble"
    if re.search(r"\d+k_assemble_epiE", mangled):
        return "k_assemble_epi"
    m = re.search(r"\d+k_transferILi(\d+)ELi(\d+)ELb(\d)ELb(\d)EE", mangled)
    if m:
        return "k_transfer<Pc=%s,Pf=%s,%s%s>" % (m.group(1), m.group(2), "prolong" if m.group(3) == "1" else "restrict", ",weighted" if m.group(4) == "1" else "")
    return None


def short_name(mangled):
    m = re.search(r"k_fused_pencilILi(\d+)ELi(\d+)ELi(\d+)ELi(\d)E", mangled)
    if not m:
        o = other_name(mangled)
        if o:
            return o, 99, 0          # q = 99: listed, not guarded
        return None, 0, 0
    P, Q, qf, geo = (int(x) for x in m.groups())
    eo = 1 if 4 <= Q <= 7 else 0     # kernels.hpp, pencil_even_odd
    return f"k_fused_pencil<P={P},Q={Q},{QF_NAMES.get(qf, qf)},geo={geo},eo={eo}>", Q, eo


def instruction_counts(co):
    """{mangled name: {mnemonic: count}} from the disassembly."""
    txt = run(f"{LLVM}/llvm-objdump", "-d", f"--mcpu={ARCH}", co)
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\w+)", line)
        if m:
            cur[m.group(1)] = cur.get(m.group(1), 0) + 1
    return out


def main():
    global LLVM, ARCH, TARGET
    ap = argparse.ArgumentParser()
    ap.add_argument("objects", nargs="+")
    ap.add_argument("--summary")
    ap.add_argument("--arch", default=ARCH)
    ap.add_argument("--llvm", default=LLVM)
    args = ap.parse_args()
    LLVM, ARCH, TARGET = args.llvm, args.arch, f"hipv4-amdgcn-amd-amdhsa--{args.arch}"
    rows, bad = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in args.objects:
            co = code_object(obj, tmp)
            meta, ins = kernel_meta(co), instruction_counts(co)
            for name, m in sorted(meta.items()):
                short, q, eo = short_name(name)
                if short is None:
                    continue
                ic = ins.get(name, {})
                n_bad_lds = sum(ic.get(k, 0) for k in BAD_LDS)
                n_scratch = sum(v for k, v in ic.items() if k.startswith("scratch_"))
                n_barrier = ic.get("s_barrier", 0)
                n_valu = sum(v for k, v in ic.items() if k.startswith("v_"))
                n_lane = ic.get("v_readlane_b32", 0) + ic.get("v_writelane_b32", 0)
                rows.append((short, m.get("vgpr", -1), m.get("sgpr", -1), m.get("lds", -1), m.get("scratch", -1),
                             m.get("vspill", 0), m.get("sspill", 0), n_bad_lds, n_barrier, n_scratch, n_lane, n_valu,
                             ic.get("ds_read_b64", 0), ic.get("ds_write_b64", 0)))
                if q <= 8:   # guarded: every level of the metric configurations (p <= 6) and, with its own register limit, the Q = 8 ladders (degree 7)
                    why = []
                    if n_bad_lds: why.append(f"{n_bad_lds} two-address / 128-bit LDS instructions")
                    if n_barrier: why.append(f"{n_barrier} s_barrier")
                    if n_scratch or m.get("scratch", 0): why.append(f"scratch ({m.get('scratch', 0)} B, {n_scratch} instructions)")
                    # SGPR spills go to VGPR lanes (v_writelane / v_readlane): slower, not wrong.  The default (even-odd) instantiations
                    # up to Q = 5 are held to none (from Q = 6 one table fills the SGPR file); the plain-table fallback (eo=0, taken only for tables that are not centro-symmetric) has a few.
                    sgpr_spill_limit = 8 if ",fold" in short else 4   # (the opt-in folded form: a few more)
                    if m.get("vspill", 0) or (eo and q <= 5 and m.get("sspill", 0) > sgpr_spill_limit):
                        why.append(f"spills (vgpr {m.get('vspill', 0)}, sgpr {m.get('sspill', 0)})")
                    if q <= 7 and m.get("vgpr", 0) > 256: why.append(f"{m['vgpr']} VGPRs > 256 (one wave per SIMD)")
                    if q == 8 and m.get("vgpr", 0) > 512: why.append(f"{m['vgpr']} VGPRs > 512")
                    if why:
                        bad.append(f"{short}: " + "; ".join(why))
    hdr = ("kernel", "vgpr", "sgpr", "lds_B", "scratch_B", "vspill", "sspill", "lds2/128", "s_barrier", "scratch_ins",
           "v_read/writelane", "valu_ins", "ds_read_b64", "ds_write_b64")
    lines = ["\t".join(hdr)] + ["\t".join(str(x) for x in r) for r in rows]
    lines.append("")
    lines.append("VIOLATIONS: " + ("none" if not bad else ""))
    lines += bad
    text = "\n".join(lines) + "\n"
    if args.summary:
        with open(args.summary, "w") as f:
            f.write(text)
    if bad:
        sys.stderr.write("isa_guard: the pencil kernel's ISA contract is broken:\n  " + "\n  ".join(bad) + "\n")
        return 1
    print(f"isa_guard: {len(rows)} pencil kernels checked, contract holds" + (f" (summary: {args.summary})" if args.summary else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
