#!/usr/bin/env python3
"""Can this RCCL's point-to-point calls be recorded into a hipGraph?  Each attempt runs in a CHILD process (the failure
mode found in round 3 is a segmentation fault of the host process inside the capture, RCCL 2.26.6 as shipped with
torch 2.10 + rocm 7.0).  Prints one line per form: inline (RCCL on the capturing stream) / own stream (joined by events).
Round 4: the crash's C backtrace was taken with tools/segv_backtrace.c (profiles/r04_rccl_capture_probe.txt: a stack overflow in
libamdhip64.so at hipStreamEndCapture); the library now refuses the own-stream form while recording unconditionally, so the second
child is expected to end with that CeedError -- the crash itself is reproduced without this library by
tools/microbench/capture_fork_repro.hip.

    python3 tools/rccl_capture_probe.py            # parent: spawns the children BEFORE touching the GPU itself
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(inline: str):
    sys.path.insert(0, ROOT)
    os.environ["CEED_MI355X_COMM_INLINE"] = inline
    import ctypes as C
    import numpy as np
    import torch  # noqa: F401  (loads the HIP runtime and RCCL this library binds to)
    bt = os.path.join(ROOT, "tools", "libsegv_bt.so")   # (round 4) C backtrace of the crash: module(+offset) per frame on stderr
    from ceedpetscsolid_amd import ceed as cd
    ceed = cd.Ceed(cd.CeedLib(cd.PRODUCT_LIB), "/gpu/hip/mi355x")
    L = ceed.L
    ident = C.create_string_buffer(128)
    L.chk(L.lib.CeedXCommGetUniqueId(ceed.h, ident))
    L.chk(L.lib.CeedXCommInit(ceed.h, 1, 0, ident))
    n = 100_000
    idx = np.arange(0, n, 3, dtype=np.int32)
    h = C.c_void_p()
    L.chk(L.lib.CeedXHaloCreate(ceed.h, 1, (C.c_int * 1)(0), (C.c_int * 1)(idx.size), (C.POINTER(C.c_int) * 1)(idx.ctypes.data_as(C.POINTER(C.c_int))), C.byref(h)))
    y0 = np.random.default_rng(0).uniform(-1, 1, n)
    Y = ceed.vector(n).set_array(y0)
    Y.device_pointer()

    def exchange():
        L.chk(L.lib.CeedXHaloStart(h, Y.h)); L.chk(L.lib.CeedXHaloFinish(h, Y.h))
    exchange()
    print("eager exchange ok", flush=True)
    if os.path.exists(bt):      # installed LAST (the runtimes loaded above may set handlers of their own), on a stack of its own
        C.CDLL(bt).segv_backtrace_install()
    g = ceed.capture(exchange)
    print("captured", flush=True)
    g.launch(); ceed.synchronize()
    want = y0.copy()
    for _ in range(2):
        want[idx] += want[idx]
    print("replayed: result", "correct" if np.array_equal(Y.to_numpy(), want) else "WRONG", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for inline in ("1", "0"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), inline], capture_output=True, text=True, timeout=300)
        last = [l for l in p.stdout.splitlines() if l and not l.startswith(("RCCL", "HIP", "ROCm", "Hostname", "Librccl"))]
        print(f"COMM_INLINE={inline}: exit {p.returncode} ({'signal ' + str(-p.returncode) if p.returncode < 0 else 'ok' if p.returncode == 0 else 'error'}); "
              f"got as far as: {last[-1] if last else '(nothing)'}; stderr tail: {p.stderr.strip().splitlines()[-1][:200] if p.stderr.strip() else ''}")
        if "=== SIGSEGV backtrace" in p.stderr:
            print(p.stderr[p.stderr.index("=== SIGSEGV backtrace"):])
