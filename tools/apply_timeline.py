#!/usr/bin/env python3
"""Per-apply timeline from a rocprofv3 --kernel-trace CSV: which kernels one operator apply launches, when each starts
and ends relative to the apply's first kernel, and which of them overlap.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --emulate-rank 3 --of 8 ...
    python3 tools/apply_timeline.py $(find /tmp/kt -name '*kernel_trace.csv') --per-apply 2 --last 50

An apply starts at every `--per-apply`-th dispatch whose name matches `--anchor` (default: the fused operator kernel; a
split-phase apply has two of them) and runs until the next apply's first dispatch.  Only the last `--last` applies
(the timed ones) are averaged.  Times in microseconds."""
import argparse
import csv
import json
import re
import statistics as st


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("cps::", "")
    m = re.match(r"(k_fused_pencil)<(\d+), *(\d+), *(\d+)", name)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)},qf{m.group(4)}>"
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--anchor", default="k_fused_pencil")
    ap.add_argument("--per-apply", type=int, default=1)
    ap.add_argument("--last", type=int, default=50)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
                         int(r.get("Queue_Id", 0) or 0)))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if re.search(a.anchor, r[2])]
    starts = anchors[::a.per_apply]
    if len(starts) < 3:
        raise SystemExit("fewer than three applies in the trace")
    applies = [rows[starts[k]:starts[k + 1]] for k in range(len(starts) - 1)]
    applies = applies[-a.last:]
    # the modal launch sequence (applies interrupted by other work -- the exchange timed alone, say -- are left out)
    sig = lambda ap_: tuple(short(r[2]) for r in ap_)
    modal = max(set(map(sig, applies)), key=lambda s: sum(1 for x in applies if sig(x) == s))
    good = [x for x in applies if sig(x) == modal]
    t0s = [x[0][0] for x in good]
    print(f"{len(good)} of {len(applies)} applies share the modal sequence of {len(modal)} launches")
    print(f"{'#':>2} {'kernel':<62} {'wgs':>6} {'q':>2} {'start':>8} {'end':>8} {'dur':>8}")
    out = []
    for j, name in enumerate(modal):
        s = st.mean((x[j][0] - x[0][0]) / 1e3 for x in good)
        e = st.mean((x[j][1] - x[0][0]) / 1e3 for x in good)
        d = st.mean((x[j][1] - x[j][0]) / 1e3 for x in good)
        print(f"{j:>2} {name:<62} {good[0][j][3]:>6} {good[0][j][4]:>2} {s:8.1f} {e:8.1f} {d:8.1f}")
        out.append({"kernel": name, "workgroups": good[0][j][3], "queue": good[0][j][4], "start_us": s, "end_us": e, "dur_us": d})
    span = [(max(r[1] for r in x) - x[0][0]) / 1e3 for x in good]
    period = [(t0s[k + 1] - t0s[k]) / 1e3 for k in range(len(t0s) - 1) if t0s[k + 1] - t0s[k] < 5 * st.median(span) * 1e3]
    print(f"span first start -> last end: mean {st.mean(span):.1f} us (min {min(span):.1f}, max {max(span):.1f}); "
          f"start-to-start period: mean {st.mean(period):.1f} us" if period else "")
    if a.json:
        json.dump({"applies": len(good), "launches": out, "span_us_mean": st.mean(span), "span_us_min": min(span), "span_us_max": max(span),
                   "period_us_mean": st.mean(period) if period else None}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
