#!/usr/bin/env python3
"""Kernel timeline of the end of a rocprofv3 --kernel-trace CSV: start (us), duration, gap to the previous kernel's end,
queue, name — to see where a launch path leaves the GPU idle.

    python tools/kernel_gaps.py TRACE.csv [--last-ms 30] [--grep iter_warp]
"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last-ms", type=float, default=30.0)
    ap.add_argument("--min-gap", type=float, default=0.0, help="only rows whose gap to the previous kernel is at least this (us)")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    t_end = rows[-1][1]
    sel = [r for r in rows if r[0] > t_end - int(a.last_ms * 1e6)]
    prev = None
    for s, e, n, q in sel:
        gap = (s - prev) / 1e3 if prev else 0.0
        prev = max(prev or e, e)
        if gap >= a.min_gap:
            n = n.replace("(anonymous namespace)::", "").replace("void ", "")
            print(f"{(s - sel[0][0]) / 1e3:10.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  q={q} {n[:70]}")


if __name__ == "__main__":
    main()
