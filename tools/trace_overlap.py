#!/usr/bin/env python3
"""How busy is the GPU in a multi-stream run?  From a rocprofv3 --kernel-trace CSV: the span of the trace's last
`--tail` seconds, the union of the kernel intervals (some kernel running), the time two or more kernels ran together, the
idle time, and the sum of the kernel durations.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --mode train --graph --steps 5 --warmup 2 ...
    python tools/trace_overlap.py DIR/*/*_kernel_trace.csv [--tail 0.15]
"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail", type=float, default=0.15, help="analyse the last TAIL seconds of the trace")
    a = ap.parse_args()
    ev = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    ev.sort()
    t_end = max(e for _, e in ev)
    t0 = t_end - int(a.tail * 1e9)
    ev = [(max(s, t0), e) for s, e in ev if e > t0]
    pts = []
    for s, e in ev:
        pts.append((s, 1))
        pts.append((e, -1))
    pts.sort()
    busy = multi = 0
    depth, last = 0, pts[0][0]
    for t, d in pts:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    span = t_end - min(s for s, _ in ev)
    total = sum(e - s for s, e in ev)
    print(f"span {span / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), two or more "
          f"{multi / 1e6:.2f} ms ({100 * multi / span:.1f} %), idle {(span - busy) / 1e6:.2f} ms ({100 * (span - busy) / span:.1f} %), "
          f"sum of kernel durations {total / 1e6:.2f} ms, {len(ev)} kernels")


if __name__ == "__main__":
    main()
