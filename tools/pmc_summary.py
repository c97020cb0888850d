#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; counter_collection.csv each) into the per-launch
HBM traffic summary bench.py reads (profiles/rNN_pmc_traffic.json).

    python tools/pmc_summary.py FETCH.csv WRITE.csv OUT.json [--skip W]

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of the bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so traffic = 2 * FETCH_SIZE + WRITE_SIZE.  The first `--skip`
launches of each kernel (warm-up passes) are dropped, the rest averaged.
"""
import argparse
import csv
import json
from collections import defaultdict

KERNELS = {                      # kernel symbol prefix -> bench.py name (first match wins)
    "chain_bwd_finish_kernel": "mag_reduce",
    "iter_warp_kernel": "warp", "linear_warp_kernel": "warp", "splat_stats_kernel": "iwe_splat",
    "loss_reduce_kernel": "loss_reduce", "iter_chain_bwd_kernel": "chain_bwd",
    "linear_bwd_kernel": "chain_bwd", "dflow_splat_kernel": "dflow_splat", "pack_flow_kernel": "pack_flow",
    "grad_planes_kernel": "grad_planes", "image_count_kernel": "image_count", "mag_reduce_kernel": "mag_reduce",
}


def per_kernel(path, counter, skip):
    vals = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            sym = row["Kernel_Name"]
            for pre, name in KERNELS.items():
                if "::" + pre + "(" in sym or "::" + pre + "<" in sym or sym.startswith(pre) or ("<" in pre and pre in sym):     # plain or templated
                    vals[name].append(float(row["Counter_Value"]) * 1024.0)
                    break
    return {k: (sum(v[skip:]) / max(1, len(v[skip:])), len(v[skip:])) for k, v in vals.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("out_json")
    ap.add_argument("--skip", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--passes", type=int, default=10)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--events", type=int, default=10000)
    ap.add_argument("--detached", type=int, default=0)
    ap.add_argument("--res", type=int, nargs=2, default=[128, 128])
    ap.add_argument("--warping", default="Iterative")
    a = ap.parse_args()
    fetch = per_kernel(a.fetch_csv, "FETCH_SIZE", a.skip)
    write = per_kernel(a.write_csv, "WRITE_SIZE", a.skip)
    kernels = {}
    for k in sorted(set(fetch) & set(write)):
        f, nf = fetch[k]
        w, _ = write[k]
        kernels[k] = {"FETCH_SIZE_bytes_raw": int(f), "WRITE_SIZE_bytes": int(w), "traffic_bytes": int(2 * f + w),
                      "launches": nf}
    doc = ("HBM-side bytes per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes).  "
           "Counters are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 "
           "(calibration: pack_flow reads 1 MiB coalesced, writes 2 MiB).  traffic = 2*FETCH_SIZE + WRITE_SIZE.  "
           "Made by tools/pmc_summary.py.")
    out = {"_doc": doc,
           "config": {"batch": a.batch, "passes": a.passes, "heads": a.heads, "events": a.events,
                      "detached": a.detached, "res": a.res, "warping": a.warping},
           "kernels": kernels}
    with open(a.out_json, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in kernels.items():
        print(f"{k:12s} fetch*2 {2 * v['FETCH_SIZE_bytes_raw'] / 1e6:9.1f} MB  write {v['WRITE_SIZE_bytes'] / 1e6:9.1f} MB"
              f"  traffic {v['traffic_bytes'] / 1e6:9.1f} MB  ({v['launches']} launches)")


if __name__ == "__main__":
    main()
