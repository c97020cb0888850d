"""Per-window summary of a rocprofv3 kernel_stats.csv of `bench.py --mode train`: launches, kernel ms, MFMA vs glue.
    python tools/kstats.py gpurun_out/DIR/kernel_stats.csv [WINDOWS] [--top N]"""
import csv
import sys

path = sys.argv[1]
windows = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 4
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
rows = list(csv.DictReader(open(path)))
mfma = lambda n: any(k in n for k in ("conv3x3_halo_kernel", "wgrad3x3_halo_kernel", "gemm_nt_kernel"))
loss = lambda n: any(k in n for k in ("iter_warp", "splat_stats", "chain_bwd", "dflow_splat", "image_count", "loss_reduce", "update_pass", "update_window"))
tot = {"mfma": [0, 0.0], "loss": [0, 0.0], "glue": [0, 0.0]}
for r in rows:
    k = "mfma" if mfma(r["Name"]) else "loss" if loss(r["Name"]) else "glue"
    tot[k][0] += int(r["Calls"])
    tot[k][1] += float(r["TotalDurationNs"]) / 1e6
n = sum(v[0] for v in tot.values())
ms = sum(v[1] for v in tot.values())
print(f"{path}: {windows} windows; per window: {n / windows:.0f} launches, {ms / windows:.2f} ms of kernels")
for k, v in tot.items():
    print(f"  {k:5s} {v[0] / windows:7.0f} launches {v[1] / windows:8.3f} ms")
if top:
    g = sorted((r for r in rows if not mfma(r["Name"]) and not loss(r["Name"])), key=lambda r: -float(r["TotalDurationNs"]))
    for r in g[:top]:
        print(f"    {r['Name'][:90]:90s} {int(r['Calls']) / windows:6.0f} {float(r['TotalDurationNs']) / 1e6 / windows:7.3f} ms {float(r['AverageNs']) / 1e3:7.1f} us")
