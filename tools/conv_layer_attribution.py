"""Per-layer attribution of one training window (BASELINE configs[2]: B = 8, 128 x 128, P = 10), round 6 / VERDICT item 3:
every layer and direction of RecEVFlowNet bracketed by HIP events inside the C walker (tef_profile_layers), the window run
eagerly on ONE stream so that an event pair measures the layer's own launches (convolutions + the reduce / activation launches
that belong to it).  Writes a CSV: label, scopes, ms per window, GFLOP per window, TFLOP/s, fraction of the fp32 MFMA peak,
ms lost against that peak.

    python tools/conv_layer_attribution.py [OUT.csv]
"""
import copy
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from taming_event_flow_amd import _lib, train  # noqa: E402

PEAK = 157.3      # TFLOP/s, v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/conv_layer_attribution.csv"
dev = torch.device("cuda:0")
cfg = copy.deepcopy(train.DEFAULT_CONFIG)
B, H, W, P = cfg["loader"]["batch_size"], *cfg["loader"]["resolution"], cfg["data"]["passes_loss"]
tr = train.Trainer(cfg, dev, streams=False)
src = train.SyntheticSequences(cfg, dev, 10000, seq_len=10 ** 9, seed=1)
tr.reset()
lib = _lib.lib()
for w in range(3):
    if w == 2:
        torch.cuda.synchronize()
        lib.tef_profile_enable(1)
    for _ in range(P):
        tr.step(src.next(), new_seq=False)
torch.cuda.synchronize()
lib.tef_profile_collect()
n = lib.tef_profile_layers(None, 0)
buf = ctypes.create_string_buffer(n + 1)
lib.tef_profile_layers(buf, n + 1)
lib.tef_profile_enable(0)

# 2 M N K per pass of every convolution
width, bins = [64, 128, 256, 512], 2
flops = {}
cin, h, w = bins, H, W
for i, c in enumerate(width):
    h, w = h // 2, w // 2
    m = B * h * w
    flops[f"enc{i}.head"] = 2.0 * m * c * 9 * cin
    flops[f"enc{i}.gru.ur"] = 2.0 * m * 2 * c * 9 * 2 * c
    flops[f"enc{i}.gru.og"] = 2.0 * m * c * 9 * 2 * c
    flops[f"enc{i}.gru"] = flops[f"enc{i}.gru.ur"] + flops[f"enc{i}.gru.og"]
    cin = c
for j in range(2):
    for k in (1, 2):
        flops[f"res{j}.conv{k}"] = 2.0 * B * h * w * 512 * 9 * 512
srcc, outs = 512, [256, 128, 64, 32]
for k, o in enumerate(outs):
    h, w = h * 2, w * 2
    cin = srcc + (2 if k else 0)
    flops[f"dec{k}"] = 2.0 * B * h * w * o * 9 * cin
    flops[f"pred{k}"] = 2.0 * B * h * w * 2 * o
    srcc = o

rows = []
for line in buf.value.decode().strip().splitlines():
    label, scopes, ms = line.rsplit(",", 2)
    name, direction = label.rsplit(" ", 1)
    gf = flops.get(name, 0.0) * P / 1e9 if direction in ("fwd", "dgrad", "wgrad", "bwd") else 0.0
    if direction == "bwd" and not name.endswith(".gru"):
        gf = 0.0
    ms = float(ms)
    tf = gf / ms if ms > 0 else 0.0      # GFLOP / ms = TFLOP/s
    rows.append((label, int(scopes), ms, gf, tf, tf / PEAK, ms - gf / PEAK))
rows.sort(key=lambda r: -r[6])
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
with open(out_path, "w") as f:
    f.write("label,scopes,ms_per_window,GFLOP_per_window,TFLOP_per_s,frac_of_fp32_mfma_peak,ms_lost_vs_peak\n")
    for r in rows:
        f.write(f"{r[0]},{r[1]},{r[2]:.4f},{r[3]:.2f},{r[4]:.1f},{r[5]:.3f},{r[6]:.4f}\n")
tot_ms, tot_gf = sum(r[2] for r in rows), sum(r[3] for r in rows)
print(f"{len(rows)} labels, {tot_ms:.2f} ms in labelled layers per window, {tot_gf / 1e3:.3f} TFLOP, {tot_gf / tot_ms:.1f} TFLOP/s")
for r in rows[:16]:
    print(f"  {r[0]:22s} {r[2]:7.3f} ms {r[3]:8.1f} GF {r[4]:6.1f} TF/s  lost {r[6]:6.3f} ms")
