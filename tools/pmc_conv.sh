#!/bin/bash
# MFMA utilisation of the convolution kernels from counters (not flops / time): one rocprofv3 --pmc pass per counter over ONE
# training window on one stream, and the same passes over a bare fp32-MFMA loop (tools/mfma_f32_peak.hip) as the 100 % mark.
#   tools/pmc_conv.sh OUTDIR
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$1
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o $O/mfma_f32_peak tools/mfma_f32_peak.hip 2> $O/build.err
# (round 6: + what the wavefronts wait for — LDS, any memory — and what they issue besides MFMA)
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT; do
  TEF_TWO_STREAMS=0 TEF_WGRAD_GROUP=0 timeout -k 5 300 rocprofv3 --pmc $C --kernel-include-regex "halo_kernel|gemm_nt_kernel" --output-format csv -d $O/conv_$C -- python3 bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline > $O/conv_$C.json 2> $O/conv_$C.err
  timeout -k 5 100 rocprofv3 --pmc $C --output-format csv -d $O/peak_$C -- $O/mfma_f32_peak > $O/peak_$C.out 2> $O/peak_$C.err
done
python3 - <<PY
import csv, glob, collections, json
def load(tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in glob.glob("$O/%s_*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:72]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
    return acc, n
peak, _ = load("peak")
conv, n = load("conv")
out = {"calibration": {k: dict(v) for k, v in peak.items()}, "kernels": {}}
ref = None
for k, v in peak.items():
    if v.get("GRBM_GUI_ACTIVE"):
        ref = v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["GRBM_GUI_ACTIVE"]
out["mfma_busy_per_gui_active_at_full_rate"] = ref
tot_b = tot_g = 0.0
for k, v in sorted(conv.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    g = v.get("GRBM_GUI_ACTIVE", 0)
    if not g: continue
    u = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / g / ref if ref else None
    tot_b += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0); tot_g += g
    out["kernels"][k] = {"launches": n[k], "mfma_utilisation": round(u, 4) if u is not None else None, **{c: x for c, x in v.items()}}
out["all_conv_kernels_mfma_utilisation"] = round(tot_b / tot_g / ref, 4) if ref and tot_g else None
json.dump(out, open("$O/mfma_utilisation.json", "w"), indent=1)
print(json.dumps({k: (v["launches"], v["mfma_utilisation"]) for k, v in list(out["kernels"].items())[:12]}, indent=0))
print("all conv kernels:", out["all_conv_kernels_mfma_utilisation"], "calibration busy/active:", ref)
PY
