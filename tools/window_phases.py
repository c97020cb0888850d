#!/usr/bin/env python3
"""Timeline of ONE training window out of a rocprofv3 --kernel-trace CSV of `bench.py --mode train [--graph]`: when the
forward, the loss, BPTT, the deferred weight gradients and the optimiser step run (ms from the window's first kernel), and
per phase how long some kernel / two or more kernels were running.  Windows are delimited by the Adam kernel.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --mode train --graph --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events
    python tools/window_phases.py DIR/*/*_kernel_trace.csv
"""
import csv, sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# window boundaries: adam_clip_step kernel ends a window
ends=[e for s,e,n in rows if "adam" in n.lower()]
print("adam kernels", len(ends))
w0, w1 = ends[-3], ends[-2]
win=[(s,e,n) for s,e,n in rows if s>=w0 and e<=w1+1000]
t0=win[0][0]
def first(pat): 
    x=[s for s,e,n in win if pat in n]; return (min(x)-t0)/1e6 if x else None
def last(pat):
    x=[e for s,e,n in win if pat in n]; return (max(x)-t0)/1e6 if x else None
print("window span ms", (win[-1][1]-t0)/1e6, "kernels", len(win))
for pat in ["iter_warp","splat_stats","iter_chain_bwd","dflow_splat","wgrad3x3","l2_norm","adam","pack_weight","pack_halo","upsample","conv3x3_halo"]:
    print(pat, first(pat), last(pat))
# busy analysis per phase
def busy(a,b):
    pts=[]
    for s,e,n in win:
        s2=max(s,t0+int(a*1e6)); e2=min(e,t0+int(b*1e6))
        if e2>s2: pts+= [(s2,1),(e2,-1)]
    pts.sort(); d=0; last=None; one=two=0
    for t,k in pts:
        if d>=1: one+=t-last
        if d>=2: two+=t-last
        d+=k; last=t
    return one/1e6, two/1e6
fw_end=first("iter_warp"); bw_start=last("dflow_splat"); 
wg=[ (s,e) for s,e,n in win if "wgrad3x3" in n]
print("forward phase 0..%.2f: busy %.2f multi %.2f" % ((fw_end,)+busy(0,fw_end)))
bw_end=max(e for s,e,n in win if "cell_bwd" in n); bw_end=(bw_end-t0)/1e6
print("bptt phase %.2f..%.2f: busy %.2f multi %.2f" % ((bw_start,bw_end)+busy(bw_start,bw_end)))
print("tail %.2f..%.2f: busy %.2f multi %.2f" % ((bw_end,(win[-1][1]-t0)/1e6)+busy(bw_end,(win[-1][1]-t0)/1e6)))
tot_wg=sum(e-s for s,e in wg)/1e6
print("wgrad kernel time total", tot_wg, "of which after bptt end", sum(max(0,e-max(s,t0+int(bw_end*1e6))) for s,e in wg)/1e6)
