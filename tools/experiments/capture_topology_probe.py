"""Which stream topologies does hipStreamEndCapture survive on this stack?  (round 6: the pipelined encoder levels.)
    python tools/experiments/capture_topology_probe.py MODE"""
import sys, torch
mode = sys.argv[1]
dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev); b = torch.zeros(1 << 20, device=dev); c = torch.zeros(1 << 20, device=dev)
low, high = torch.cuda.Stream(), torch.cuda.Stream()
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
def body():
    main = torch.cuda.current_stream()
    for t in range(P):
        low.wait_stream(main)
        with torch.cuda.stream(low):
            a.add_(1.0)
        high.wait_stream(main)
        high.wait_stream(low)
        with torch.cuda.stream(high):
            b.add_(1.0)
    main.wait_stream(low); main.wait_stream(high)
    c.add_(1.0)
    for t in range(P):
        if "no_main" not in mode:
            high.wait_stream(main)
        with torch.cuda.stream(high):
            b.add_(1.0)
        if "via_main" in mode:
            main.wait_stream(high)
            low.wait_stream(main)
        elif "no_cross" not in mode:
            low.wait_stream(high)
        if "no_main" not in mode:
            low.wait_stream(main)
        if "high_only" not in mode:
            with torch.cuda.stream(low):
                a.add_(1.0)
    main.wait_stream(low); main.wait_stream(high)
    c.add_(1.0)
body(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay(); torch.cuda.synchronize()
print(mode, P, "ok", float(a[0]), float(b[0]), float(c[0]))
