import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from taming_event_flow_amd import synth
from taming_event_flow_amd.loss.flow import Iterative
dev = torch.device("cuda:0")
B,H,W,P,F,N = 8,128,128,10,4,10000
cfg = {"loader": {"resolution": [H, W], "batch_size": B}, "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False, "iterative_mode": "two"}, "data": {"passes_loss": P, "scales_loss": 1}}
rng = np.random.default_rng(0)
win = synth.make_window(rng, B, H, W, P, F, N, 0, sigma=2.0)
L = Iterative(cfg, dev)
flows = [[torch.tensor(win["flows"][t][i], device=dev, requires_grad=True) for i in range(F)] for t in range(P)]
for t in range(P):
    L.update(flows[t], torch.tensor(win["ev"][t], device=dev), torch.tensor(win["pm"][t], device=dev), torch.tensor(win["dev"][t], device=dev), torch.tensor(win["dpm"][t], device=dev))
leaves = [f for row in flows for f in row]
def step():
    loss = L()
    return torch.autograd.grad(loss, leaves)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/300:.3f} ms/step, total {1e3*(t2-t0)/300:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
