import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, copy
t0 = time.time()
def T(msg):
    torch.cuda.synchronize(); print(f"{time.time()-t0:8.2f}s {msg}", flush=True)
from taming_event_flow_amd import train
dev = torch.device("cuda:0")
cfg = copy.deepcopy(train.DEFAULT_CONFIG)
cfg["loader"]["batch_size"] = 2; cfg["loader"]["resolution"] = [32, 32]; cfg["data"]["passes_loss"] = 3
T("import")
tr = train.Trainer(cfg, dev); T("trainer built")
src = train.SyntheticSequences(cfg, dev, 380, seq_len=100); T("source built")
for it in range(6):
    inp = src.next(); T(f"batch {it}")
    x = tr.model(inp["net_input"]); T("  model fwd")
    flows = [f * 32 for f in x["flow"]]
    tr.loss_function.update(flows, inp["event_list"], inp["event_list_pol_mask"], inp["d_event_list"], inp["d_event_list_pol_mask"]); T("  update")
    if tr.loss_function.num_passes >= 3:
        loss = tr.loss_function(); T("  loss fwd")
        loss.backward(); T("  backward")
        tr.bucket.all_reduce_sum(); n = tr.bucket.clip_(100.0); T("  clip")
        tr.optimizer.step(); T("  adam")
        tr.bucket.zero(); tr.model.detach_states(); tr.loss_function.reset(); T("  reset")
