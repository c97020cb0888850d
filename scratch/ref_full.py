import sys, os, numpy as np
sys.argv=[sys.argv[0]]
sys.path.insert(0,'/root/repo/tests/golden')
import make_golden as mg, torch
torch.set_num_threads(8)
from taming_event_flow_amd import synth
for seed in (0,1):
    rng=np.random.default_rng(seed)
    win=synth.make_window(rng,8,128,128,10,4,10000,0,sigma=2.0,kind="smooth")
    cfg=mg.make_config(128,128,8,10,1,"two")
    l64,l32,g=mg.run_loss("Iterative",cfg,win)
    np.save(f"/root/repo/scratch/ref_full_{seed}.npy",g)
    print(seed,l64)
