import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from conftest import *
from oracle import oracle
oracle.build()
worst=0
for name in ITERATIVE_CASES+LINEAR_CASES+FULL_RES_CASES:
    meta, win, loss, dflows = load_case(name)
    w = oracle.Window(win["flows"], win["ev"], win["pm"], win["dev"], win["dpm"], S=meta["S"], mode=meta["mode"],
                      round_ts=meta["round_ts"], loss_scaling=meta.get("loss_scaling", True),
                      border_compensation=meta.get("border_compensation", True))
    l, d = w.loss(meta["kind"], meta["spat"], meta["temp"])
    b=dflows.astype(np.float64); a=d.astype(np.float64)
    plain=(np.abs(a-b)/(1e-4*np.abs(b)+1e-6*np.abs(b).max())).max()
    mass = w.gradient_mass(meta["kind"])
    if meta["spat"] is not None or meta["temp"] is not None: mass = mass + np.abs(w.smoothing(meta["spat"], meta["temp"])[1])
    print(f"{name:24s} loss {abs(l-loss)/abs(loss):.1e} grad {rel_err(d,dflows):.1e} plain-elementwise {plain:.3f} mass-excess {elementwise_excess(d,dflows,mass)[0]:.3f}")
