import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from conftest import load_bench_window, rel_err
from oracle import oracle
for name in ("bench_window_0","bench_window_1"):
    meta, win, gold = load_bench_window(name)
    w = oracle.Window(win["flows"], win["ev"], win["pm"], win["dev"], win["dpm"], S=1, mode="two")
    l, d = w.loss("Iterative", None, None)
    lat = d[..., ::4, ::4]; ref = gold["dflows_lattice"]
    diff = np.abs(lat.astype(np.float64)-ref)
    print(name, "loss rel", abs(l-float(gold["loss"]))/float(gold["loss"]), "gmax", np.abs(ref).max())
    idx = np.argsort(diff.ravel())[::-1][:10]
    for i in idx:
        u = np.unravel_index(i, diff.shape)
        print(u, diff[u], lat[u], ref[u], "mapmax", gold["dflows_max"][u[:4]])
    print("quantiles", np.quantile(diff/np.abs(ref).max(), [0.5,0.9,0.99,0.999,0.9999,1]))
    mass = w.gradient_mass("Iterative")[..., ::4, ::4]
    ex = diff/(1e-4*mass+1e-7*np.abs(ref).max())
    print("elementwise excess max", ex.max(), "n>1", (ex>1).sum(), "n>4", (ex>4).sum())
    np.save(f"scratch/{name}_oracle_d.npy", d)
