#!/usr/bin/env python3
"""bench.py — events/s through the IWE + contrast-maximisation loss (BASELINE.json metric).

A step = one pass of the hot path over one loss window: `loss = L(); loss.backward()` of the
`Iterative` loss (mode "two", scales_loss 1) on B=8 samples x P=10 passes x N=10 000 events at
128x128 with F=4 flow heads — BASELINE.json configs[1].  Inputs (flow maps, event lists) are
resident in HBM and already handed to `update()` when the timed region starts (SURVEY.md §8d:
`update()` is excluded from t and reported separately as `ms_update_per_window`).

Multi-GPU: one process per GPU, the batch dimension shards with no data-path collective (the loss
is a sum over samples, reference loss/flow.py:129); weak scaling, per-GPU B fixed.

Prints ONE JSON line (rank 0).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default: a >= 0.2 s timed region)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU")
    ap.add_argument("--passes", type=int, default=10)
    ap.add_argument("--events", type=int, default=10000, help="grad events per pass per sample")
    ap.add_argument("--detached", type=int, default=0, help="detached events per pass per sample")
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--res", type=int, nargs=2, default=[128, 128])
    ap.add_argument("--flow", default="smooth", choices=["smooth", "iid"])
    ap.add_argument("--warping", default="Iterative", choices=["Iterative", "Linear"])
    ap.add_argument("--mode", default="loss", choices=["loss", "train", "eval", "dropin"],
                    help="loss: IWE + contrast-max loss fwd+bwd (BASELINE.json metric, configs[1]); "
                         "train: full training window, RecEVFlowNet + loss + DP all-reduce + Adam (configs[2]/[3])")
    ap.add_argument("--graph", action="store_true", help="train mode: replay the window from a captured hipGraph")
    ap.add_argument("--windows", type=int, default=2, help="distinct pre-staged windows cycled through")
    ap.add_argument("--presort", default="none", choices=["none", "y", "pol_y", "pol_tile8", "pol_tile16", "pol_yx", "pol_y4"],
                    help="experiment: pre-sort the synthetic events of each pass on the host")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--graph-steps", type=int, default=4,
                    help="loss mode: consecutive steps captured into ONE hipGraph (a graph launch costs ~40 us of idle GPU, "
                         "which a training window pays once for its whole pass sequence); 1 = a graph per step")
    ap.add_argument("--step-graph", default="auto", choices=["auto", "on", "off"],
                    help="loss mode: replay a captured hipGraph of the step (for hosts too slow to enqueue 0.8 ms steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-extra", action="store_true",
                    help="loss mode at 1 GPU: skip the short training-window measurement appended as `extra`")
    ap.add_argument("--cpu-batch", type=int, default=0, help="samples in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--event-every", type=int, default=50,
                    help="per-kernel HIP events are recorded on every K-th timed step: that step is launched eagerly (~1.1 ms of host "
                         "time against ~0.7 of kernels), so it costs the timed region ~0.4 ms")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline sample: repeat the window this long")
    return ap.parse_args()


def make_cfg(a):
    return {
        "loader": {"resolution": list(a.res), "batch_size": a.batch},
        "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False,
                 "iterative_mode": "two", "warping": a.warping},
        "data": {"passes_loss": a.passes, "scales_loss": 1},
    }


def pmc_traffic(a):
    """HBM bytes per launch measured with rocprofv3 PMC counters (newest profiles/rNN_pmc_traffic.json), only when the
    benchmark runs the exact configuration they were collected on."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not found:
        return {}
    path = found[-1]
    with open(path) as f:
        d = json.load(f)
    c = d["config"]
    same = (c["batch"] == a.batch and c["passes"] == a.passes and c["heads"] == a.heads and c["events"] == a.events
            and c["detached"] == a.detached and list(c["res"]) == list(a.res) and c["warping"] == a.warping)
    return {k: v["traffic_bytes"] for k, v in d["kernels"].items()} if same else {}


def algorithmic_bytes(a, delta):
    """Per-launch algorithmic bytes of each kernel for the Iterative window.

    An event of pass t is only looked at (splat, gradient sweep, flow-gradient splat) at reference times within
    delta of t, so trajectory planes / per-map vectors outside that reach are neither written nor read."""
    B, P, F, N, Nd = a.batch, a.passes, a.heads, a.events, a.detached
    H, W = a.res
    HW, FB = H * W, F * B
    pairs = 0           # (tref, bin) pairs of one window = image/bin incidences
    for tref in range(P + 1):
        pairs += max(0, min(P, tref + delta) - max(0, tref - delta))
    planes = sum(min(P, t + delta) - max(0, t - delta) + 1 for t in range(P))          # trajectory planes kept, over bins
    vecs = sum(min(P - 1, t + delta - 1) - max(0, t - delta + 1) + 1 for t in range(P))  # (bin, map) pairs with a vector
    nimg = P + 1
    splats = pairs * (N + Nd) * FB
    maps = P * FB * 2 * HW * 4
    return {
        # 16 B per event-splat (position 8 + timestamp 4 + mask word 4)  [SURVEY.md §8d]; the kernel also turns its
        # images into (A, R) and the focus-loss sums (what image_stats did): that 8 B x 2 x HW write-out per image is
        # NOT counted here (roofline_scatter reports it separately)
        "iwe_splat": splats * 16,
        # per (event, head): 12 B event + 8 B masks in, kept positions + 1 meta word out; flow maps once
        "warp": FB * (N + Nd) * (P * (20 + 4) + planes * 8) + maps,
        # per (grad event, head): kept positions in, one 8-byte vector per reachable map out; (A,R) images + flow maps once
        "chain_bwd": FB * N * (P * 24 + planes * 8 + vecs * 8) + nimg * FB * 2 * HW * 8 + maps,
        # per (grad event, head, reachable map): vector 8 + position 8; gradient maps written once
        "dflow_splat": FB * N * vecs * 16 + maps,
    }, splats


def launch_ranks(a):
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks as CHILD processes (the same command
    the driver uses: torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) before anything in this process
    has touched the GPU, pass their output through (rank 0 prints the JSON line) and exit with their code.  The parent
    never initialises HIP and never re-execs."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    return subprocess.call(cmd, env=env)


def init_ranks(a):
    """-> (torch, dist | None, device, rank, world).  One process per GPU; backend "nccl" = RCCL over xGMI.
    TEF_BENCH_BACKEND=gloo + TEF_BENCH_SHARE_GPU=1 let the multi-rank control flow be exercised on a 1-GPU box."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with "
                         f"torch.distributed.run --nproc-per-node {a.gpus} (or plain `python bench.py --gpus {a.gpus}`)")
    if os.environ.get("TEF_BENCH_LAUNCH_ONLY") == "1":      # launcher test hook (tests/test_bench_launch.py): no GPU needed
        os.write(1, (json.dumps({"launch_only": True, "rank": rank, "world": world, "local": local}) + "\n").encode())
        raise SystemExit(0)
    import torch

    ndev = torch.cuda.device_count()
    share = os.environ.get("TEF_BENCH_SHARE_GPU") == "1"
    if ndev < 1 or (world > ndev and not share):
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, {ndev} visible")
    idx = local % ndev if share else local
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TEF_BENCH_BACKEND", "nccl")
        # fail fast: a broken RCCL bring-up must not sit in a metered lease for the default 10 minutes
        from datetime import timedelta

        tmo = timedelta(seconds=float(os.environ.get("TEF_BENCH_DIST_TIMEOUT", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        assert dist.get_world_size() == a.gpus and dist.get_rank() == rank
    return torch, dist, dev, rank, world


def release_now(torch):
    """Collect what a finished phase left behind (trainers with their streams and hipGraphs, autograd records holding device
    arenas) with the device idle, instead of at whatever allocation wakes the cycle collector (tests/conftest.py tells why)."""
    import gc

    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def watchdog():
    """(Re-)arm the progress watchdog.  A rank that stops making progress (a wedged collective, a deadlocked stream) must
    not sit in a metered lease: TEF_BENCH_WATCHDOG_S seconds (default 600) after the last call every thread's stack goes to
    stderr and the process exits non-zero, which takes the launcher and the other ranks down with it.  Called at the phase
    boundaries of a run (rendezvous, staging, warm-up, timed region, extras)."""
    import faulthandler

    world = int(os.environ.get("WORLD_SIZE", "1"))
    default = "600" if world == 1 else "180"            # N > 1: a wedged collective costs a metered 8-GPU lease
    faulthandler.dump_traceback_later(float(os.environ.get("TEF_BENCH_WATCHDOG_S", default)), exit=True)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    watchdog()
    torch, dist, dev, rank, world = init_ranks(a)
    watchdog()

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if dist:
        dist.barrier()
    from taming_event_flow_amd import _lib, synth
    from taming_event_flow_amd.loss.flow import Iterative, Linear

    lib = _lib.lib()
    if a.mode == "train":
        return bench_train(a, torch, dist, dev, rank, world, lib)
    if a.mode == "eval":
        return bench_eval(a, torch, dist, dev, rank, world, lib)
    if a.mode == "dropin":
        if os.environ.get("TEF_BENCH_DROPIN_STREAM", "0") == "1":      # (experiment: the caller's stream is not the null stream)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                out = dropin_extra(a, torch, dev, windows=max(3, min(a.steps, 20)))
        else:
            out = dropin_extra(a, torch, dev, windows=max(3, min(a.steps, 20)))
        if rank == 0:
            print(json.dumps({"metric": "ms per training window, literal train_flow.py loop", "unit": "ms", "n_gpus": 1,
                              "higher_is_better": False, "data": "synthetic", "value": out.get("dropin_window_ms"),
                              "config": {"workload": "reference loop body (train_flow.py:83-137) on this package's modules, "
                                                     f"{a.res[0]}x{a.res[1]} B={a.batch} P={a.passes} N={a.events}"},
                              "extra": out}))
        return 0
    H, W = a.res
    B, P, F = a.batch, a.passes, a.heads
    cfg = make_cfg(a)
    cls = Iterative if a.warping == "Iterative" else Linear

    # ---- stage `windows` distinct loss windows (untimed): synthetic inputs -> HBM -> update() -----------------
    staged, host_windows = [], []
    t_updates = []
    t_updates_deferred = []
    for wi in range(a.windows):
        rng = np.random.default_rng(1000 * rank + wi)
        win = synth.make_window(rng, B, H, W, P, F, a.events, a.detached, sigma=2.0, kind=a.flow)
        if a.presort != "none":
            for t in range(P):
                for key_e, key_m in (("ev", "pm"), ("dev", "dpm")):
                    ev_, pm_ = win[key_e][t], win[key_m][t]
                    for b in range(B):
                        yy, xx, pol = ev_[b, :, 1], ev_[b, :, 2], 1e6 * (ev_[b, :, 3] < 0)
                        k = {"y": yy, "pol_y": yy + pol, "pol_tile8": (yy // 8) * 64 + xx // 8 + pol,
                             "pol_tile16": (yy // 16) * 64 + xx // 16 + pol, "pol_yx": yy * 1024 + xx + pol,
                             "pol_y4": yy // 4 + pol}[a.presort]
                        o = np.argsort(k, kind="stable")
                        ev_[b], pm_[b] = ev_[b][o], pm_[b][o]
        host_windows.append(win)
        flows = [[torch.tensor(win["flows"][t][i], device=dev, requires_grad=True) for i in range(F)] for t in range(P)]
        evs = [(torch.tensor(win["ev"][t], device=dev), torch.tensor(win["pm"][t], device=dev),
                torch.tensor(win["dev"][t], device=dev), torch.tensor(win["dpm"][t], device=dev)) for t in range(P)]
        L = cls(cfg, dev)
        # a throwaway window first (copies of the lists: update() shifts time stamps in place): the module then holds its
        # capacity hints, like the second and every later window of a training loop
        for t in range(P):
            L.update(flows[t], *[x_.clone() for x_ in evs[t]])
        L.reset()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for t in range(P):
            L.update(flows[t], *evs[t])
        e1.record()
        torch.cuda.synchronize()
        t_updates.append((time.perf_counter() - t0, 1e-3 * e0.elapsed_time(e1)))
        staged.append((L, flows))
        # the same window through a deferred update(): P recorded passes + ONE tef_update_window launch (the module's
        # `defer_update`; copies of the lists, the time stamps of the originals are already shifted)
        L2 = cls(cfg, dev)
        L2.defer_update = True
        for rep in range(2):
            copies = [[x_.clone() for x_ in evs[t]] for t in range(P)]
            for c_ in copies:
                c_[0][:, :, 0] -= 0.0      # (touch: the clone kernels have run before the clock starts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for t in range(P):
                L2.update(flows[t], *copies[t])
            t_rec = time.perf_counter() - t0
            L2._flush_updates()
            t_flush = time.perf_counter() - t0 - t_rec
            e1.record()
            torch.cuda.synchronize()
            t_def = (time.perf_counter() - t0, 1e-3 * e0.elapsed_time(e1), t_rec, t_flush)
            L2.reset()
        t_updates_deferred.append(t_def)
    # the first window also pays for one-time costs (code-object load, first allocations): report a warm one
    t_update, t_update_dev = min(t_updates)
    t_update_def, t_update_def_dev, t_update_def_rec, t_update_def_flush = min(t_updates_deferred)

    def step(k):
        # loss forward + backward w.r.t. the F*P flow tensors; autograd.grad hands the gradient
        # views back directly (leaf .grad accumulation would add 40 copy kernels that a training loop never runs:
        # there the flows are network outputs, not leaves)
        L, flows = staged[k % len(staged)]
        loss = L()
        grads = torch.autograd.grad(loss, [f for row in flows for f in row], grad_outputs=one)      # (no ones_like launch per step)
        return loss, grads

    one = torch.ones((), dtype=torch.float32, device=dev)

    # loss and gradient of window 0, once, untimed: the number `cpu_baseline.loss` (the same window through the CPU port) is
    # comparable with, and — when the window is the one tests/golden/make_golden.py --bench-windows recorded from the
    # reference (the default workload on rank 0) — its distance to the reference's own result
    loss_w0, parity = None, None
    if a.warping == "Iterative":
        L0, fl0 = staged[0]
        l0 = L0()
        g0 = torch.autograd.grad(l0, [f for row in fl0 for f in row], grad_outputs=one)
        loss_w0 = float(l0.item())
        parity = parity_vs_golden(a, rank, loss_w0, g0, P, F)
        del l0, g0

    def barrier():
        watchdog()           # a phase boundary: re-arm
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(a.warmup):
        step(k)
    barrier()
    # One step is five kernel launches behind ~0.3-0.5 ms of Python (autograd + ctypes), enqueued asynchronously: the GPU
    # sets the pace unless the host is slow (0.9-1.1 ms per step seen on some boxes against 0.75 of kernels).  The step is
    # therefore also captured as a hipGraph (same launches, same buffers) and, by default, a short probe decides which of
    # the two launch paths the timed region uses (`--step-graph on|off` forces one); every `event_every`-th step stays
    # eager because its launches carry HIP events.
    def probe_ms(fn, n=24):
        for k in range(4):
            fn(k)
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for k in range(n):
            fn(k)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0_) / n

    # the eager launch path is timed FIRST, in the state a drop-in caller of the loss module sees (after the graph captures
    # below the same probe reads 1.1 ms on a 0.65 ms path: capture leaves the process with stream / allocator state that
    # slows the eager steps that follow — an artefact of measuring both paths in one process, not a property of either)
    eager_probe = round(min(probe_ms(step), probe_ms(step)), 4)      # (best of two: the first seconds of a fresh box are slow)
    graphs, probe = [], None
    if a.step_graph != "off":
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for k in range(len(staged)):
                    step(k)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # One capture stream for every graph, and the rest of this run on that stream: the flow maps are LEAF tensors
            # here, their AccumulateGrad nodes are created while a step is captured — on the capture stream — and the
            # captured outputs keep them alive.  An eager step on any OTHER stream then makes autograd order the two
            # streams for each of the 40 gradients (41 event records per step from the autograd thread, ~10 us of command
            # processor time each: a kernel trace shows a 430 us hole behind every eager step, 1.12 ms instead of 0.66).
            # In a training loop the flows are network outputs, not leaves.
            cap_stream = torch.cuda.Stream()
            for k in range(len(staged)):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=cap_stream):
                    out = step(k)
                graphs.append((gph, out))
            cap_stream.wait_stream(torch.cuda.current_stream())
            torch.cuda.set_stream(cap_stream)
        except Exception as e:                                    # noqa: BLE001
            print(f"[bench] step graph capture failed ({e!r}); running every step eagerly", file=sys.stderr)
            graphs = []
        torch.cuda.synchronize()
    if dist and a.step_graph != "off":
        # a rank whose capture failed must not skip the collectives of the probe below: all ranks agree first, and all
        # drop their graphs if any of them has none
        from taming_event_flow_amd import parallel

        if parallel.any_rank(not graphs):
            graphs = []
    if graphs and a.step_graph == "auto":
        # which launch path does this host sustain?  a short untimed probe of both; the graph is used when it is faster
        probe = {"eager_ms": eager_probe, "graph_ms": round(probe_ms(lambda k: graphs[k % len(graphs)][0].replay()), 4)}
        # (graph replay unless it is clearly slower: the eager steps of THIS process after the captures are not the
        # eager_ms measured above — see there)
        use = probe["graph_ms"] < 1.1 * probe["eager_ms"]
        if dist:                                                  # every rank takes the same path
            flag = torch.tensor([1 if use else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            use = bool(flag.item())
        if not use:
            graphs = []
    if os.environ.get("TEF_BENCH_PROFILE_EAGER") == "1":
        import cProfile
        import pstats

        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        for k in range(100):
            step(k)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(25)
    # groups of `graph_steps` consecutive steps as one graph each (same launches, same buffers, in the same order)
    G = max(1, a.graph_steps)
    groups = []
    if graphs and G > 1:
        try:
            for k in range(len(staged)):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=cap_stream):
                    for j in range(G):
                        out = step(k + j)
                groups.append((gph, out))
            torch.cuda.synchronize()
        except Exception as e:                                    # noqa: BLE001
            print(f"[bench] step-group graph capture failed ({e!r}); one graph per step", file=sys.stderr)
            groups = []
        if probe is not None and groups:
            probe["graph_x%d_ms" % G] = round(probe_ms(lambda k: groups[k % len(groups)][0].replay(), n=8) / G, 4)
    # per-kernel HIP events (start / stop of each launch, on the launch stream) on every `event_every`-th timed step
    lib.tef_profile_enable(0 if a.no_kernel_events else 1)
    if not a.no_kernel_events:
        # one untimed step with kernel events first: the first dispatch that carries timing events switches the queue into
        # profiling mode, ~50 ms of host time once per process (it landed inside a 20-step timed region otherwise)
        for k in range(len(staged)):
            step(k)
        torch.cuda.synchronize()
        lib.tef_profile_collect()
        lib.tef_profile_enable(1)          # (clears what that step recorded)
        lib.tef_profile_pause(1)
    # device time of every step from a HIP event pair on the stream the kernels are launched on (torch's current stream):
    # SURVEY.md section 8d asks for the median over >= 20 steps beside the wall-clock mean that `value` is made of
    # (a short run still profiles at least three steps, so the per-kernel times — and the roofline fractions made of them —
    # are means over several launches, never one sample)
    event_every = max(1, min(a.event_every, a.steps // 3))
    nprof = 0 if a.no_kernel_events else max(1, min(a.steps, a.steps // event_every))

    def is_profiled(k):
        # The profiled steps are the LAST `nprof` steps of the timed region, one after the other.  They are launched eagerly
        # (a graph replay carries no per-launch events), and a kernel trace of a 20-step run shows what that costs: ~70 us of
        # small gaps inside an eager step, nothing when a graph follows a graph — and a 425 us idle gap whenever a graph
        # launch follows eager launches.  Spread over the region (steps 5 / 11 / 17) that was three such gaps, 0.73 ms per
        # step instead of 0.66; at the end of the region no graph follows them.  (Their host time, ~0.8 ms each, hides behind
        # the graph replays queued before them.)
        return k >= a.steps - nprof

    # the timed region as a list of units: (first step, number of steps, how it is launched)
    units, k = [], 0
    while k < a.steps:
        if is_profiled(k) or not graphs:
            units.append((k, 1, "eager"))
            k += 1
        elif groups and k + G <= a.steps and not any(is_profiled(j) for j in range(k, k + G)):
            units.append((k, G, "group"))
            k += G
        else:
            units.append((k, 1, "graph"))
            k += 1
    unit_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in units]
    for e0, e1 in unit_events:          # (HIP creates an event on its first record: tens of ms for the first ones of a process)
        e0.record()
        e1.record()
    barrier()
    # (also: the first call into the caching allocator after the graph captures takes ~40 ms on this stack — whatever it
    # tidies up, it must not land on the first eager step of the timed region, where it turned 0.66 ms steps into 2.6)
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    dbg = [] if os.environ.get("TEF_BENCH_DEBUG") == "1" else None
    for u, (k, n, how) in enumerate(units):
        if dbg is not None:
            dbg.append((how + (" idle" if torch.cuda.current_stream().query() else ""), time.perf_counter()))
        if not a.no_kernel_events:
            lib.tef_profile_pause(0 if how == "eager" and is_profiled(k) else 1)
        if dbg is not None and how == "eager":
            tq = time.perf_counter()
        unit_events[u][0].record()
        if dbg is not None and how == "eager":
            print("[bench] unit %d: pause %.3f ms, record %.3f ms" % (u, 1e3 * (tq - dbg[-1][1]), 1e3 * (time.perf_counter() - tq)), file=sys.stderr)
        if how == "group":
            gph, out = groups[k % len(groups)]
            gph.replay()
            last, last_grads = out
        elif how == "graph":
            gph, out = graphs[k % len(graphs)]
            gph.replay()
            last, last_grads = out
        else:
            if dbg is not None and os.environ.get("TEF_BENCH_DEBUG_SPLIT") == "1":
                tz = time.perf_counter()
                na = torch.cuda.memory_stats()["num_device_alloc"]
                seg0 = {(s_["address"], s_["total_size"]) for s_ in torch.cuda.memory_snapshot()}
                ta = time.perf_counter()
                print("[bench] memory_stats %.3f ms" % (1e3 * (ta - tz)), file=sys.stderr)
                L_, fl_ = staged[k % len(staged)]
                q0 = torch.cuda.current_stream().query()
                loss_ = L_()
                q1 = torch.cuda.current_stream().query()
                tb = time.perf_counter()
                last, last_grads = loss_, torch.autograd.grad(loss_, [f for row in fl_ for f in row])
                tc = time.perf_counter()
                print("[bench] stream idle before forward / after forward / after backward:", q0, q1, torch.cuda.current_stream().query(),
                      file=sys.stderr)
                seg1 = {(s_["address"], s_["total_size"]) for s_ in torch.cuda.memory_snapshot()}
                print("[bench] new segments:", sorted(sz for _, sz in seg1 - seg0), "released:", sorted(sz for _, sz in seg0 - seg1),
                      file=sys.stderr)
                print("[bench] eager unit %d: forward %.2f ms, backward %.2f ms, device allocations %d" % (
                    u, 1e3 * (tb - ta), 1e3 * (tc - tb), torch.cuda.memory_stats()["num_device_alloc"] - na), file=sys.stderr)
            else:
                last, last_grads = step(k)
        unit_events[u][1].record()
    t_enqueue = time.perf_counter() - t0      # host time to enqueue all steps (diagnostic: host- vs device-bound)
    ms1 = torch.cuda.memory_stats()
    alloc_delta = {k_: ms1[k_] - ms0[k_] for k_ in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")
                   if k_ in ms0}
    if dbg is not None:
        dbg.append(("end", time.perf_counter()))
        print("[bench] host ms per unit:", [(h, round(1e3 * (dbg[i + 1][1] - t_), 3)) for i, (h, t_) in enumerate(dbg[:-1])],
              file=sys.stderr)
    barrier()
    elapsed = time.perf_counter() - t0
    if dbg is not None:
        print("[bench] device ms per unit:", [(how, n, round(e0.elapsed_time(e1), 3)) for (k, n, how), (e0, e1) in zip(units, unit_events)],
              file=sys.stderr)
        print("[bench] gaps between units (device ms):", [round(unit_events[i][1].elapsed_time(unit_events[i + 1][0]), 3)
                                                          for i in range(len(units) - 1)], file=sys.stderr)
    step_ms = sorted(ms for (k, n, how), (e0, e1) in zip(units, unit_events) for ms in [e0.elapsed_time(e1) / n] * n)
    step_ms_median = step_ms[len(step_ms) // 2]
    lib.tef_profile_collect()
    kern = {}
    for s in range(lib.tef_profile_slots()):
        n = lib.tef_profile_calls(s)
        if n:
            kern[lib.tef_profile_name(s).decode()] = (lib.tef_profile_ms(s) / n, n)
    lib.tef_profile_enable(0)
    loss_val = float(last.item())
    assert all(torch.isfinite(g_).all().item() for g_ in last_grads[:4])

    rank_ms, single_ms = None, None
    if dist:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [round(1e3 * float(t_.item()) / a.steps, 4) for t_ in every]      # each rank's own clock around the K steps
        elapsed = max(float(t_.item()) for t_ in every)
        # the same steps on rank 0 ALONE (every other rank waits at the barrier): the N = 1 figure of THIS run, on this box,
        # for the weak-scaling efficiency of the batch-sharded path (no data-path collective: anything below 1 is contention
        # for the host, the fabric or the clocks)
        watchdog()
        dist.barrier()
        if rank == 0:
            n1 = max(10, min(a.steps, 100))
            if groups:
                run, per = (lambda k_: groups[k_ % len(groups)][0].replay()), G
            elif graphs:
                run, per = (lambda k_: graphs[k_ % len(graphs)][0].replay()), 1
            else:
                run, per = step, 1
            run(0)
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for k_ in range(n1):
                run(k_)
            torch.cuda.synchronize()
            single_ms = 1e3 * (time.perf_counter() - t0_) / (n1 * per)
        dist.barrier()

    events_per_step = B * P * (a.events + a.detached)
    total_events = events_per_step * a.steps * world
    value = total_events / elapsed
    # more than one rank: the loss step above has no collective (the batch shards), so its 1 -> N curve says nothing about
    # the gradient all-reduce; the DP TRAINING window does, and every rank takes part in measuring it
    dp_extra = None
    if world > 1 and not a.no_train_extra and a.warping == "Iterative":
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.default_stream(dev)):      # (where a training loop runs; not the capture stream above)
            dp_extra = dp_train_extra(a, torch, dist, dev, rank, world)

    if rank == 0:
        delta = a.passes // 2
        alg, splats = algorithmic_bytes(a, delta)
        kernels = {}
        for name, (ms, n) in kern.items():
            e = {"ms": round(ms, 5), "calls_per_step": round(n / a.steps, 3), "samples": n}
            if name in alg and a.warping == "Iterative":
                e["algorithmic_bytes"] = alg[name]
                e["GBps"] = round(alg[name] / (ms * 1e-3) / 1e9, 1)
            kernels[name] = e
        traffic = pmc_traffic(a)
        dominant = max(kern, key=lambda k_: kern[k_][0] * kern[k_][1]) if kern else None
        roofline = None
        if dominant and dominant in alg and a.warping == "Iterative":
            ach = alg[dominant] / (kern[dominant][0] * 1e-3) / 1e9
            roofline = {"kernel": dominant, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get(dominant)}
        out = {
            "metric": "events/sec through IWE+contrast-max loss, 128x128 bs=8",
            "value": round(value, 1), "unit": "events/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # N > 1: every rank's own ms per step (the headline takes the slowest), and rank 0 running the same steps alone
            "ms_per_step_per_rank": rank_ms,
            "ms_per_step_min": min(rank_ms) if rank_ms else None, "ms_per_step_max": max(rank_ms) if rank_ms else None,
            "ms_per_step_rank0_alone": round(single_ms, 4) if single_ms else None,
            "weak_scaling_efficiency_vs_rank0_alone": (round(single_ms / (1e3 * elapsed / a.steps), 4) if single_ms else None),
            "backend": (dist.get_backend() if dist else None), "rccl_world_size": (dist.get_world_size() if dist else 1),
            "config": {"workload": f"{a.warping}/two loss fwd+bwd, {H}x{W}, B={B}/GPU, P={P}, F={F}, "
                                   f"N={a.events}+{a.detached} events/pass/sample, {a.flow} flows sigma=2px "
                                   "(BASELINE.json configs[1])",
                       "global_batch": B * world, "events_per_window_per_gpu": events_per_step,
                       "parallelism": f"dp{world} (batch-sharded, no data-path collective)",
                       "launch": (("hipGraph replay, %d consecutive steps per graph" % G if groups else
                                   "hipGraph replay of the step") +
                                  ("; the last %d steps eager with per-kernel HIP events" % nprof
                                   if not a.no_kernel_events else "")) if graphs else "eager"},
            # window 0 (evaluated once before the timed region; `cpu_baseline.loss` is the same window through the CPU port)
            "loss": round(loss_w0 if loss_w0 is not None else loss_val, 6),
            "loss_last_timed_step": round(loss_val, 6),      # window (steps - 1) % windows
            "parity_vs_golden": parity,
            "ms_per_step_hip_event_median": round(step_ms_median, 4),
            "ms_update_per_window": round(1e3 * t_update, 3),
            "ms_update_per_window_device": round(1e3 * t_update_dev, 3),      # HIP events around the P update() calls
            # update() (AoS -> SoA packing + sort of the P passes) is outside the timed region as SURVEY.md section 8d
            # defines the metric; this is the rate with its wall time added to every step
            "value_including_update": round(events_per_step * world / (elapsed / a.steps + t_update), 1),
            # update() deferred to the evaluation: P recorded passes + one tef_update_window launch per window
            "ms_deferred_update_per_window": round(1e3 * t_update_def, 3),
            "ms_deferred_update_per_window_device": round(1e3 * t_update_def_dev, 3),
            "ms_deferred_update_host": {"record_P_passes": round(1e3 * t_update_def_rec, 3), "flush_call": round(1e3 * t_update_def_flush, 3)},
            "value_including_deferred_update": round(events_per_step * world / (elapsed / a.steps + t_update_def), 1),
            "host_enqueue_ms_per_step": round(1e3 * t_enqueue / a.steps, 4),
            "allocator_events_in_timed_region": alloc_delta,
            "launch_probe": probe,
            # the eager launch path (what a drop-in caller of the loss module runs: Python + ctypes + 7 launches per step)
            "eager_ms": eager_probe,
            "kernel_event_steps": nprof,        # the last steps of the timed region, launched eagerly
            # (round 6) `roofline` is SURVEY.md section 8d's headline: the WHOLE loss step on its compulsory bytes — filled in
            # below; the kernel with the largest share of the step keeps its own line as `roofline_dominant_kernel`
            "roofline": None,
            "roofline_dominant_kernel": roofline,
            "kernels": kernels,
        }
        if "iwe_splat" in kernels and "GBps" in kernels["iwe_splat"]:
            out["roofline_scatter"] = {
                "kernel": "iwe_splat", "bound": "hbm", "achieved": kernels["iwe_splat"]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(kernels["iwe_splat"]["GBps"] / HBM_PEAK_GBS, 4),
                "traffic": traffic.get("iwe_splat"),
                "splats_per_launch": splats,
                "note": "kernel = IWE scatter fused with the image statistics; SURVEY.md section 8d's 16 B per event-splat "
                        "(position 8 + time stamp 4 + flags 4); roofline_scatter_all_in also counts the (A, C + eps) write-out"}
            # the same launch with the write-out of the materialised (A, C + eps) planes counted (8 B x 2 polarities x H x W per
            # image: SURVEY.md section 8d's ~19.6 B per splat "all-in" reading)
            all_in = alg["iwe_splat"] + (P + 1) * F * B * 2 * H * W * 8
            gb = all_in / (kernels["iwe_splat"]["ms"] * 1e-3) / 1e9
            out["roofline_scatter_all_in"] = {
                "kernel": "iwe_splat", "bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gb / HBM_PEAK_GBS, 4), "traffic": traffic.get("iwe_splat"),
                "bytes_per_splat": round(all_in / max(1, splats), 2)}
        # whole loss against the HBM roofline by SURVEY.md section 8d's compulsory traffic: events and masks read once per
        # direction, flow maps read twice and their gradients written once, everything else on chip
        bpe = 48.0 + 24.0 * F * H * W / max(1, a.events + a.detached)
        ach = (events_per_step / (elapsed / a.steps)) * bpe / 1e9          # per GPU
        whole = {"bound": "hbm", "bytes_per_event": round(bpe, 1), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        if traffic and all(k_ in traffic for k_ in kern if k_ in alg):
            moved = sum(traffic[k_] for k_ in kern if k_ in traffic)
            whole["hbm_bytes_per_step_pmc"] = int(moved)
            whole["pmc_over_compulsory"] = round(moved / (bpe * events_per_step), 2)
        out["roofline_whole_loss"] = whole
        out["roofline"] = {"kernel": "whole loss step (K1 warp, K2 scatter + statistics, count, reduce, K6 chain backward, K7 "
                                     "flow-gradient scatter) on SURVEY.md section 8d's compulsory bytes",
                           "bound": "hbm", "achieved": whole["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": whole["frac"], "traffic": whole.get("hbm_bytes_per_step_pmc"),
                           "bytes_per_event": whole["bytes_per_event"]}
        if not a.no_cpu_baseline and a.warping == "Iterative" and world == 1:     # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(a, host_windows[0])
            out["cpu_baseline_1thread"] = cpu_baseline(a, host_windows[0], threads=1, seconds=min(a.cpu_seconds, 6.0), batch=2)
        if world == 1 and not a.no_train_extra and a.warping == "Iterative":
            torch.cuda.synchronize()
            with torch.cuda.stream(torch.cuda.default_stream(dev)):      # (where a training loop runs; not the capture stream above)
                out["extra"] = train_extra(a, torch, dev)
                if "error" not in out["extra"]:
                    out["extra"].update(dropin_extra(a, torch, dev, windows=10))      # the literal train_flow.py loop, no Trainer
                    # (10 windows: the loop ends with ~17 ms of backward still queued, which 3 timed windows showed as + 5.7 ms each)
                    out["extra"].update(dropin_fresh_process(a))
        elif dp_extra is not None:
            out["extra"] = dp_extra
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate


def conv_flops_per_pass(B, H, W, bins=2):
    """2*M*N*K of the 28 convolutions of one RecEVFlowNet forward."""
    fl = 0
    h, w, cin = H, W, bins
    enc = []
    for c in (64, 128, 256, 512):
        h, w = h // 2, w // 2
        fl += 2 * B * h * w * c * cin * 9            # strided head conv
        fl += 3 * 2 * B * h * w * c * (2 * c) * 9    # ConvGRU gates
        enc.append((c, h, w))
        cin = c
    fl += 4 * 2 * B * h * w * 512 * 512 * 9          # 2 residual blocks
    cin_dec = [512, 258, 130, 66]
    for (cin_, cout) in zip(cin_dec, (256, 128, 64, 32)):
        h, w = h * 2, w * 2
        fl += 2 * B * h * w * cout * cin_ * 9        # upsample conv
        fl += 2 * B * h * w * 2 * cout               # 1x1 prediction
    return fl


def bench_train(a, torch, dist, dev, rank, world, lib):
    """Full training window (reference train_flow.py:80-156): P x (encode, RecEVFlowNet forward, loss.update), loss,
    backward through the window (BPTT), DP all-reduce SUM, clip, Adam.  events/s = events of the window / time."""
    import copy

    from taming_event_flow_amd import train

    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"]["batch_size"] = a.batch
    cfg["loader"]["resolution"] = list(a.res)
    cfg["loader"]["max_num_grad_events"] = a.events
    cfg["data"]["passes_loss"] = a.passes
    cfg["loss"]["warping"] = a.warping
    torch.manual_seed(1234)                      # identical initial weights on every rank
    P = a.passes
    from taming_event_flow_amd import parallel

    def build(graph):
        cfg["optimizer"]["capturable"] = bool(graph)
        torch.manual_seed(1234)
        tr_ = train.Trainer(cfg, dev)
        src_ = train.SyntheticSequences(cfg, dev, a.events + a.detached, seq_len=10 ** 9, seed=100 + rank)
        return tr_, src_

    tr, src = build(a.graph)

    def window():
        for _ in range(P):
            tr.step(src.next(), new_seq=False)

    if a.graph:
        # under DP the captured window contains the RCCL all-reduce; the new_seq flag is exchanged on the host.  If the
        # capture fails on ANY rank, every rank falls back to the eager window together.
        tr.reset()
        failed = False
        try:
            captured = tr.capture_window([src.next() for _ in range(P)])
        except Exception as e:                                    # noqa: BLE001
            print(f"[bench] rank {rank}: window graph capture failed ({e!r}); eager window", file=sys.stderr)
            failed = True
        if parallel.any_rank(failed):
            a.graph = False
            torch.cuda.synchronize()
            tr, src = build(False)
        else:
            window = captured

    def barrier():
        watchdog()           # a phase boundary: re-arm
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    if not a.graph:
        tr.reset()
    for _ in range(a.warmup):
        window()
    barrier()
    # (the timed region runs WITHOUT per-kernel events: two hipEventRecords around each of the ~1300 launches of a window
    # cost an eager window 10 ms of host time — the per-kernel times come from one extra, untimed eager window below)
    lib.tef_profile_enable(0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        window()
    t_enqueue = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    kern = {}
    if not a.no_kernel_events:
        lib.tef_profile_enable(1)
        with one_stream(tr):     # (a kernel's events measure its own duration only when nothing runs beside it)
            if a.graph:          # a graph replay carries no per-launch events: one eager window on the same static inputs
                for b in window.inputs:
                    tr.step({k: v.clone() for k, v in b.items()}, new_seq=False)
            else:
                window()
            torch.cuda.synchronize()
        lib.tef_profile_collect()
        for s in range(lib.tef_profile_slots()):
            n = lib.tef_profile_calls(s)
            if n:
                kern[lib.tef_profile_name(s).decode()] = (lib.tef_profile_ms(s) * a.steps, n)      # (scaled: reported per window below)
        lib.tef_profile_enable(0)
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ev_step = a.batch * P * (a.events + a.detached)
    if rank == 0:
        fl_pass = conv_flops_per_pass(a.batch, a.res[0], a.res[1])
        gemm_ms = {k: v[0] / a.steps for k, v in kern.items() if k.startswith("conv_")}
        roofline = None
        if gemm_ms:
            tot_ms = sum(gemm_ms.values())
            flops = 3 * fl_pass * P      # forward + input-gradient + weight-gradient contractions
            ach = flops / (tot_ms * 1e-3) / 1e12
            roofline = {"kernel": "conv gemm_nt (fwd+dgrad+wgrad)", "bound": "mfma", "achieved": round(ach, 2),
                        "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                        "traffic": None, "gemm_ms_per_window": {k: round(v, 3) for k, v in gemm_ms.items()},
                        "note": "kernel durations from one profiled window on ONE stream; the timed windows run the "
                                "Trainer's streams" if tr.dec_stream is not None else "one stream",
                        "window_achieved": round(flops / (1e-3 * 1e3 * elapsed / a.steps) / 1e12, 2)}
        out = {
            "metric": "events/sec through the full training window (RecEVFlowNet + IWE/contrast-max loss), 128x128 bs=8",
            "value": round(ev_step * a.steps * world / elapsed, 1), "unit": "events/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"training window: {P} x (count encoding, RecEVFlowNet fwd, update) + {a.warping}/two "
                                   f"loss + BPTT backward + all-reduce(SUM) + clip + Adam, {a.res[0]}x{a.res[1]}, "
                                   f"B={a.batch}/GPU, N={a.events}+{a.detached} (BASELINE.json configs[2]/[3])",
                       "global_batch": a.batch * world, "parallelism": f"dp{world} (RCCL all-reduce SUM of 125.5 MB grads)",
                       "launch": "hipGraph replay of the whole window" if a.graph else "eager",
                       "streams": 1 + (tr.dec_stream is not None) + (tr.wgrad_stream is not None)
                                  + 2 * (getattr(tr.model.arch.engine, "enc_streams", None) is not None)},
            "loss": round(float(tr.last_loss.item()), 6),
            "host_enqueue_ms_per_step": round(1e3 * t_enqueue / a.steps, 3),
            "conv_gflop_per_pass_fwd": round(fl_pass / 1e9, 2),
            "roofline": roofline,
            "kernels_ms_per_window": {k: round(v[0] / a.steps, 4) for k, v in kern.items()},
        }
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_eval(a, torch, dist, dev, rank, world, lib):
    """DSEC-eval-shape inference (BASELINE.json configs[4], reference eval_flow.py:70-175): replicas only — every rank
    runs its own sequence at batch 1 (eval_flow.py:30 hard-wires it), no collective in the data path.  One step = one
    metric window: P x (device loader stage on raw events, RecEVFlowNet forward, x flow_scaling, validation update:
    forward / backward event warping, forward-propagated and accumulated flow) + FWL + RSAT, then reset."""
    import numpy as np

    from taming_event_flow_amd.dataloader.base import collate_raw_events
    from taming_event_flow_amd.loss import flow_val
    from taming_event_flow_amd.models.model import RecEVFlowNet

    H, W = (480, 640) if a.res == [128, 128] else a.res      # the default resolution of this mode is DSEC's
    N = a.events if a.events != 10000 else 100000             # events per pass (DSEC-like rate), stated in the output
    P = a.passes
    cfg = {"loader": {"resolution": [H, W], "batch_size": 1}, "loss": {"round_ts": False, "flow_scaling": 128},
           "vis": {"mask_output": True}, "metrics": {"name": ["FWL", "RSAT"]}, "data": {"passes_loss": P}}
    torch.manual_seed(1234)
    model = RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2).to(dev)
    model.eval()
    crit = flow_val.Iterative(cfg, dev)
    rng = np.random.default_rng(200 + rank)
    raws = []
    for k in range(P):
        raws.append({"xs": torch.tensor(rng.integers(0, W, N).astype(np.float32), device=dev),
                     "ys": torch.tensor(rng.integers(0, H, N).astype(np.float32), device=dev),
                     "ts": torch.tensor((np.sort(rng.random(N)) + k).astype(np.float32), device=dev),
                     "ps": torch.tensor(rng.integers(0, 2, N).astype(np.float32), device=dev)})

    # two streams (models/engine.py): the decoder half of pass t and the validation update behind it run on a side stream
    # beside the loader stage and the encoders of pass t + 1; TEF_TWO_STREAMS=0: everything on one stream
    side = torch.cuda.Stream(device=dev) if os.environ.get("TEF_TWO_STREAMS", "1") != "0" else None
    eng = model.arch.engine
    model.arch.flow_scale = float(cfg["loss"]["flow_scaling"])      # (eval_flow.py's multiply rides on the last kernel)

    def window():
        model.reset_states()
        crit.reset()
        with torch.no_grad():
            for r in raws:
                b = collate_raw_events(r["xs"], r["ys"], r["ts"], r["ps"], [0, N], (H, W))
                if side is None:
                    crit.update(model(b["net_input"])["flow"], b["event_list"], b["event_list_pol_mask"], b["event_mask"])
                    continue
                eng.side_stream, eng.defer_join = side, True
                try:
                    flows = model(b["net_input"])["flow"]
                finally:
                    eng.defer_join = False
                for k in ("event_list", "event_list_pol_mask", "event_mask"):
                    b[k].record_stream(side)
                with torch.cuda.stream(side):
                    crit.update(flows, b["event_list"], b["event_list_pol_mask"], b["event_mask"])
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            return crit.fwl(), crit.rsat()

    def barrier():
        watchdog()           # a phase boundary: re-arm
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        window()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fwl, rsat = window()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        fl = conv_flops_per_pass(1, H, W)
        out = {
            "metric": "events/sec through DSEC-eval-shape inference (RecEVFlowNet forward + FWL/RSAT validation), 480x640",
            "value": round(N * P * a.steps * world / elapsed, 1), "unit": "events/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"eval window: {P} x (loader stage, RecEVFlowNet fwd, validation update) + FWL + RSAT, "
                                   f"{H}x{W}, B=1/GPU, N={N} events/pass (BASELINE.json configs[4])",
                       "global_batch": world, "parallelism": f"{world} independent replicas (no collective)"},
            "fwl": round(float(fwl.item()), 6), "rsat": round(float(rsat.item()), 6),
            "conv_gflop_per_pass_fwd": round(fl / 1e9, 2),
        }
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def one_stream(tr):
    """Context: the Trainer's passes on ONE stream (per-kernel HIP events only measure a kernel's own duration when nothing
    runs beside it; the timed windows use the Trainer's two streams)."""
    import contextlib

    @contextlib.contextmanager
    def ctx():
        eng = getattr(getattr(tr.model, "arch", None), "engine", None)
        saved = (tr.dec_stream, eng.side_stream if eng is not None else None, eng.wgrad_stream if eng is not None else None,
                 getattr(eng, "enc_streams", None))
        tr.dec_stream = None
        if eng is not None:
            eng.join()
            eng.side_stream = eng.wgrad_stream = eng.enc_streams = None      # (enc_streams: the pipelined encoder levels, round 6)
        try:
            yield
        finally:
            tr.dec_stream = saved[0]
            if eng is not None:
                eng.side_stream, eng.wgrad_stream, eng.enc_streams = saved[1], saved[2], saved[3]

    return ctx()


def train_extra(a, torch, dev):
    """A short measurement of the full training window (the `--mode train --graph` workload, configs[2]) appended to
    the default line so that the driver's own run carries it: ms per window and the conv contractions against the fp32
    MFMA peak.  Never allowed to take the headline down: any failure is reported as a string."""
    import copy

    try:
        from taming_event_flow_amd import _lib, train

        lib = _lib.lib()
        cfg = copy.deepcopy(train.DEFAULT_CONFIG)
        cfg["loader"].update(batch_size=a.batch, resolution=list(a.res), max_num_grad_events=a.events)
        cfg["data"]["passes_loss"] = a.passes
        cfg["optimizer"]["capturable"] = True
        torch.manual_seed(1234)
        tr = train.Trainer(cfg, dev)
        src = train.SyntheticSequences(cfg, dev, a.events + a.detached, seq_len=10 ** 9, seed=100)
        tr.reset()
        window = tr.capture_window([src.next() for _ in range(a.passes)], warmup=1)
        window()
        torch.cuda.synchronize()
        n = 5
        lib.tef_profile_enable(0)
        t0 = time.perf_counter()
        for _ in range(n):
            window()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        # the same window launched eagerly (what a drop-in train_flow.py loop does: one C call per network pass and direction)
        def eager_window():
            for b in window.inputs:
                tr.step({k: v.clone() for k, v in b.items()}, new_seq=False)

        eager_window()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eager_window()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t0) / 3
        # the conv kernels' own time needs per-launch events, which a graph replay does not carry: one more eager window
        lib.tef_profile_enable(1)
        with one_stream(tr):
            eager_window()
            torch.cuda.synchronize()
        lib.tef_profile_collect()
        conv_ms = sum(lib.tef_profile_ms(s_) for s_ in range(lib.tef_profile_slots())
                      if lib.tef_profile_name(s_).decode().startswith("conv_"))
        lib.tef_profile_enable(0)
        flops = 3 * conv_flops_per_pass(a.batch, a.res[0], a.res[1]) * a.passes
        ev = a.batch * a.passes * (a.events + a.detached)
        tr_streams = (1 + (tr.dec_stream is not None) + (tr.wgrad_stream is not None)
                      + 2 * (getattr(tr.model.arch.engine, "enc_streams", None) is not None))
        win_mode = bool(getattr(tr, "window_decode", False))
        del tr, window
        release_now(torch)
        return {"workload": "training window as one hipGraph (bench.py --mode train --graph): RecEVFlowNet fwd + loss + BPTT "
                            "+ clip + Adam, BASELINE configs[2]",
                "train_window_ms": round(ms, 3), "train_events_per_s": round(ev / (ms * 1e-3), 1), "windows_timed": n,
                "train_window_eager_ms": round(eager_ms, 3), "train_window_eager_host_enqueue_ms": round(1e3 * t_host / 3, 3),
                # the kernels' own durations: HIP events around every convolution launch of one window run on ONE stream
                "conv_ms_per_window_eager": round(conv_ms, 3), "conv_tflops": round(flops / (conv_ms * 1e-3) / 1e12, 2),
                "conv_frac_of_fp32_mfma_peak": round(flops / (conv_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                # the same flops over the WHOLE window (two streams, every other launch included)
                "window_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                "window_frac_of_fp32_mfma_peak": round(flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                "streams": tr_streams,
                # (round 6) encoder halves pass by pass — their levels pipelined over two streams —, the decoder halves of the window
                # as one batch (train.Trainer.window_decode); the captured window keeps its weight-gradient reductions on the
                # capture stream, the eager one on a stream of their own
                "window_decode": win_mode}
    except Exception as e:                                    # noqa: BLE001
        return {"error": repr(e)}


def dropin_fresh_process(a):
    """The same loop in a process of its own (`bench.py --mode dropin`), which is where a train_flow.py user runs it.  Inside
    this process — after the staged loss windows, the graph captures and the Trainer's streams — the loop is 4-5 ms per
    window slower than in a fresh one, and on one stream (TEF_LAZY_FLOWS=0) a fresh process is 10-18 ms SLOWER than this one
    (the host side of ~1 600 launches per window depends on what the runtime's queues have seen)."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "dropin", "--steps", "6", "--batch", str(a.batch), "--passes",
           str(a.passes), "--events", str(a.events), "--detached", str(a.detached), "--res", str(a.res[0]), str(a.res[1])]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        e = json.loads(line)["extra"]
        return {"dropin_fresh_process_window_ms": e.get("dropin_window_ms"), "dropin_fresh_process_host_ms": e.get("dropin_host_ms"),
                "dropin_fresh_process_windows_timed": e.get("dropin_windows_timed")}
    except Exception as e:                                    # noqa: BLE001
        return {"dropin_fresh_process_error": repr(e)}


def dropin_extra(a, torch, dev, windows=3):
    """The number a reference user gets by changing three imports: the reference's loop body restated call for call
    (train_flow.py:83-87, :101-137 — model(x), * flow_scaling, loss.update, loss(), .item(), backward, clip_grad_norm_,
    torch.optim.Adam.step, zero_grad() with torch's set_to_none default, detach_states, reset) on this package's
    RecEVFlowNet / Iterative, no train.Trainer.  ms per window (wall) and the host's share of it."""
    import copy

    try:
        from taming_event_flow_amd import train
        from taming_event_flow_amd.loss.flow import Iterative
        from taming_event_flow_amd.models.model import RecEVFlowNet

        cfg = copy.deepcopy(train.DEFAULT_CONFIG)
        cfg["loader"].update(batch_size=a.batch, resolution=list(a.res), max_num_grad_events=a.events)
        cfg["data"]["passes_loss"] = a.passes
        torch.manual_seed(1234)
        model = RecEVFlowNet(cfg["model"].copy(), 2, key="flow").to(dev)
        model.train()
        loss_function = Iterative(cfg, dev)
        optimizer = torch.optim.Adam(model.parameters(), lr=cfg["optimizer"]["lr"])
        optimizer.zero_grad()
        src = train.SyntheticSequences(cfg, dev, a.events + a.detached, seq_len=10 ** 9, seed=100)
        batches = [src.next() for _ in range(a.passes)]
        state = {"loss": 0.0}

        def window():
            for inputs in batches:
                inputs = {k: v.clone() for k, v in inputs.items()}          # (update() shifts the timestamps in place)
                x = model(inputs["net_input"].to(dev))
                for i in range(len(x["flow"])):
                    x["flow"][i] = x["flow"][i] * cfg["loss"]["flow_scaling"]
                loss_function.update(x["flow"], inputs["event_list"].to(dev), inputs["event_list_pol_mask"].to(dev),
                                     inputs["d_event_list"].to(dev), inputs["d_event_list_pol_mask"].to(dev))
                if loss_function.num_passes >= cfg["data"]["passes_loss"]:
                    loss = loss_function()
                    t_i = time.perf_counter()
                    state["loss"] += loss.item()
                    state["item_s"] = state.get("item_s", 0.0) + time.perf_counter() - t_i
                    loss.backward()
                    if cfg["loss"]["clip_grad"] is not None:
                        torch.nn.utils.clip_grad.clip_grad_norm_(model.parameters(), cfg["loss"]["clip_grad"])
                    optimizer.step()
                    optimizer.zero_grad()
                    model.detach_states()
                    loss_function.reset()

        loss_function.reset()
        model.reset_states()
        window()
        window()
        torch.cuda.synchronize()
        import gc

        gc_mode = os.environ.get("TEF_BENCH_GC", "on")
        if gc_mode == "freeze":
            gc.collect()
            gc.freeze()
        elif gc_mode == "off":
            gc.collect()
            gc.disable()
        state["item_s"] = 0.0
        ms0 = torch.cuda.memory_stats(dev)
        t0 = time.perf_counter()
        for _ in range(windows):
            window()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / windows
        if gc_mode == "off":
            gc.enable()
        elif gc_mode == "freeze":
            gc.unfreeze()
        arch = model.arch
        out = {"dropin_workload": "literal train_flow.py loop body (torch.optim.Adam, clip_grad_norm_, zero_grad(set_to_none)), "
                                  "no train.Trainer", "dropin_window_ms": round(ms, 3), "dropin_windows_timed": windows,
               "dropin_in_place_grads": bool(arch._bucket is not None and arch.direct_grads),
               "dropin_deferred_wgrad": bool(arch.deferred_wgrad), "loss_finite": bool(np.isfinite(state["loss"])),
               # flows handed out as LazyFlow tensors (models/lazy.py): decoder half, scaling and update() beside the next encoders
               "dropin_lazy_flows": bool(arch.engine.lazy_flows),
               # host view of a window: waiting in loss.item() (the GPU is behind) / everything else (enqueueing)
               "dropin_host_ms": {"in_loss_item": round(1e3 * state["item_s"] / windows, 3),
                                  "enqueue": round(1e3 * (t_host - state["item_s"]) / windows, 3)},
               # hipMalloc / hipFree calls of torch's caching allocator inside the timed windows (each one synchronises the device)
               "dropin_allocator_events": {k: int(torch.cuda.memory_stats(dev).get(k, 0) - ms0.get(k, 0))
                                           for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")},
               "dropin_streams": [int(st.cuda_stream) if st is not None else None
                                  for st in (arch.engine.side_stream, arch.engine.wgrad_stream, torch.cuda.current_stream())]}
        del model, loss_function, optimizer, src, batches
        release_now(torch)
        return out
    except Exception as e:                                    # noqa: BLE001
        return {"error": repr(e)}


def dp_train_extra(a, torch, dist, dev, rank, world, windows=3):
    """world > 1: the data-parallel TRAINING window (BASELINE configs[3] at this batch size; reference train_flow.py:83-87,
    :120-137 applied to the global batch) — two hipGraphs around the eager gradient all-reduce (train.CapturedWindow) — so
    that the driver's `bench.py --gpus N` line carries the number DP scaling is about.  Called by EVERY rank at the same
    point (it contains collectives); returns the dict on every rank, rank 0 prints it.  A failure on any rank is agreed on
    by all ranks and reported as a string; it never takes the headline down."""
    import copy

    from taming_event_flow_amd import parallel, train

    out = {"workload": "DP training window: RecEVFlowNet fwd + loss + BPTT as one hipGraph, all-reduce(SUM) of the flat "
                       "gradient bucket enqueued eagerly, clip + Adam + state hand-over as a second hipGraph (BASELINE "
                       "configs[3] at B=%d per GPU)" % a.batch,
           "backend": dist.get_backend(), "rccl_world_size": dist.get_world_size()}
    # proof that N ranks contribute to a collective on the data path's backend: sum of (rank + 1) == N (N + 1) / 2
    chk = torch.tensor([float(rank + 1)], device=dev)
    dist.all_reduce(chk, op=dist.ReduceOp.SUM)
    out["rank_checksum"] = float(chk.item())
    out["rank_checksum_expected"] = world * (world + 1) / 2
    failed, err, window, tr = False, None, None, None
    watchdog()                                           # a phase of its own: capture
    try:
        cfg = copy.deepcopy(train.DEFAULT_CONFIG)
        cfg["loader"].update(batch_size=a.batch, resolution=list(a.res), max_num_grad_events=a.events)
        cfg["data"]["passes_loss"] = a.passes
        cfg["optimizer"]["capturable"] = True
        torch.manual_seed(1234)                          # identical initial weights on every rank
        tr = train.Trainer(cfg, dev)
        src = train.SyntheticSequences(cfg, dev, a.events + a.detached, seq_len=10 ** 9, seed=100 + rank)
        tr.reset()
        window = tr.capture_window([src.next() for _ in range(a.passes)], warmup=1)
    except Exception as e:                                    # noqa: BLE001
        failed, err = True, repr(e)
    if parallel.any_rank(failed):                        # (host-side exchange: every rank reaches it)
        out["error"] = err or "window capture failed on another rank"
        return out
    watchdog()                                           # ... and the timed windows
    try:
        window()
        torch.cuda.synchronize()
        dist.barrier()
        window.allreduce_events = []
        t0 = time.perf_counter()
        for _ in range(windows):
            window()
        torch.cuda.synchronize()
        dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        ms = 1e3 * float(el.item()) / windows
        # (start, local work done, stop) per window: start -> stop is the gradient reduction as the main stream sees it
        # (with overlap: the decoder half's all-reduce beside the encoder half's last weight-gradient reduction, then the
        # encoder half's all-reduce); local work done -> stop is what nothing was left to hide
        ar = sorted(e[0].elapsed_time(e[2]) for e in window.allreduce_events)
        ex = sorted(e[1].elapsed_time(e[2]) for e in window.allreduce_events)
        both = torch.tensor([ar[len(ar) // 2], ex[len(ex) // 2]], dtype=torch.float64, device=dev)
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        ar_ms, ex_ms = float(both[0].item()), float(both[1].item())
        nbytes = tr.bucket.flat.numel() * tr.bucket.flat.element_size()
        # replicas must still agree after the updates: every rank applied the same reduced gradient
        p0 = next(iter(tr.model.parameters())).detach().reshape(-1)[:4096].double().sum().reshape(1)
        lo, hi = p0.clone(), p0.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ev = a.batch * a.passes * (a.events + a.detached) * world
        # the per-pass `new_seq` agreement of the eager Trainer loop (train_flow.py:83-87 for the global batch): a blocking
        # all-reduce(MAX) on the host-side group, once per pass
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(200):
            parallel.any_rank(False)
        flag_ms = 1e3 * (time.perf_counter() - t0) / 200
        out.update({
            "new_seq_exchange_ms_per_pass": round(flag_ms, 4),
            # (round 6) ranks of one node exchange the flag over a shared-memory board; over the host-side gloo group otherwise
            "new_seq_exchange_transport": "shared-memory flag board" if parallel._FLAG_BOARD else "gloo all-reduce",
            "dp_train_window_ms": round(ms, 3), "dp_train_events_per_s": round(ev / (ms * 1e-3), 1), "windows_timed": windows,
            "allreduce_ms": round(ar_ms, 3), "allreduce_exposed_ms": round(ex_ms, 3), "allreduce_bytes": nbytes,
            "allreduce_overlap": window.graph_mid is not None,
            # gloo reduces device tensors through the host and blocks it: the "exposed" time then says nothing about overlap
            "allreduce_overlap_measured": dist.get_backend() == "nccl",
            # bus bandwidth of a ring all-reduce: every rank sends and receives 2 (N - 1) / N of the buffer
            "allreduce_GBps": round(nbytes * 2.0 * (world - 1) / world / (ar_ms * 1e-3) / 1e9, 2),
            "replicas_bit_identical": bool(lo.item() == hi.item()),
            "loss_rank0": round(float(tr.last_loss.item()), 6),
        })
    except Exception as e:                                    # noqa: BLE001
        out["error"] = repr(e)
    del tr, window
    release_now(torch)
    return out


def cpu_baseline(a, win, threads=None, seconds=None, batch=None):
    """The CPU restatement (oracle/, "port") on the host cores: one window of the same workload,
    all (head, sample) pairs in parallel over the available cores; bounded to ~10-30 s."""
    from oracle import oracle

    ncores = len(os.sched_getaffinity(0))
    nthr = oracle.threads(threads or ncores)
    bs = batch or a.cpu_batch or a.batch
    sub = {k: ([[m[:bs] for m in row] for row in win["flows"]] if k == "flows" else [x[:bs] for x in win[k]])
           for k in win}
    w = oracle.Window(sub["flows"], sub["ev"], sub["pm"], sub["dev"], sub["dpm"], S=1, mode="two")
    w.iterative(backward=True)                      # untimed: thread pool start-up, page faults
    times = []
    while (sum(times) < (seconds or a.cpu_seconds) or len(times) < 5) and len(times) < 1000:      # at least 5 repetitions
        t0 = time.perf_counter()
        loss, _ = w.iterative(backward=True)
        times.append(time.perf_counter() - t0)
    times.sort()
    ev1 = bs * a.passes * (a.events + a.detached)
    med = times[len(times) // 2]
    return {"value": round(ev1 / med, 1), "unit": "events/s", "cores": nthr, "kind": "port",
            "sample": f"{len(times)} x 1 window (window 0), B={bs}, same P/F/N/resolution as the GPU workload, fwd+bwd, "
                      f"{sum(times):.1f} s; value = events / MEDIAN window time",
            "window_ms": {"min": round(1e3 * times[0], 2), "median": round(1e3 * med, 2), "max": round(1e3 * times[-1], 2)},
            "value_at_min": round(ev1 / times[0], 1),
            "loss": round(float(loss), 6)}


def parity_vs_golden(a, rank, loss, grads, P, F):
    """Distance of this run's window 0 to what the REFERENCE returned for it (tests/golden/bench_window_0.npz: loss, the
    stride-4 lattice of d loss / d flow).  None when the workload is not the recorded one or the fixture is absent."""
    import numpy as np

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "bench_window_0.npz")
    if rank != 0 or not os.path.exists(path):
        return None
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    same = (meta["B"] == a.batch and [meta["H"], meta["W"]] == list(a.res) and meta["P"] == a.passes and meta["F"] == a.heads
            and meta["n_grad"] == a.events and a.detached == 0 and a.flow == "smooth" and a.presort == "none" and meta["seed"] == 0)
    if not same:
        return None
    import torch

    s = meta["stride"]
    lat = torch.stack([torch.stack([grads[t * F + i][..., ::s, ::s] for i in range(F)]) for t in range(P)]).cpu().numpy()
    ref = z["dflows_lattice"]
    return {"fixture": "tests/golden/bench_window_0.npz (reference loss/flow.py Iterative, CPU PyTorch fp32)",
            "loss_reference": float(z["loss"]),
            "loss_rel_err": float(abs(loss - float(z["loss"])) / abs(float(z["loss"]))),
            "dflow_lattice_max_rel_err": float(np.abs(lat.astype(np.float64) - ref).max() / np.abs(ref).max())}

if __name__ == "__main__":
    main()
