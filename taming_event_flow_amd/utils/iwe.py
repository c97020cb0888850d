"""Drop-in for the reference's ``utils/iwe.py``.

``compute_pol_iwe`` / ``deblur_events`` (:139-257; eval_flow.py:104-111, visualisation) are one fused launch.  The
training-time primitives (:5-136) live fused inside the loss kernels (tef_loss.hip); for callers that import them one
by one they are also available here as stand-alone, FORWARD-ONLY functions on the HIP kernels of tef_val.hip — the
differentiable path is the loss module (``loss.flow``), these carry no autograd graph.  In the reference they are
ordinary differentiable torch functions: a call that WOULD need a gradient here (grad mode on and an input that
requires grad) raises instead of silently returning a constant."""

import torch

try:
    from .. import _lib
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib


def _forward_only(fn, **tensors):
    """The HIP primitives below carry no autograd graph: refuse a call whose result the caller could differentiate."""
    if torch.is_grad_enabled():
        bad = [k for k, t in tensors.items() if isinstance(t, torch.Tensor) and t.requires_grad]
        if bad:
            raise RuntimeError(
                f"utils.iwe.{fn} is forward-only on the MI355X path, but {', '.join(bad)} require(s) grad: gradients of the "
                "warping / IWE primitives are computed inside loss.flow.Linear / Iterative (the fused HIP loss); build the "
                "loss with those modules, or call this under torch.no_grad() / on detached tensors if a constant is meant")


def compute_pol_iwe(flow, event_list, res, pol_mask, round_idx=True, round_flow=True):
    """Per-polarity IWE [B, 2, H, W] of `event_list` ([B, N, 4], ts in [0, 1]) warped to t = 1 with `flow` [B, 2, H, W]."""
    _forward_only("compute_pol_iwe", flow=flow, event_list=event_list, pol_mask=pol_mask)
    for name, t in (("flow", flow), ("event_list", event_list), ("pol_mask", pol_mask)):
        _lib.require_device_tensor(t, name)
    flow, ev, pm = (t.to(torch.float32).contiguous() for t in (flow, event_list, pol_mask))
    B, N = ev.shape[0], ev.shape[1]
    H, W = int(res[0]), int(res[1])
    out = torch.empty((B, 2, H, W), dtype=torch.float32, device=flow.device)
    rc = _lib.lib().tef_pol_iwe(flow.data_ptr(), ev.data_ptr(), pm.data_ptr(), B, N, H, W, 1 if round_idx else 0,
                                1 if round_flow else 0, out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "tef_pol_iwe")
    return out


def deblur_events(flow, event_list, res, round_idx=True, polarity_mask=None, round_flow=True):
    """Single-polarity form (reference :139-224): polarity_mask [B, N, 1] (None = every event) -> [B, 1, H, W]."""
    B, N = event_list.shape[0], event_list.shape[1]
    if polarity_mask is None:
        polarity_mask = torch.ones((B, N, 1), dtype=torch.float32, device=event_list.device)
    pm = torch.cat([polarity_mask, torch.zeros_like(polarity_mask)], dim=2)
    return compute_pol_iwe(flow, event_list, res, pm, round_idx, round_flow)[:, 0:1]


# ---- the training-time primitives, stand-alone and forward-only (reference :5-136) ---------------------------------------
def event_propagation(events_ts, events_idx, flow, tref):
    """Linear warp of event locations to `tref` with their per-event flow (reference :5-14)."""
    return events_idx + (tref - events_ts) * flow


def get_event_flow(flow_map_x, flow_map_y, event_loc):
    """Bilinear lookup (align_corners, zero padding) of [B, H, W] flow maps at [B, N, 2] (y, x) -> [B, N, 2] (f_y, f_x)
    (reference :17-40)."""
    _forward_only("get_event_flow", flow_map_x=flow_map_x, flow_map_y=flow_map_y, event_loc=event_loc)
    for name, t in (("flow_map_x", flow_map_x), ("flow_map_y", flow_map_y), ("event_loc", event_loc)):
        _lib.require_device_tensor(t, name)
    fx, fy = flow_map_x.detach().to(torch.float32).contiguous(), flow_map_y.detach().to(torch.float32).contiguous()
    loc = event_loc.detach().to(torch.float32).contiguous()
    B, H, W = fx.shape
    N = loc.shape[1]
    out = torch.empty((B, N, 2), dtype=torch.float32, device=loc.device)
    lib = _lib.lib()
    for b in range(B):            # the validation kernel is per sample (flow_val.py runs at batch 1)
        if N:
            rc = lib.tef_val_event_step(fx[b].data_ptr(), fy[b].data_ptr(), H, W, loc[b].data_ptr(), None, None, N, 0.0, 0,
                                        out[b].data_ptr(), _lib.stream_ptr())
            _lib.check(rc, "tef_val_event_step")
    return out


def purge_unfeasible(event_loc, event_pol_mask, res):
    """Zero the locations and polarity masks of events outside [0, H-1] x [0, W-1] (reference :43-60)."""
    y, x = event_loc[:, :, 0:1], event_loc[:, :, 1:2]
    inside = ((y >= 0) & (y <= res[0] - 1.0) & (x >= 0) & (x <= res[1] - 1.0)).to(event_loc.dtype)
    return event_loc * inside, event_pol_mask * inside


def get_interpolation(warped_events, res, round_idx=False, zeros=None):
    """Scatter indices and bilinear (or nearest-pixel) weights of [B, N, 2] locations (reference :63-113):
    -> idx, weights [B, 4N, 1] (corner blocks TL, TR, BL, BR) or [B, N, 1] with round_idx."""
    _forward_only("get_interpolation", warped_events=warped_events)
    _lib.require_device_tensor(warped_events, "warped_events")
    loc = warped_events.detach().to(torch.float32).contiguous()
    B, N = loc.shape[0], loc.shape[1]
    n_out = N if round_idx else 4 * N
    idx = torch.empty((B, n_out, 1), dtype=torch.float32, device=loc.device)
    wgt = torch.empty((B, n_out, 1), dtype=torch.float32, device=loc.device)
    rc = _lib.lib().tef_interp_corners(loc.data_ptr(), B, N, int(res[0]), int(res[1]), 1 if round_idx else 0, idx.data_ptr(),
                                       wgt.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "tef_interp_corners")
    return idx, wgt


def interpolate(idx, weights, res, polarity_mask=None, zeros=None):
    """Image [B, 1, H, W] of the weights scattered to their indices (reference :116-136)."""
    _forward_only("interpolate", idx=idx, weights=weights, polarity_mask=polarity_mask, zeros=zeros)
    _lib.require_device_tensor(idx, "idx")
    B, n = idx.shape[0], idx.shape[1]
    HW = int(res[0]) * int(res[1])
    ix, w = idx.detach().to(torch.float32).contiguous(), weights.detach().to(torch.float32).contiguous()
    pm = polarity_mask.detach().to(torch.float32).contiguous() if polarity_mask is not None else None
    out = torch.empty((B, HW), dtype=torch.float32, device=idx.device)
    rc = _lib.lib().tef_scatter_add(ix.data_ptr(), w.data_ptr(), pm.data_ptr() if pm is not None else None, B, n, HW,
                                    out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "tef_scatter_add")
    out = out.view(B, 1, int(res[0]), int(res[1]))
    if zeros is not None:
        out = out + zeros.view_as(out)
    return out
