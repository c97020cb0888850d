"""One-shot image of warped events — drop-in for ``compute_pol_iwe`` / ``deblur_events`` of the reference's
``utils/iwe.py`` (:139-257; used by eval_flow.py:104-111 for visualisation).  The training-time primitives of that file
(event_propagation, get_event_flow, purge_unfeasible, get_interpolation, interpolate, :5-136) are fused inside the HIP
loss kernels (tef_loss.hip) and are not exposed one by one."""

import torch

try:
    from .. import _lib
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib


def compute_pol_iwe(flow, event_list, res, pol_mask, round_idx=True, round_flow=True):
    """Per-polarity IWE [B, 2, H, W] of `event_list` ([B, N, 4], ts in [0, 1]) warped to t = 1 with `flow` [B, 2, H, W]."""
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
