"""Loader-side event formatting — counterpart of the reference's ``dataloader/base.py`` + the per-sample part of
``dataloader/h5.py::__getitem__`` (:340-431), without HDF5 / OpenCV.

Two layers:

* ``BaseDataLoader`` keeps the reference's static helpers with their names, argument meaning and ``[C x N]`` layouts
  (``create_list_encoding`` :252-263, ``create_polarity_mask`` :265-278, ``split_event_list`` :348-377,
  ``custom_collate`` :392-434), written with plain tensor indexing — host plumbing, any device;
* ``collate_raw_events`` does the same work for a whole batch of ragged raw event streams on the MI355X in
  ``tef_collate_events`` + ``tef_encode_event_lists`` (tef_collate.hip, tef_encode.hip) and returns the batch dict
  the training loop consumes (``train_flow.py:101-117``): no per-sample host loop, no padding copies.
"""

import ctypes

import torch

try:
    from .. import _lib
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib
from . import encodings

AUG_BITS = {"Horizontal": 1, "Vertical": 2, "Polarity": 4}      # include/tef.h TEF_AUG_*


class BaseDataLoader:
    """Static helpers of the reference's BaseDataLoader (the HDF5 machinery around them is out of scope)."""

    @staticmethod
    def create_list_encoding(xs, ys, ts, ps):
        """[4 x N] list (ts, y, x, p); base.py:252-263."""
        return torch.stack([ts, ys, xs, ps])

    @staticmethod
    def create_polarity_mask(ps):
        """[2 x N] mask: row 0 = positive events, row 1 = negative events; base.py:265-278."""
        return torch.stack([(ps > 0).to(ps.dtype), (ps < 0).to(ps.dtype)])

    @staticmethod
    def split_event_list(event_list, event_list_pol_mask, max_num_grad_events, sampled_indices=None):
        """Gradient list of at most `max_num_grad_events` randomly chosen events + the rest as detached list;
        base.py:348-377.  `sampled_indices` (optional) replaces the multinomial draw for reproducibility."""
        d_event_list = torch.zeros((4, 0), device=event_list.device)
        d_event_list_pol_mask = torch.zeros((2, 0), device=event_list.device)
        n = event_list.shape[1]
        if max_num_grad_events is not None and n > max_num_grad_events:
            if sampled_indices is None:
                probs = torch.ones(n, dtype=torch.float32) / n
                sampled_indices = probs.multinomial(max_num_grad_events, replacement=False)
            sampled_indices = sampled_indices.to(event_list.device)
            unsampled = torch.ones(n, dtype=torch.bool, device=event_list.device)
            unsampled[sampled_indices] = False
            d_event_list = event_list[:, unsampled]
            d_event_list_pol_mask = event_list_pol_mask[:, unsampled]
            event_list = event_list[:, sampled_indices]
            event_list_pol_mask = event_list_pol_mask[:, sampled_indices]
        return event_list, event_list_pol_mask, d_event_list, d_event_list_pol_mask

    @staticmethod
    def custom_collate(batch):
        """List of per-sample dicts -> dict of batched tensors; the four event lists are zero-padded to the longest of
        the batch and transposed to [B, N, C]; base.py:392-434."""
        batch_dict = {key: [entry[key] for entry in batch] for key in batch[0].keys()}
        for key, items in batch_dict.items():
            if items[0] is None:
                batch_dict[key] = None
                continue
            if key in ["event_list", "event_list_pol_mask", "d_event_list", "d_event_list_pol_mask"]:
                N = max(it.shape[1] for it in items)
                items = [torch.cat((it, torch.zeros((it.shape[0], N - it.shape[1]), device=it.device)), dim=1)
                         for it in items]
            item = torch.stack(items)
            if len(item.shape) == 3:
                item = item.transpose(2, 1)
            batch_dict[key] = item
        return batch_dict


def draw_sampled_indices(counts, max_num_grad_events, generator=None):
    """The loader's random choice (base.py:363-366) for every sample that needs a split: int32 [B, G] on the host,
    rows of samples that are not split are left at -1."""
    B = len(counts)
    G = int(max_num_grad_events or 0)
    out = torch.full((B, max(G, 1)), -1, dtype=torch.int32)
    for b, n in enumerate(counts):
        if G and n > G:
            probs = torch.ones(n, dtype=torch.float32) / n
            out[b, :G] = probs.multinomial(G, replacement=False, generator=generator).to(torch.int32)
    return out


def collate_raw_events(xs, ys, ts, ps, offsets, resolution, max_num_grad_events=None, augmentation=None,
                       sampled_indices=None, voxel=None):
    """Batch dict from the raw events of B samples, all on the device.

    xs, ys, ts, ps: 1-D fp32 device tensors, the samples' event streams back to back (ps in {0, 1}, raw timestamps);
    offsets: B+1 host ints; augmentation: per-sample lists of active mechanisms ("Horizontal", "Vertical", "Polarity")
    or TEF_AUG_* bit masks; sampled_indices: int32 [B, max_num_grad_events] (default: drawn like the reference does).
    Returns net_input, event_cnt, event_mask, event_list [B,N,4], event_list_pol_mask [B,N,2], d_event_list,
    d_event_list_pol_mask (h5.py:413-431 after custom_collate)."""
    for name, t in (("xs", xs), ("ys", ys), ("ts", ts), ("ps", ps)):
        _lib.require_device_tensor(t, name)
    xs, ys, ts, ps = (t.to(torch.float32).contiguous() for t in (xs, ys, ts, ps))
    offsets = [int(o) for o in offsets]
    B = len(offsets) - 1
    H, W = int(resolution[0]), int(resolution[1])
    G = int(max_num_grad_events or 0)
    lib = _lib.lib()
    offs = (ctypes.c_int * (B + 1))(*offsets)
    n_g, n_d = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.tef_collate_counts(offs, B, G, ctypes.byref(n_g), ctypes.byref(n_d)), "tef_collate_counts")
    N, Nd = n_g.value, n_d.value
    flags = None
    if augmentation is not None:
        bits = [a if isinstance(a, int) else sum(AUG_BITS[m] for m in a) for a in augmentation]
        flags = (ctypes.c_int * B)(*bits)
    sampled = None
    if Nd > 0:
        if sampled_indices is None:
            counts = [offsets[b + 1] - offsets[b] for b in range(B)]
            sampled_indices = draw_sampled_indices(counts, G)
        sampled = sampled_indices.to(device=xs.device, dtype=torch.int32).contiguous()
        assert tuple(sampled.shape) == (B, G), "sampled_indices must be [B, max_num_grad_events]"
    dev = xs.device
    ev = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    pm = torch.empty((B, N, 2), dtype=torch.float32, device=dev)
    d_ev = torch.empty((B, Nd, 4), dtype=torch.float32, device=dev)
    d_pm = torch.empty((B, Nd, 2), dtype=torch.float32, device=dev)
    nbytes = lib.tef_collate_workspace_bytes(offs, B)
    ws = torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=dev)
    rc = lib.tef_collate_events(xs.data_ptr(), ys.data_ptr(), ts.data_ptr(), ps.data_ptr(), offs,
                                sampled.data_ptr() if sampled is not None else None, flags, B, G, N, Nd, H, W,
                                ws.data_ptr(), nbytes, ev.data_ptr() if N else None, pm.data_ptr() if N else None,
                                d_ev.data_ptr() if Nd else None, d_pm.data_ptr() if Nd else None, _lib.stream_ptr())
    _lib.check(rc, "tef_collate_events")
    cnt = encodings.event_list_to_channels(ev, (H, W), d_ev)                     # base.py:280-297
    mask = (cnt.sum(dim=1, keepdim=True) > 0.0).to(torch.float32)               # base.py:299-311
    if voxel is None:                                                           # h5.py:387-391
        net_input = cnt.clone()
    else:
        net_input = encodings.event_list_to_voxel(ev, voxel, (H, W), d_ev)
    return {"net_input": net_input, "event_cnt": cnt, "event_mask": mask, "event_list": ev,
            "event_list_pol_mask": pm, "d_event_list": d_ev, "d_event_list_pol_mask": d_pm}
