"""numpy restatement of the reference loader's per-sample event formatting, split and collate.

*** TEST INFRASTRUCTURE ONLY *** (see oracle/oracle.py).  Parity is pinned by tests/golden/loader.npz: outputs of the
reference's own BaseDataLoader methods (event_formatting, augment_events, create_list_encoding, create_polarity_mask,
split_event_list, custom_collate) on seeded raw streams, recorded by tests/golden/make_golden_loader.py.  (The reference
module imports OpenCV at its top, which this image lacks and none of these methods use; the generator executes the class
definition as written, parsed from the reference file, without that import.)  tests/test_loader_oracle.py replays the
fixture bit for bit through this file.  The window / new_seq state machine of dataloader/h5.py (:268-338) needs h5py and
a dataset and stays restated-only.
"""

import numpy as np

from . import oracle

AUG_HORIZONTAL, AUG_VERTICAL, AUG_POLARITY = 1, 2, 4


def event_formatting(xs, ys, ts, ps):
    """dataloader/base.py:153-177: fp32 casts, ps*2-1, ts normalised to [0, 1] over the window."""
    xs, ys, ts = (np.asarray(a).astype(np.float32) for a in (xs, ys, ts))
    ps = np.asarray(ps).astype(np.float32) * np.float32(2) - np.float32(1)
    if ts.shape[0] > 0:
        ts = (ts - ts[0]) / (ts[-1] - ts[0])
    return xs, ys, ts, ps


def augment_events(xs, ys, ps, flags, res):
    """dataloader/base.py:192-222."""
    if flags & AUG_HORIZONTAL:
        xs = np.float32(res[1] - 1) - xs
    if flags & AUG_VERTICAL:
        ys = np.float32(res[0] - 1) - ys
    if flags & AUG_POLARITY:
        ps = ps * np.float32(-1)
    return xs, ys, ps


def create_list_encoding(xs, ys, ts, ps):
    """dataloader/base.py:252-263 -> [4 x N] (ts, y, x, p)."""
    return np.stack([ts, ys, xs, ps])


def create_polarity_mask(ps):
    """dataloader/base.py:265-278 -> [2 x N]."""
    m = np.stack([ps, ps]).astype(np.float32)
    m[0, :][m[0, :] < 0] = 0
    m[0, :][m[0, :] > 0] = 1
    m[1, :][m[1, :] < 0] = -1
    m[1, :][m[1, :] > 0] = 0
    m[1, :] *= -1
    return m


def split_event_list(event_list, mask, max_num_grad_events, sampled_indices):
    """dataloader/base.py:348-377 with the multinomial draw given."""
    d_list = np.zeros((4, 0), np.float32)
    d_mask = np.zeros((2, 0), np.float32)
    if max_num_grad_events is not None and event_list.shape[1] > max_num_grad_events:
        unsampled = np.ones(event_list.shape[1], bool)
        unsampled[sampled_indices] = False
        d_list, d_mask = event_list[:, unsampled], mask[:, unsampled]
        event_list, mask = event_list[:, sampled_indices], mask[:, sampled_indices]
    return event_list, mask, d_list, d_mask


def custom_collate(batch):
    """dataloader/base.py:392-434."""
    out = {}
    for key in batch[0]:
        items = [entry[key] for entry in batch]
        if key in ("event_list", "event_list_pol_mask", "d_event_list", "d_event_list_pol_mask"):
            N = max(it.shape[1] for it in items)
            items = [np.concatenate((it, np.zeros((it.shape[0], N - it.shape[1]), np.float32)), axis=1) for it in items]
        item = np.stack(items)
        if item.ndim == 3:
            item = item.transpose(0, 2, 1)
        out[key] = np.ascontiguousarray(item)
    return out


def get_item(xs, ys, ts, ps, res, max_num_grad_events, flags, sampled_indices, voxel):
    """The event part of dataloader/h5.py::__getitem__ (:340-431) for one sample."""
    if np.asarray(xs).shape[0] <= 10:                                  # :340-345
        xs = ys = ts = ps = np.empty([0])
    xs, ys, ts, ps = event_formatting(xs, ys, ts, ps)                 # :348
    xs, ys, ps = augment_events(xs, ys, ps, flags, res)               # :356
    event_list = create_list_encoding(xs, ys, ts, ps)                 # :362
    mask = create_polarity_mask(ps)                                   # :363
    cnt = oracle.events_to_channels(xs, ys, ps, res[0], res[1])       # :366 (base.py:280-297)
    ev_mask = (cnt.sum(axis=0, keepdims=True) > 0).astype(np.float32)  # :367 (base.py:299-311)
    if voxel is None:                                                 # :374-377
        net_input = cnt.copy()
    else:
        net_input = oracle.events_to_voxel(xs, ys, ts, ps, voxel, res[0], res[1])
    event_list, mask, d_list, d_mask = split_event_list(event_list, mask, max_num_grad_events, sampled_indices)   # :413
    return {"net_input": net_input, "event_cnt": cnt, "event_mask": ev_mask, "event_list": event_list,
            "event_list_pol_mask": mask, "d_event_list": d_list, "d_event_list_pol_mask": d_mask}


def collate_raw_events(xs, ys, ts, ps, offsets, res, max_num_grad_events=None, flags=None, sampled=None, voxel=None):
    """B x get_item + custom_collate; `sampled` int [B, G] (rows of unsplit samples ignored)."""
    B = len(offsets) - 1
    batch = []
    for b in range(B):
        sl = slice(offsets[b], offsets[b + 1])
        batch.append(get_item(xs[sl], ys[sl], ts[sl], ps[sl], res, max_num_grad_events, flags[b] if flags is not None else 0,
                              None if sampled is None else np.asarray(sampled[b]), voxel))
    return custom_collate(batch)
