"""Event input representations — drop-in for the reference's ``dataloader/encodings.py``
(adapted there from Monash University's events_contrast_maximization).

Same functions and argument meaning: ``events_to_image`` (:8), ``events_to_voxel`` (:32), ``events_to_channels``
(:59) on 1-D per-sample tensors, plus the batched forms ``event_list_to_channels`` / ``event_list_to_voxel`` that
encode a collated ``[B, N, 4] = (ts, y, x, p)`` list in one launch.  The scatter runs in tef_encode.hip
(LDS-resident fp64 accumulation); there is no PyTorch fallback.
"""

import torch

try:
    from .. import _lib
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib

MODE_IMAGE, MODE_CHANNELS, MODE_VOXEL = 0, 1, 2


def _prep(t, name):
    _lib.require_device_tensor(t, name)
    return t.to(torch.float32).contiguous()


def _encode(xs, ys, ts, ps, mode, channels, sensor_size):
    xs, ys, ps = _prep(xs, "xs"), _prep(ys, "ys"), _prep(ps, "ps")
    ts = _prep(ts, "ts") if ts is not None else None
    H, W = int(sensor_size[0]), int(sensor_size[1])
    C = {MODE_IMAGE: 1, MODE_CHANNELS: 2, MODE_VOXEL: channels}[mode]
    out = torch.empty((C, H, W), dtype=torch.float32, device=xs.device)
    rc = _lib.lib().tef_encode_events(xs.data_ptr(), ys.data_ptr(), ts.data_ptr() if ts is not None else None,
                                      ps.data_ptr(), 1, 0, 1, xs.numel(), mode, C, H, W, out.data_ptr(),
                                      _lib.stream_ptr())
    _lib.check(rc, "tef_encode_events")
    return out


def events_to_image(xs, ys, ps, sensor_size=(180, 240), accumulate=True):
    """Accumulate events into an image (reference encodings.py:8-29)."""
    if not accumulate:
        raise NotImplementedError("accumulate=False (last-writer-wins index_put_) is never used by the reference")
    return _encode(xs, ys, None, ps, MODE_IMAGE, 1, sensor_size)[0]


def events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size=(180, 240)):
    """Voxel grid with temporal bilinear interpolation (reference encodings.py:32-56)."""
    assert len(xs) == len(ys) and len(ys) == len(ts) and len(ts) == len(ps)
    return _encode(xs, ys, ts, ps, MODE_VOXEL, int(num_bins), sensor_size)


def events_to_channels(xs, ys, ps, sensor_size=(180, 240)):
    """Two-channel per-polarity event counts (reference encodings.py:59-81)."""
    assert len(xs) == len(ys) and len(ys) == len(ps)
    return _encode(xs, ys, None, ps, MODE_CHANNELS, 2, sensor_size)


def _encode_list(event_list, mode, channels, sensor_size, d_event_list=None):
    ev = _prep(event_list, "event_list")
    B, N, four = ev.shape
    assert four == 4, "event_list must be [B, N, 4] = (ts, y, x, p)"
    dev, Nd = None, 0
    if d_event_list is not None and d_event_list.shape[1] > 0:
        dev = _prep(d_event_list, "d_event_list")
        assert dev.shape[0] == B and dev.shape[2] == 4
        Nd = dev.shape[1]
    H, W = int(sensor_size[0]), int(sensor_size[1])
    C = 2 if mode == MODE_CHANNELS else channels
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=ev.device)
    rc = _lib.lib().tef_encode_event_lists(ev.data_ptr() if N else None, N, dev.data_ptr() if Nd else None, Nd, B,
                                           mode, C, H, W, out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "tef_encode_event_lists")
    return out


def event_list_to_channels(event_list, sensor_size, d_event_list=None):
    """Batched events_to_channels over a zero-padded collated list [B, N, 4] (plus, optionally, the detached list of
    the same batch) -> [B, 2, H, W]."""
    return _encode_list(event_list, MODE_CHANNELS, 2, sensor_size, d_event_list)


def event_list_to_voxel(event_list, num_bins, sensor_size, d_event_list=None):
    """Batched events_to_voxel over a collated list [B, N, 4] (ts in [0, 1]) -> [B, num_bins, H, W]."""
    return _encode_list(event_list, MODE_VOXEL, int(num_bins), sensor_size, d_event_list)
