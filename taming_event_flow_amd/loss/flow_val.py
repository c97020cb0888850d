"""Validation metrics — drop-in for the reference's ``loss/flow_val.py`` (FWL, RSAT, AEE, windowed images).

Same classes and methods: ``BaseValidation`` (:12), ``Linear`` (:317), ``Iterative`` (:419) with
``update(flow_list, event_list, pol_mask, event_mask)``, ``reset()``, ``num_passes``, ``rsat()``, ``fwl()``,
``compute_aee(pred, gt, mask=None)``, ``window_events()``, ``window_flow()``, ``window_iwe()``.  Like the reference
(:439, :597-599) the module is batch-1 only.  Arithmetic runs in tef_val.hip; the growing lists the reference rebuilds
with ``torch.cat`` every pass live in buffers with spare capacity (``_Rows``), same values, each row copied once.
"""

import torch

try:
    from .. import _lib
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib


def _f32(t, name):
    _lib.require_device_tensor(t, name)
    return t.to(torch.float32).contiguous()


class _Rows:
    """Append-only rows of one of the module's growing lists.  ``append(new)`` returns what the reference's
    ``torch.cat([old, new], dim=0)`` holds, as a view of a buffer whose capacity doubles when it runs out: the rows
    already stored are not copied again every pass (the reference's O(P^2) re-concatenation), and in-place kernels
    keep working on the stored rows."""

    def __init__(self):
        self.buf, self.n = None, 0

    def append(self, new):
        m = new.shape[0]
        if self.buf is None or self.n + m > self.buf.shape[0] or self.buf.shape[1:] != new.shape[1:]:
            cap = max(2 * (self.n + m), 16)
            buf = torch.empty((cap,) + tuple(new.shape[1:]), dtype=new.dtype, device=new.device)
            if self.n:
                buf[:self.n].copy_(self.buf[:self.n])
            self.buf = buf
        self.buf[self.n:self.n + m].copy_(new)
        self.n += m
        return self.buf[:self.n]


class BaseValidation(torch.nn.Module):
    """Base class for validation metrics (reference loss/flow_val.py:12-314)."""

    def __init__(self, config, device):
        super().__init__()
        self.res = config["loader"]["resolution"]
        self.device = torch.device(device)
        self.config = config
        self._passes = 0
        self._event_ts = None          # [N]
        self._event_loc = None         # [N, 2] (y, x)
        self._event_pol_mask = None    # [N, 2]
        self._flow_maps_x = None       # [P, H, W]
        self._flow_maps_y = None
        self._event_mask = None        # [P, H, W]
        self._rows = {}                # attribute name -> _Rows

    # ---- helpers -------------------------------------------------------------------------------------------------
    @property
    def num_passes(self):
        return self._passes

    def _hw(self):
        return int(self.res[0]), int(self.res[1])

    def _event_step(self, fx, fy, loc, ts, mask, tref, do_warp=True, want_flow=False):
        """In-place warping step on (loc, ts, mask); returns the sampled flow [N, 2] = (f_y, f_x) if wanted."""
        H, W = self._hw()
        N = loc.shape[0]
        flow = torch.empty((N, 2), dtype=torch.float32, device=loc.device) if want_flow else None
        rc = _lib.lib().tef_val_event_step(fx.data_ptr(), fy.data_ptr(), H, W, loc.data_ptr(),
                                           ts.data_ptr() if ts is not None else None,
                                           mask.data_ptr() if mask is not None else None, N, float(tref),
                                           1 if do_warp else 0, flow.data_ptr() if flow is not None else None,
                                           _lib.stream_ptr())
        _lib.check(rc, "tef_val_event_step")
        return flow

    def _event_image(self, loc, mask, ts=None, round_idx=True):
        """-> (cnt [2,H,W], tsum [2,H,W] or None)"""
        H, W = self._hw()
        cnt = torch.empty((2, H, W), dtype=torch.float32, device=loc.device)
        tsum = torch.empty((2, H, W), dtype=torch.float32, device=loc.device) if ts is not None else None
        rc = _lib.lib().tef_val_event_image(loc.data_ptr(), mask.data_ptr(), ts.data_ptr() if ts is not None else None,
                                            loc.shape[0], H, W, 1 if round_idx else 0, cnt.data_ptr(),
                                            tsum.data_ptr() if tsum is not None else None, _lib.stream_ptr())
        _lib.check(rc, "tef_val_event_image")
        return cnt, tsum

    def forward_prop_flow(self, i, tref, flow_maps_x, flow_maps_y, inplace=False):
        """Forward propagation of flow map i to time tref with bilinear splatting (reference :43-74).
        flow_maps_*: [P, H, W]; returns ([1,1,H,W], [1,1,H,W]) = (x, y) like the reference.  inplace: the result
        replaces map i (the splat reads the map, a second launch writes it: no hazard)."""
        H, W = self._hw()
        dev = flow_maps_x.device
        scratch = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        ox = flow_maps_x[i] if inplace else torch.empty((H, W), dtype=torch.float32, device=dev)
        oy = flow_maps_y[i] if inplace else torch.empty((H, W), dtype=torch.float32, device=dev)
        rc = _lib.lib().tef_val_forward_prop_flow(flow_maps_x[i].data_ptr(), flow_maps_y[i].data_ptr(), H, W,
                                                  float(tref - i), scratch.data_ptr(), ox.data_ptr(), oy.data_ptr(),
                                                  _lib.stream_ptr())
        _lib.check(rc, "tef_val_forward_prop_flow")
        return ox.view(1, 1, H, W), oy.view(1, 1, H, W)

    # ---- reference API -------------------------------------------------------------------------------------------
    def update_base(self, flow_list, event_list, pol_mask, event_mask):
        """reference :76-116.  Shifts event_list[:, :, 0] in place by the pass index (:88)."""
        for name, t in (("event_list", event_list), ("pol_mask", pol_mask), ("event_mask", event_mask)):
            _lib.require_device_tensor(t, name)
        if event_list.shape[0] != 1:
            raise NotImplementedError("validation metrics are batch-1 only, like the reference (flow_val.py:439,597)")
        H, W = self._hw()
        event_list[:, :, 0:1] += self._passes
        ev = event_list[0].to(torch.float32)
        ts = ev[:, 0].clone().contiguous()
        if self.config["loss"]["round_ts"]:
            ts[...] = ts.min() + 0.5
        loc = ev[:, 1:3].contiguous()
        pm = _f32(pol_mask[0], "pol_mask")
        self._append("_event_ts", ts)
        self._append("_event_loc", loc)
        self._append("_event_pol_mask", pm)
        flow = _f32(flow_list[-1], "flow map")            # only the highest-resolution flow (:103)
        self._append("_flow_maps_x", flow[0, 0:1])
        self._append("_flow_maps_y", flow[0, 1:2])
        self._append("_event_mask", _f32(event_mask, "event_mask").reshape(-1, H, W))
        return ts, loc, pm

    def _append(self, name, new):
        """self.<name> = cat([self.<name>, new]) (the rows of `new` are copied, never aliased)"""
        rows = self._rows.get(name)
        if rows is None or getattr(self, name) is None:
            rows = self._rows[name] = _Rows()
        setattr(self, name, rows.append(new))

    def reset_base(self):
        self._passes = 0
        self._rows = {}
        self._event_ts = self._event_loc = self._event_pol_mask = None
        self._flow_maps_x = self._flow_maps_y = self._event_mask = None

    def window_events_base(self, round_idx=False):
        """reference :129-143 -> [1, 2, H, W]"""
        cnt, _ = self._event_image(self._event_loc, self._event_pol_mask, None, round_idx)
        return cnt.unsqueeze(0)

    def window_flow_base(self, flow_maps_x, flow_maps_y, mask=False, divisor=None):
        """reference :145-172 -> [1, 2, H, W] (x, y); flow_maps_* [P, H, W]"""
        H, W = self._hw()
        fx, fy = flow_maps_x.contiguous(), flow_maps_y.contiguous()
        out = torch.empty((2, H, W), dtype=torch.float32, device=fx.device)
        em = self._event_mask.contiguous() if mask else None
        rc = _lib.lib().tef_val_average_flow(fx.data_ptr(), fy.data_ptr(), fx.shape[0], H, W,
                                             divisor.data_ptr() if divisor is not None else None,
                                             em.data_ptr() if em is not None else None,
                                             em.shape[0] if em is not None else 0, out.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tef_val_average_flow")
        return out.unsqueeze(0)

    def _metrics(self, fw_loc, zero_loc, fw_mask, zero_mask, ts):
        """-> tensor (fwl, rsat), reference compute_fwl :189-212 / compute_rsat :214-274"""
        H, W = self._hw()
        cf, tf = self._event_image(fw_loc, fw_mask, ts, True)
        cz, tz = self._event_image(zero_loc, zero_mask, ts, True)
        out = torch.empty((2,), dtype=torch.float32, device=fw_loc.device)
        nbytes = _lib.lib().tef_val_metrics_scratch_bytes(H, W)
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=fw_loc.device)
        rc = _lib.lib().tef_val_metrics(cf.data_ptr(), tf.data_ptr(), cz.data_ptr(), tz.data_ptr(), H, W,
                                        float(self._passes), out.data_ptr(), scratch.data_ptr(), nbytes, _lib.stream_ptr())
        _lib.check(rc, "tef_val_metrics")
        return out

    def compute_fwl(self, fw_events, zero_events, fw_pol_mask, zero_pol_mask):
        """Flow Warp Loss: variance ratio of the warped / un-warped event images (Stoffregen et al., ECCV 2020)."""
        ts = torch.zeros((fw_events.shape[0],), dtype=torch.float32, device=fw_events.device)
        return self._metrics(fw_events, zero_events, fw_pol_mask, zero_pol_mask, ts)[0]

    def compute_rsat(self, fw_events, zero_events, fw_pol_mask, zero_pol_mask, ts_list):
        """Ratio of the Squared Averaged Timestamps (Hagenaars, Paredes-Valles et al., NeurIPS 2021)."""
        return self._metrics(fw_events, zero_events, fw_pol_mask, zero_pol_mask, ts_list)[1:2]

    def compute_aee(self, pred, gt, mask=None):
        """Average endpoint error (reference :276-314); mask = event mask [1, P, H, W] (pixels without events are out)."""
        pred, gt = _f32(pred, "pred")[0], _f32(gt, "gt")[0]
        m = None
        if mask is not None:
            m = _f32(mask, "mask").reshape(-1, pred.shape[1], pred.shape[2])
            metrics = self.config.get("metrics", {})
            if "res_aee" in metrics:
                yoff = (self.res[0] - metrics["res_aee"][0]) // 2
                xoff = (self.res[1] - metrics["res_aee"][1]) // 2
                m, pred, gt = m[:, yoff:-yoff, xoff:-xoff], pred[:, yoff:-yoff, xoff:-xoff], gt[:, yoff:-yoff, xoff:-xoff]
            if "vertical_crop_aee" in metrics:
                c = metrics["vertical_crop_aee"]
                m, pred, gt = m[:, :c, :], pred[:, :c, :], gt[:, :c, :]
            m = m.contiguous()
        pred, gt = pred.contiguous(), gt.contiguous()
        out = torch.empty((1,), dtype=torch.float32, device=pred.device)
        rc = _lib.lib().tef_val_aee(pred.data_ptr(), gt.data_ptr(), m.data_ptr() if m is not None else None,
                                    m.shape[0] if m is not None else 0, pred.shape[1], pred.shape[2], out.data_ptr(),
                                    _lib.stream_ptr())
        _lib.check(rc, "tef_val_aee")
        return out[0]


class Linear(BaseValidation):
    """Linear event warping validation class (reference loss/flow_val.py:317-416)."""

    def __init__(self, config, device):
        super().__init__(config, device)
        self._event_flow = None

    def update(self, flow_list, event_list, pol_mask, event_mask):
        ts, loc, pm = self.update_base(flow_list, event_list, pol_mask, event_mask)
        # flow for every new event from the latest map (:338-342)
        flow = self._event_step(self._flow_maps_x[-1], self._flow_maps_y[-1], loc.clone(), None, None, 0.0,
                                do_warp=False, want_flow=True)
        self._append("_event_flow", flow)
        self._passes += 1

    def reset(self):
        self.reset_base()
        self._event_flow = None

    def _fw_events(self):
        # event_propagation(ts, loc, flow, passes) (:399, :407), no purge
        return (self._event_loc + (self._passes - self._event_ts).unsqueeze(1) * self._event_flow).contiguous()

    def window_events(self, round_idx=False):
        return self.window_events_base(round_idx)

    def window_flow(self, mode=None, mask=None):
        if mask is None:
            mask = self.config["vis"]["mask_output"]
        fx, fy = self._flow_maps_x.clone(), self._flow_maps_y.clone()
        for i in range(self._passes - 1):
            wx, wy = self.forward_prop_flow(i, self._passes - 1, self._flow_maps_x, self._flow_maps_y)
            fx[i], fy[i] = wx[0, 0], wy[0, 0]
        return self.window_flow_base(fx, fy, mask=mask)

    def window_iwe(self, mode=None, round_idx=False):
        cnt, _ = self._event_image(self._fw_events(), self._event_pol_mask, None, round_idx)
        return cnt.unsqueeze(0)

    def rsat(self):
        return self.compute_rsat(self._fw_events(), self._event_loc, self._event_pol_mask, self._event_pol_mask,
                                 self._event_ts)

    def fwl(self):
        return self.compute_fwl(self._fw_events(), self._event_loc, self._event_pol_mask, self._event_pol_mask)


class Iterative(BaseValidation):
    """Iterative event warping validation class (reference loss/flow_val.py:419-694)."""

    def __init__(self, config, device):
        super().__init__(config, device)
        self._reset_iter()

    def _reset_iter(self):
        self._fw_event_loc = self._fw_event_warp_ts = self._fw_event_pol_mask = None
        self._bw_event_loc = self._bw_event_pol_mask = None
        self._fw_prop_flow_maps_x = self._fw_prop_flow_maps_y = None
        self._accum_flow_map_x = self._accum_flow_map_y = None
        self._flow_warping_indices = None
        self._flow_out_mask = None

    def update(self, flow_list, event_list, pol_mask, event_mask):
        ts, loc, pm = self.update_base(flow_list, event_list, pol_mask, event_mask)
        H, W = self._hw()
        fx_new, fy_new = self._flow_maps_x[-1], self._flow_maps_y[-1]

        # forward warping of EVERY event seen so far with the newest map, to time passes + 1 (:487-517)
        self._append("_fw_event_warp_ts", ts)
        self._append("_fw_event_loc", loc)
        self._append("_fw_event_pol_mask", pm)
        self._event_step(fx_new, fy_new, self._fw_event_loc, self._fw_event_warp_ts, self._fw_event_pol_mask,
                         self._passes + 1)

        # backward warping of the new events through maps passes, passes - 1, ..., 0 (:519-558)
        bw_loc, bw_ts, bw_pm = loc.clone(), ts.clone(), pm.clone()
        for k in range(self._passes, -1, -1):
            self._event_step(self._flow_maps_x[k], self._flow_maps_y[k], bw_loc, bw_ts, bw_pm, k)
        self._append("_bw_event_loc", bw_loc)
        self._append("_bw_event_pol_mask", bw_pm)

        # forward-propagated flow: every earlier map moves one step along itself (:560-576)
        self._append("_fw_prop_flow_maps_x", fx_new.unsqueeze(0))
        self._append("_fw_prop_flow_maps_y", fy_new.unsqueeze(0))
        for i in range(self._passes):
            self.forward_prop_flow(i, i + 1, self._fw_prop_flow_maps_x, self._fw_prop_flow_maps_y, inplace=True)

        # accumulated flow by backward warping of the pixel grid (:578-604)
        if self._flow_warping_indices is None:
            my, mx = torch.meshgrid(torch.arange(H, device=loc.device), torch.arange(W, device=loc.device), indexing="ij")
            self._flow_warping_indices = torch.stack([my, mx], dim=0).float().contiguous()
            self._flow_out_mask = torch.zeros((H, W), dtype=torch.float32, device=loc.device)
            self._accum_flow_map_x = torch.empty((H, W), dtype=torch.float32, device=loc.device)
            self._accum_flow_map_y = torch.empty((H, W), dtype=torch.float32, device=loc.device)
        rc = _lib.lib().tef_val_accum_flow(fx_new.data_ptr(), fy_new.data_ptr(), H, W,
                                           self._flow_warping_indices.data_ptr(), self._flow_out_mask.data_ptr(),
                                           self._accum_flow_map_x.data_ptr(), self._accum_flow_map_y.data_ptr(),
                                           _lib.stream_ptr())
        _lib.check(rc, "tef_val_accum_flow")
        self._passes += 1

    def reset(self):
        self.reset_base()
        self._reset_iter()

    def window_events(self, round_idx=False):
        return self.window_events_base(round_idx)

    def window_flow(self, mode=None, mask=None):
        if mask is None:
            mask = self.config["vis"]["mask_output"]
        if mode == "forward":
            return self.window_flow_base(self._fw_prop_flow_maps_x, self._fw_prop_flow_maps_y, mask=mask)
        if mode == "backward":
            return self.window_flow_base(self._accum_flow_map_x.unsqueeze(0), self._accum_flow_map_y.unsqueeze(0),
                                         mask=mask, divisor=self._flow_out_mask)
        return self.window_flow_base(self._flow_maps_x, self._flow_maps_y, mask=mask)

    def window_iwe(self, mode="forward", round_idx=False):
        if mode == "forward":
            loc, pm = self._fw_event_loc, self._fw_event_pol_mask
        elif mode == "backward":
            loc, pm = self._bw_event_loc, self._bw_event_pol_mask
        else:
            raise ValueError("Invalid IWE mode: {}".format(mode))
        cnt, _ = self._event_image(loc, pm, None, round_idx)
        return cnt.unsqueeze(0)

    def rsat(self):
        return self.compute_rsat(self._fw_event_loc, self._event_loc, self._fw_event_pol_mask, self._event_pol_mask,
                                 self._event_ts)

    def fwl(self):
        return self.compute_fwl(self._fw_event_loc, self._event_loc, self._fw_event_pol_mask, self._event_pol_mask)
