"""ctypes binding of libtef_hip.so (C ABI declared in include/tef.h).

The product path has NO fallback: if the HIP library is missing or a call fails, a RuntimeError is
raised.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""

import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TEF_HIP_LIB", os.path.join(PKG, "libtef_hip.so"))   # override: A/B builds only

TEF_MAX_PASSES = 64
TEF_MAX_SCALES = 6
KIND_ITERATIVE = 0
KIND_LINEAR = 1

_fp = ctypes.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class Events(ctypes.Structure):
    """struct tef_events (include/tef.h)"""

    _fields_ = [("ts", _fp), ("y", _fp), ("x", _fp), ("mp", _fp), ("mn", _fp), ("bin", _fp), ("cls", _fp),
                ("cap", ctypes.c_int)]


class UpdateDesc(ctypes.Structure):
    """struct tef_update_desc (include/tef.h)"""

    _fields_ = [("flows", _fp), ("stride_b", _fp), ("stride_c", _fp), ("ev", _fp), ("pm", _fp), ("ts_override", _fp),
                ("dev", _fp), ("dpm", _fp), ("dts_override", _fp), ("N", ctypes.c_int), ("Nd", ctypes.c_int),
                ("pass_idx", ctypes.c_int), ("slot0", ctypes.c_int), ("dslot0", ctypes.c_int)]


class LossCfg(ctypes.Structure):
    """struct tef_loss_cfg (include/tef.h)"""

    _fields_ = [
        ("kind", ctypes.c_int), ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
        ("P", ctypes.c_int), ("F", ctypes.c_int), ("S", ctypes.c_int), ("mode_div", ctypes.c_int),
        ("M", ctypes.c_int), ("Md", ctypes.c_int),
        ("off", ctypes.c_int * (TEF_MAX_PASSES + 1)), ("doff", ctypes.c_int * (TEF_MAX_PASSES + 1)),
        ("loss_scaling", ctypes.c_int), ("border_compensation", ctypes.c_int),
    ]


class ConvDesc(ctypes.Structure):
    """struct tef_conv_desc (include/tef.h)"""

    _fields_ = [(n, ctypes.c_int) for n in ("B", "C0", "C1", "H", "W", "N", "ksize", "stride", "act")]


class PackJob(ctypes.Structure):
    """struct tef_pack_job (include/tef.h): one weight part of tef_conv_pack_weights"""

    _fields_ = [("desc", ConvDesc), ("weight", ctypes.c_void_p), ("rows", ctypes.c_int), ("row0", ctypes.c_int),
                ("wp", ctypes.c_void_p), ("w2", ctypes.c_void_p)]


class GruDesc(ctypes.Structure):
    """struct tef_gru_desc (include/tef.h)"""

    _fields_ = [(n, ctypes.c_int) for n in ("B", "C", "H", "W")]


ACT = {None: 0, "relu": 1, "tanh": 2, "sigmoid": 3}

TEF_NET_MAX_LEVELS = 6
TEF_NET_MAX_RES = 4


class NetConv(ctypes.Structure):
    """struct tef_net_conv (include/tef.h)"""

    _fields_ = [("wp", _fp), ("w2", _fp), ("bias", _fp), ("dw", _fp), ("dw2", _fp), ("db", _fp), ("db2", _fp),
                ("defer", ctypes.c_int)]


class NetPlan(ctypes.Structure):
    """struct tef_net_plan (include/tef.h)"""

    _fields_ = [(n, ctypes.c_int) for n in ("B", "H", "W", "bins", "levels", "nres", "nout", "final_act")] + [
        ("width", ctypes.c_int * TEF_NET_MAX_LEVELS), ("dec_out", ctypes.c_int * TEF_NET_MAX_LEVELS),
        ("crop_top", ctypes.c_int), ("crop_left", ctypes.c_int), ("flow_scale", ctypes.c_float),
        ("head", NetConv * TEF_NET_MAX_LEVELS), ("gate_ur", NetConv * TEF_NET_MAX_LEVELS),
        ("gate_o", NetConv * TEF_NET_MAX_LEVELS), ("res1", NetConv * TEF_NET_MAX_RES), ("res2", NetConv * TEF_NET_MAX_RES),
        ("dec", NetConv * TEF_NET_MAX_LEVELS), ("pred", NetConv * TEF_NET_MAX_LEVELS),
        ("hn_ext", _fp * TEF_NET_MAX_LEVELS), ("dec_only", ctypes.c_int),
        ("copy_batch", ctypes.c_int), ("wgrad_ws", _fp), ("wgrad_ws_bytes", ctypes.c_size_t),
    ]

# name -> (restype, argtypes); every symbol include/tef.h declares
SIGNATURES = {
    "tef_version": (ctypes.c_int, []),
    "tef_last_error": (ctypes.c_char_p, []),
    "tef_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "tef_profile_pause": (ctypes.c_int, [ctypes.c_int]),
    "tef_profile_collect": (ctypes.c_int, []),
    "tef_profile_layers": (ctypes.c_long, [ctypes.c_char_p, ctypes.c_size_t]),
    "tef_profile_slots": (ctypes.c_int, []),
    "tef_profile_name": (ctypes.c_char_p, [ctypes.c_int]),
    "tef_profile_ms": (ctypes.c_double, [ctypes.c_int]),
    "tef_profile_calls": (ctypes.c_long, [ctypes.c_int]),
    "tef_pack_events": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _fp,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp,
                                       _fp, _fp, _fp, _fp, _fp]),
    "tef_pack_flow": (ctypes.c_int, [_fp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                     _fp, _fp]),
    "tef_pack_flows": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_long),
                                      ctypes.POINTER(ctypes.c_long), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      _fp, _fp, _fp]),
    "tef_update_pass": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_long),
                                       ctypes.POINTER(ctypes.c_long), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       _fp, _fp, _fp, _fp, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, _fp, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.POINTER(Events), ctypes.POINTER(Events), _fp]),
    "tef_update_window": (ctypes.c_int, [ctypes.POINTER(UpdateDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, _fp, _fp, ctypes.POINTER(Events), ctypes.POINTER(Events), _fp]),
    "tef_encode_events": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp]),
    "tef_encode_event_lists": (ctypes.c_int, [_fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp]),
    "tef_collate_counts": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "tef_collate_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "tef_collate_events": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.POINTER(ctypes.c_int), _fp,
                                          ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_size_t, _fp, _fp, _fp,
                                          _fp, _fp]),
    "tef_conv_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "tef_conv_packed_weight_floats": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_size_t),
                                                        ctypes.POINTER(ctypes.c_size_t)]),
    "tef_conv_pack_weight": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp]),
    "tef_conv_pack_weights": (ctypes.c_int, [ctypes.POINTER(PackJob), ctypes.c_int, _fp]),
    "tef_upsample_bilinear": (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_float, _fp, _fp]),
    "tef_upsample_bilinear_backward": (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_int, ctypes.c_float, _fp, _fp]),
    "tef_conv_forward": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_size_t,
                                        _fp]),
    "tef_conv_backward": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                         ctypes.c_size_t, _fp]),
    "tef_conv_forward_split": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                              _fp, ctypes.c_size_t, _fp]),
    "tef_conv_forward_blend": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                              _fp, _fp, _fp, _fp, ctypes.c_size_t, _fp]),
    "tef_conv_backward_split": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                               ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, _fp,
                                               ctypes.c_size_t, _fp]),
    "tef_conv_backward_keep": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                              ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, _fp, _fp,
                                              ctypes.c_size_t, _fp]),
    "tef_conv_wgrad_parts_supported": (ctypes.c_int, [ctypes.POINTER(ConvDesc)]),
    "tef_conv_wgrad_parts": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                            ctypes.POINTER(ctypes.c_void_p), _fp, _fp, ctypes.c_int, _fp]),
    "tef_upsample_bilinear_crop": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, _fp, _fp]),
    "tef_upsample_bilinear_crop_backward": (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                           ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, _fp,
                                                           _fp]),
    "tef_upsample2x_pair": (ctypes.c_int, [_fp, _fp, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, _fp]),
    "tef_upsample2x_pair_backward": (ctypes.c_int, [_fp, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int,
                                                    _fp]),
    "tef_convgru_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(GruDesc)]),
    "tef_convgru_cell_fwd": (ctypes.c_int, [ctypes.POINTER(GruDesc)] + [_fp] * 10 + [_fp, ctypes.c_size_t, _fp]),
    "tef_convgru_cell_bwd": (ctypes.c_int, [ctypes.POINTER(GruDesc)] + [_fp] * 5
                             + [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int] + [_fp] * 12 + [_fp, ctypes.c_size_t, _fp]),
    "tef_convgru_cell_bwd_head": (ctypes.c_int, [ctypes.POINTER(GruDesc)] + [_fp] * 5
                                  + [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int] + [_fp] * 12 + [ctypes.c_int, _fp, _fp]
                                  + [_fp, ctypes.c_size_t, _fp]),
    "tef_grad_act": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, _fp, _fp, _fp]),
    "tef_conv_backward_post": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _fp, _fp, _fp, _fp, ctypes.c_void_p, _fp, ctypes.c_size_t, _fp]),
    "tef_dec_head_backward": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _fp, _fp, _fp,
                                             ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "tef_add_act": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_size_t, _fp, _fp]),
    "tef_net_tape_floats": (ctypes.c_size_t, [ctypes.POINTER(NetPlan)]),
    "tef_net_gtape_floats": (ctypes.c_size_t, [ctypes.POINTER(NetPlan)]),
    "tef_net_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(NetPlan)]),
    "tef_net_layout": (ctypes.c_int, [ctypes.POINTER(NetPlan)] + [ctypes.POINTER(ctypes.c_size_t)] * 4),
    "tef_net_pass_forward": (ctypes.c_int, [ctypes.POINTER(NetPlan), _fp, ctypes.POINTER(ctypes.c_void_p), _fp, _fp,
                                            ctypes.c_size_t, _fp]),
    "tef_net_pass_backward": (ctypes.c_int, [ctypes.POINTER(NetPlan), _fp, ctypes.POINTER(ctypes.c_void_p), _fp,
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                             _fp, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_int),
                                             ctypes.POINTER(ctypes.c_int), _fp, ctypes.c_size_t, _fp]),
    "tef_net_pass_forward_part": (ctypes.c_int, [ctypes.POINTER(NetPlan), ctypes.c_int, _fp, ctypes.POINTER(ctypes.c_void_p), _fp,
                                                 _fp, ctypes.c_size_t, _fp]),
    "tef_net_pass_backward_part": (ctypes.c_int, [ctypes.POINTER(NetPlan), ctypes.c_int, _fp, ctypes.POINTER(ctypes.c_void_p), _fp,
                                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                                  _fp, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_longlong),
                                                  ctypes.POINTER(ctypes.c_int), _fp, ctypes.c_size_t, _fp]),
    "tef_net_pass_backward_part2": (ctypes.c_int, [ctypes.POINTER(NetPlan), ctypes.c_int, _fp, ctypes.POINTER(ctypes.c_void_p), _fp,
                                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                                   ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                                   _fp, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_longlong),
                                                   ctypes.POINTER(ctypes.c_int), _fp, ctypes.c_size_t, _fp]),
    "tef_net_pass_forward_levels": (ctypes.c_int, [ctypes.POINTER(NetPlan), ctypes.c_int, ctypes.c_int, _fp,
                                                   ctypes.POINTER(ctypes.c_void_p), _fp, _fp, ctypes.c_size_t, _fp]),
    "tef_net_pass_backward_levels": (ctypes.c_int, [ctypes.POINTER(NetPlan), ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                                    ctypes.POINTER(ctypes.c_void_p), _fp, ctypes.POINTER(ctypes.c_void_p),
                                                    ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _fp,
                                                    ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_longlong),
                                                    ctypes.POINTER(ctypes.c_int), _fp, ctypes.c_size_t, _fp]),
    "tef_net_window_wgrads": (ctypes.c_int, [ctypes.POINTER(NetPlan), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.POINTER(ctypes.c_void_p)), ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_ulonglong), _fp]),
    "tef_net_window_wgrads_workspace": (ctypes.c_size_t, [ctypes.POINTER(NetPlan), ctypes.c_int]),
    "tef_net_window_wgrads_part": (ctypes.c_int, [ctypes.POINTER(NetPlan), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                                  ctypes.POINTER(ctypes.POINTER(ctypes.c_void_p)), ctypes.POINTER(ctypes.c_void_p),
                                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_ulonglong), _fp]),
    "tef_l2_norm_scratch_bytes": (ctypes.c_size_t, []),
    "tef_l2_norm": (ctypes.c_int, [_fp, ctypes.c_size_t, _fp, _fp, _fp, _fp]),
    "tef_adam_clip_step": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_size_t, _fp, ctypes.c_float, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_double, _fp, _fp]),
    "tef_adam_clip_step_hp": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_size_t, _fp, _fp, _fp, _fp]),
    "tef_gru_blend": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_size_t, _fp, _fp]),
    "tef_gru_blend_backward": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_size_t, _fp, _fp, _fp, _fp]),
    "tef_val_event_step": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_int, _fp, _fp]),
    "tef_val_event_image": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp,
                                           _fp]),
    "tef_val_metrics_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "tef_val_metrics": (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _fp, _fp,
                                       ctypes.c_size_t, _fp]),
    "tef_val_forward_prop_flow": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _fp, _fp, _fp,
                                                 _fp]),
    "tef_val_accum_flow": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp]),
    "tef_val_average_flow": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, ctypes.c_int,
                                            _fp, _fp]),
    "tef_pol_iwe": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, _fp, _fp]),
    "tef_interp_corners": (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp,
                                          _fp]),
    "tef_scatter_add": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp]),
    "tef_event_flow": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, _fp, _fp]),
    "tef_event_flow_backward": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, _fp, _fp, _fp,
                                               _fp, _fp]),
    "tef_interp_corners_backward": (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp]),
    "tef_scatter_add_backward": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp]),
    "tef_val_aee": (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp]),
    "tef_loss_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(LossCfg)]),
    "tef_loss_forward": (ctypes.c_int, [ctypes.POINTER(LossCfg), _fp, ctypes.POINTER(Events), ctypes.POINTER(Events),
                                        _fp, ctypes.c_size_t, _fp, _fp]),
    "tef_loss_backward": (ctypes.c_int, [ctypes.POINTER(LossCfg), _fp, ctypes.POINTER(Events), ctypes.POINTER(Events),
                                         _fp, ctypes.c_size_t, _fp, _fp, _fp]),
    "tef_smoothing_scratch_bytes": (ctypes.c_size_t, [ctypes.POINTER(LossCfg)]),
    "tef_smoothing_forward": (ctypes.c_int, [ctypes.POINTER(LossCfg), _fp, ctypes.c_float, ctypes.c_float, _fp, _fp,
                                             _fp]),
    "tef_smoothing_backward": (ctypes.c_int, [ctypes.POINTER(LossCfg), _fp, ctypes.c_float, ctypes.c_float, _fp, _fp,
                                              _fp, _fp]),
}

_lib = None


def lib():
    """Load libtef_hip.so once; fail loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `python taming_event_flow_amd/build.py`). There is no CPU fallback."
            )
        # PyTorch-ROCm bundles its own HIP runtime: load it first so that the library binds to the runtime that owns
        # the tensors' device context (loading libtef_hip.so before torch pulls in /opt/rocm's copy next to torch's, and
        # launches then fail with "no ROCm-capable device is detected")
        import torch  # noqa: F401

        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().tef_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def require_device_tensor(t, name):
    """The HIP path only runs on device tensors; anything else is an error, not a fallback."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); the MI355X path has no CPU fallback")
    return t


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
