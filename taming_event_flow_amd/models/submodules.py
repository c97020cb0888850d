"""Building blocks of RecEVFlowNet — drop-in for the reference's ``models/submodules.py``.

Same classes, constructor arguments, parameter names / shapes (so ``state_dict`` keys match) and initialisation
(ConvLayer :33-39 uniform(+-sqrt(1/Cin)) or ``w_scale``; ConvGRU :127-132 orthogonal weights, zero bias).
Every convolution (forward, input gradient, weight gradient) runs on the fp32 MFMA GEMM of libtef_hip.so
(tef_conv.hip); only tensor plumbing (padding, concatenation, bilinear resize, additions) is left to torch.
"""

import ctypes
import math

import torch
import torch.nn as nn
import torch.nn.functional as f

from .. import _lib


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _ConvFn(torch.autograd.Function):
    """act(conv2d(cat[x0, x1 * gate1], weight, bias)) with padding k//2; see include/tef.h tef_conv_forward."""

    @staticmethod
    def forward(ctx, x0, x1, gate1, weight, bias, stride, act):
        lib = _lib.lib()
        _lib.require_device_tensor(x0, "conv input")
        x0 = x0.contiguous()
        x1 = x1.contiguous() if x1 is not None else None
        gate1 = gate1.contiguous() if gate1 is not None else None
        weight = weight.contiguous()
        B, C0, H, W = x0.shape
        C1 = x1.shape[1] if x1 is not None else 0
        N, Ct, k, _ = weight.shape
        assert Ct == C0 + C1, (Ct, C0, C1)
        d = _lib.ConvDesc(B, C0, C1, H, W, N, k, stride, _lib.ACT[act])
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = torch.empty((B, N, Ho, Wo), dtype=torch.float32, device=x0.device)
        nbytes = lib.tef_conv_workspace_bytes(ctypes.byref(d))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x0.device)
        rc = lib.tef_conv_forward(ctypes.byref(d), x0.data_ptr(), _ptr(x1), _ptr(gate1), weight.data_ptr(), _ptr(bias),
                                  out.data_ptr(), ws.data_ptr(), nbytes, _lib.stream_ptr())
        _lib.check(rc, "tef_conv_forward")
        ctx.desc = d
        ctx.has = (x1 is not None, gate1 is not None, bias is not None)
        ctx.save_for_backward(x0, x1, gate1, weight, out if act is not None else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.lib()
        x0, x1, gate1, weight, out = ctx.saved_tensors
        d = ctx.desc
        has_x1, has_gate, has_bias = ctx.has
        need = ctx.needs_input_grad
        dout = dout.contiguous()
        dx0 = torch.empty_like(x0) if need[0] else None
        dxg = torch.empty_like(x1) if has_x1 and (need[1] or need[2]) else None
        dw = torch.zeros_like(weight) if need[3] else None
        db = torch.zeros((d.N,), dtype=torch.float32, device=dout.device) if (has_bias and need[4]) else None
        nbytes = lib.tef_conv_workspace_bytes(ctypes.byref(d))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dout.device)
        rc = lib.tef_conv_backward(ctypes.byref(d), x0.data_ptr(), _ptr(x1), _ptr(gate1), weight.data_ptr(), _ptr(out),
                                   dout.data_ptr(), _ptr(dx0), _ptr(dxg), _ptr(dw), _ptr(db), ws.data_ptr(), nbytes,
                                   _lib.stream_ptr())
        _lib.check(rc, "tef_conv_backward")
        dx1 = dgate = None
        if dxg is not None:
            if has_gate:
                dx1 = dxg * gate1 if need[1] else None
                dgate = dxg * x1 if need[2] else None
            else:
                dx1 = dxg
        return dx0, dx1, dgate, dw, db, None, None


def conv2d(x0, weight, bias, stride=1, act=None, x1=None, gate1=None):
    return _ConvFn.apply(x0, x1, gate1, weight, bias, stride, act)


class _GruBlendFn(torch.autograd.Function):
    """new_state = prev_state * (1 - update) + out_inputs * update (reference submodules.py:150)."""

    @staticmethod
    def forward(ctx, h, u, o):
        h, u, o = h.contiguous(), u.contiguous(), o.contiguous()
        out = torch.empty_like(h)
        rc = _lib.lib().tef_gru_blend(h.data_ptr(), u.data_ptr(), o.data_ptr(), h.numel(), out.data_ptr(),
                                      _lib.stream_ptr())
        _lib.check(rc, "tef_gru_blend")
        ctx.save_for_backward(h, u, o)
        return out

    @staticmethod
    def backward(ctx, dhn):
        h, u, o = ctx.saved_tensors
        dhn = dhn.contiguous()
        dh, du, do = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        rc = _lib.lib().tef_gru_blend_backward(dhn.data_ptr(), h.data_ptr(), u.data_ptr(), o.data_ptr(), h.numel(),
                                               dh.data_ptr(), du.data_ptr(), do.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tef_gru_blend_backward")
        return dh, du, do


def _act_name(activation):
    if activation is None:
        return None
    if activation not in ("relu", "tanh", "sigmoid"):
        raise NotImplementedError(f"activation '{activation}' has no fused HIP epilogue (relu/tanh/sigmoid/None)")
    return activation


def _no_norm(norm):
    if norm is not None:
        raise NotImplementedError("norm layers are unused by RecEVFlowNet (norm=None, reference models/model.py:28)")


class ConvLayer(nn.Module):
    """Convolutional layer. Default: bias, ReLU, no downsampling, no batch norm (reference submodules.py:8-62)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, activation="relu", norm=None,
                 BN_momentum=0.1, w_scale=None, padding=None, bias=None):
        super().__init__()
        _no_norm(norm)
        if padding is not None and padding != kernel_size // 2:
            raise NotImplementedError("only padding = kernel_size // 2 is implemented")
        if bias is None:
            bias = True
        if w_scale is None:
            w_scale = math.sqrt(1 / in_channels)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=bias)
        nn.init.uniform_(self.conv2d.weight, -w_scale, w_scale)
        if bias:
            nn.init.zeros_(self.conv2d.bias)
        self.activation = _act_name(activation)
        self.stride = stride
        self.norm = norm

    def forward(self, x):
        return conv2d(x, self.conv2d.weight, self.conv2d.bias, self.stride, self.activation)


class ConvGRU(nn.Module):
    """Convolutional GRU cell (reference submodules.py:111-152)."""

    def __init__(self, input_size, hidden_size, kernel_size, activation=None):
        super().__init__()
        padding = kernel_size // 2
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.reset_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.update_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.out_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        assert activation is None, "ConvGRU activation cannot be set (just for compatibility)"
        nn.init.orthogonal_(self.reset_gate.weight)
        nn.init.orthogonal_(self.update_gate.weight)
        nn.init.orthogonal_(self.out_gate.weight)
        nn.init.constant_(self.reset_gate.bias, 0.0)
        nn.init.constant_(self.update_gate.bias, 0.0)
        nn.init.constant_(self.out_gate.bias, 0.0)

    def forward(self, input_, prev_state):
        if prev_state is None:
            prev_state = torch.zeros((input_.shape[0], self.hidden_size) + tuple(input_.shape[2:]),
                                     dtype=input_.dtype, device=input_.device)
        C = self.hidden_size
        # update and reset gates share their input: one GEMM with 2C output channels
        w_ur = torch.cat([self.update_gate.weight, self.reset_gate.weight], dim=0)
        b_ur = torch.cat([self.update_gate.bias, self.reset_gate.bias], dim=0)
        ur = conv2d(input_, w_ur, b_ur, 1, "sigmoid", x1=prev_state)
        update, reset = ur[:, :C].contiguous(), ur[:, C:].contiguous()
        # tanh(out_gate(cat[input_, prev_state * reset])): the product is formed inside the im2col gather
        out_inputs = conv2d(input_, self.out_gate.weight, self.out_gate.bias, 1, "tanh", x1=prev_state, gate1=reset)
        new_state = _GruBlendFn.apply(prev_state, update, out_inputs)
        return new_state, new_state


class RecurrentConvLayer(nn.Module):
    """Convolution followed by a recurrent convolutional block (reference submodules.py:65-108)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, recurrent_block_type="convgru",
                 activation_ff="relu", activation_rec=None, norm=None, BN_momentum=0.1):
        super().__init__()
        assert recurrent_block_type in ["convgru"]
        self.recurrent_block_type = recurrent_block_type
        self.conv = ConvLayer(in_channels, out_channels, kernel_size, stride, activation_ff, norm,
                              BN_momentum=BN_momentum)
        self.recurrent_block = ConvGRU(input_size=out_channels, hidden_size=out_channels, kernel_size=3,
                                       activation=activation_rec)

    def forward(self, x, prev_state):
        x = self.conv(x)
        x, state = self.recurrent_block(x, prev_state)
        return x, state


class ResidualBlock(nn.Module):
    """Residual block (reference submodules.py:155-227): conv-act-conv, += x, act; returns (out2, out1)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, activation="relu", downsample=None,
                 norm=None, BN_momentum=0.1):
        super().__init__()
        _no_norm(norm)
        if downsample is not None:
            raise NotImplementedError("downsample is unused by RecEVFlowNet")
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=kernel_size // 2, bias=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=kernel_size // 2, bias=True)
        self.activation = _act_name(activation)
        self.stride = stride
        self.norm = norm
        self.downsample = downsample

    def forward(self, x):
        out1 = conv2d(x, self.conv1.weight, self.conv1.bias, self.stride, self.activation)
        out2 = conv2d(out1, self.conv2.weight, self.conv2.bias, self.stride, None)
        out2 = out2 + x
        if self.activation is not None:
            out2 = getattr(torch, self.activation)(out2)
        return out2, out1


class UpsampleConvLayer(nn.Module):
    """Bilinear x2 (align_corners=False) + Conv2d + activation (reference submodules.py:230-273)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, activation="relu", norm=None):
        super().__init__()
        _no_norm(norm)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=True)
        self.activation = _act_name(activation)
        self.stride = stride
        self.norm = norm

    def forward(self, x):
        x_upsampled = f.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        return conv2d(x_upsampled, self.conv2d.weight, self.conv2d.bias, self.stride, self.activation)


class TransposedConvLayer(nn.Module):
    """Present in the reference (submodules.py:276-325) but never built by RecEVFlowNet (use_upsample_conv=True)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("TransposedConvLayer is unused by RecEVFlowNet (use_upsample_conv=True)")
