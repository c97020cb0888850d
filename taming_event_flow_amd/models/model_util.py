"""Helpers RecEVFlowNet's callers know by name (reference models/model_util.py): state copies and the padder.

`ImagePadder` survives as the geometry record of the top / left padding (reference :29-71, after E-RAFT): the fused pass
pads and crops inside its own kernels, so `pad` / `unpad` are only used by code that calls them explicitly."""
import torch


def recursive_clone(obj):
    """Structure-preserving clone: tensors are cloned, lists / tuples rebuilt, None kept (reference :6-18)."""
    if obj is None:
        return None
    if isinstance(obj, torch.Tensor):
        return obj.clone()
    if isinstance(obj, (list, tuple)):
        return type(obj)(recursive_clone(o) for o in obj)
    raise TypeError(f"cannot clone a recurrent state of type {type(obj).__name__}")


def copy_states(states):
    """A copy of the per-encoder state list the caller may keep or modify (reference :20-27)."""
    return recursive_clone(list(states))


class ImagePadder:
    def __init__(self, min_size=64):
        self.min_size = min_size
        self.pad_height = self.pad_width = None

    def amounts(self, height, width):
        m = self.min_size
        return (m - height % m) % m, (m - width % m) % m

    def pad(self, image):
        ph, pw = self.amounts(*image.shape[-2:])
        if self.pad_height is None:
            self.pad_height, self.pad_width = ph, pw
        elif (ph, pw) != (self.pad_height, self.pad_width):
            raise RuntimeError("ImagePadder: the input size changed between calls")     # the reference has a bare `raise`
        if not (ph or pw):
            return image
        out = image.new_zeros(image.shape[:-2] + (image.shape[-2] + ph, image.shape[-1] + pw))
        out[..., ph:, pw:] = image
        return out

    def unpad(self, image):
        return image[..., self.pad_height:, self.pad_width:]
