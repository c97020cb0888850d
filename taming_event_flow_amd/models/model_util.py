"""State helpers and the E-RAFT image padder (reference models/model_util.py)."""
import copy

import torch


def recursive_clone(tensor):
    """Clone a tensor or a nested iterable of tensors (reference model_util.py:6-18)."""
    if hasattr(tensor, "clone"):
        return tensor.clone()
    try:
        return type(tensor)(recursive_clone(t) for t in tensor)
    except TypeError:
        print("{} is not iterable and has no clone() method.".format(tensor))


def copy_states(states):
    """Deepcopy a list of Nones, clone otherwise (reference model_util.py:20-27)."""
    if states[0] is None:
        return copy.deepcopy(states)
    return recursive_clone(states)


class ImagePadder(object):
    """Pads on the LEFT and TOP to a multiple of min_size (reference model_util.py:29-71, from E-RAFT)."""

    def __init__(self, min_size=64):
        self.min_size = min_size
        self.pad_height = None
        self.pad_width = None

    def pad(self, image):
        height, width = image.shape[-2:]
        pad_height = (self.min_size - height % self.min_size) % self.min_size
        pad_width = (self.min_size - width % self.min_size) % self.min_size
        if self.pad_width is None:
            self.pad_height, self.pad_width = pad_height, pad_width
        elif pad_height != self.pad_height or pad_width != self.pad_width:
            raise RuntimeError("ImagePadder: input size changed between calls")   # the reference has a bare `raise`
        return torch.nn.ZeroPad2d((self.pad_width, 0, self.pad_height, 0))(image)

    def unpad(self, image):
        return image[..., self.pad_height:, self.pad_width:]
