"""Base class for all models (reference models/base.py, adapted there from UZH-RPG rpg_e2vid)."""
from abc import abstractmethod

import numpy as np
import torch


class BaseModel(torch.nn.Module):
    @abstractmethod
    def forward(self, *inputs):
        raise NotImplementedError

    def __str__(self):
        """Model prints with number of trainable parameters (reference models/base.py:25-31)."""
        params = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + "\nTrainable parameters: {}".format(params)
