import math
import torch
from torch import nn


class Linear(nn.Module):
    """PyG nn.dense.linear.Linear: y = x W^T (+ b); default init kaiming-uniform(a=sqrt(5)),
    'glorot' on request; bias -> zeros when weight_initializer given else uniform(+-1/sqrt(fan_in))."""
    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None, bias_initializer=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer, self.bias_initializer = weight_initializer, bias_initializer
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight_initializer == "glorot":
            a = math.sqrt(6.0 / (self.weight.size(-2) + self.weight.size(-1)))
            nn.init.uniform_(self.weight, -a, a)
        else:
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            if self.bias_initializer == "zeros":
                nn.init.zeros_(self.bias)
            else:
                bound = 1.0 / math.sqrt(self.in_channels) if self.in_channels > 0 else 0
                nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight, self.bias)
