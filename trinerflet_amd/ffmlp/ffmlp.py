"""FFMLP -- the module surface of aux_libs/ffmlp/ffmlp.py:104-170 (the reference's `--ff` back-end: tiny-cuda-nn style
fully-fused MLP on CUTLASS / WMMA, "turned off by default ... performance is not good enough", aux_libs/scripts/install_ext.sh:9-10)
so that `reconstruction/nerf/network_ff.py` constructs and runs: the same constructor, the same ONE flat `weights` parameter
(state-dict key `weights`; hidden * (input_dim + hidden * (num_layers - 1) + padded_output) elements, matrices row-major
[out, in] one after the other: first layer, num_layers - 1 hidden layers, the output layer padded to 16 rows -- the layout
ffmlp.cu:369-403 walks), the same initialisation (uniform +-sqrt(3 / hidden) under manual_seed(42)), the same arithmetic
contract (fp16 operands and activations, ReLU between layers, no output activation, fp16 result of the first output_dim
columns).

The matrices run as rocBLAS fp16 GEMMs (torch): this is the MODULE-API tier (SURVEY.md 8(f)-4's last clause), not a hot
path -- note that FFMLP(num_layers = n) holds n + 1 matrices (ffmlp.py:115: "num_layers >= 2 (3 matmuls)"), one more than
network.py's n bias-free Linear layers, so the `--ff` network is another architecture than the one the hand-written field
kernel (csrc/field.hip) is built for; README configurations do not pass `--ff`."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class FFMLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation='relu'):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        if activation != 'relu':
            raise NotImplementedError("FFMLP: only the 'relu' hidden activation network_ff.py uses is built")
        self.activation = 0                     # convert_activation('relu'), ffmlp.py:93-100
        self.output_activation = 6              # 'none'
        self.tensorcore_width = 16
        # ffmlp.py:112-115
        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"
        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        # ffmlp.py:141-144 -- including the reference's re-seeding of the global generator
        torch.manual_seed(42)
        std = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-std, std)

    def matrices(self, weights=None):
        """The flat parameter as its matrices [out, in] (views): first layer, hidden layers, padded output layer."""
        w = self.weights if weights is None else weights
        H, off, out = self.hidden_dim, 0, []
        for rows, cols in [(H, self.input_dim)] + [(H, H)] * (self.num_layers - 1) + [(self.padded_output_dim, H)]:
            out.append(w[off:off + rows * cols].view(rows, cols))
            off += rows * cols
        assert off == self.num_parameters
        return out

    def forward(self, inputs):
        # inputs: [B, input_dim] -> [B, output_dim] in fp16 (custom_fwd(cast_inputs=torch.half), ffmlp.py:17)
        with torch.autocast(device_type=inputs.device.type, enabled=False):
            h = inputs.to(torch.float16)
            mats = self.matrices(self.weights.to(torch.float16))
            for W in mats[:-1]:
                h = F.relu(F.linear(h, W))
            out = F.linear(h, mats[-1])
        return out[:, :self.output_dim]
