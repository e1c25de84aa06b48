"""Operator seam of the reference (nerfstudio field_components), backed by the HIP library.

Mirrors, with the same constructor arguments, attribute names and state-dict keys:
    HashEncoding   ns/field_components/encodings.py:251-389   (parameter `hash_table` [L*T, F], torch layout)
    SHEncoding     ns/field_components/encodings.py:679-719
    MLP            ns/field_components/mlp.py:65-179           (`layers.{i}.weight/bias`, nn.Linear layout)
    Embedding      ns/field_components/embedding.py:27-55
    SceneContraction ns/field_components/spatial_distortions.py:42-90 (order=inf only)
    trunc_exp      ns/field_components/activations.py:28-52

`implementation` accepts the reference's strings for config compatibility ("tcnn", "tcnn+fp32", "torch") and "hip";
every value runs the HIP kernels (there is no tinycudann and no torch fallback here).  Tensors must live on the GPU."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from . import ops

IMPLEMENTATIONS = ("tcnn", "torch", "tcnn+fp32", "hip")


_WARNED_TCNN = False


def _check_impl(implementation: str):
    global _WARNED_TCNN
    if implementation not in IMPLEMENTATIONS:
        raise ValueError(f"unknown implementation {implementation!r}")
    if implementation.startswith("tcnn") and not _WARNED_TCNN:
        import warnings

        _WARNED_TCNN = True
        warnings.warn(
            f"presight_amd: implementation={implementation!r} runs the semantics of nerfstudio's pure-torch path (all levels hashed, "
            "scale = floor(base * g^l), ceil/floor corners; ns/field_components/encodings.py:324-384) on the HIP kernels.  "
            "tiny-cuda-nn's grid is a different function of different parameters (dense coarse levels, +0.5 offset, flat params): a "
            "checkpoint TRAINED with tcnn is not table-compatible; checkpoints of the 'torch' implementation load unchanged.",
            stacklevel=3)


def hash_scalings(num_levels: int, min_res: int, max_res: int) -> Tensor:
    """floor(min_res * g**l) with g**l evaluated by torch in fp32, exactly as the reference does
    (ns/field_components/encodings.py:281-284); decides e.g. 2047 vs 2048 for the finest level."""
    levels = torch.arange(num_levels)
    g = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * g**levels).to(torch.float32)


class HashEncoding(nn.Module):
    def __init__(self, num_levels: int = 16, min_res: int = 16, max_res: int = 1024, log2_hashmap_size: int = 19,
                 features_per_level: int = 2, hash_init_scale: float = 0.001, implementation: str = "tcnn+fp32",
                 interpolation: Optional[str] = None) -> None:
        super().__init__()
        _check_impl(implementation)
        if interpolation not in (None, "Linear"):
            raise AssertionError(f"interpolation '{interpolation}' is not supported")
        if features_per_level not in (1, 2, 4):
            raise ValueError("features_per_level must be 1, 2 or 4 for the HIP hash grid")
        self.in_dim = 3
        self.num_levels = num_levels
        self.features_per_level = features_per_level
        self.log2_hashmap_size = log2_hashmap_size
        self.hash_table_size = 2**log2_hashmap_size
        self.scalings = hash_scalings(num_levels, min_res, max_res)
        self.hash_offset = torch.arange(num_levels) * self.hash_table_size
        self.tcnn_encoding = None
        table = torch.rand(size=(self.hash_table_size * num_levels, features_per_level)) * 2 - 1
        self.hash_table = nn.Parameter(table * hash_init_scale)
        self._scalings_dev = {}

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def scalings_on(self, device) -> Tensor:
        key = str(device)
        if key not in self._scalings_dev:
            self._scalings_dev[key] = self.scalings.to(device)
        return self._scalings_dev[key]

    def forward(self, in_tensor: Tensor) -> Tensor:
        assert in_tensor.shape[-1] == 3
        flat = in_tensor.reshape(-1, 3)
        out = ops.hashgrid_encode(flat, self.hash_table, self.scalings_on(flat.device), self.num_levels,
                                  self.features_per_level, self.log2_hashmap_size)
        return out.view(*in_tensor.shape[:-1], self.get_out_dim())


class SHEncoding(nn.Module):
    def __init__(self, levels: int = 4, implementation: str = "torch") -> None:
        super().__init__()
        _check_impl(implementation)
        if levels <= 0 or levels > 4:
            raise ValueError(f"Spherical harmonic encoding only supports 1 to 4 levels, requested {levels}")
        self.in_dim = 3
        self.levels = levels
        self.tcnn_encoding = None

    def get_out_dim(self) -> int:
        return self.levels**2

    @torch.no_grad()
    def forward(self, in_tensor: Tensor) -> Tensor:
        """The reference evaluates the harmonics on its input as given (ns/field_components/encodings.py:711-714); the fields
        hand it the shifted direction (d+1)/2 (ns/fields/base_field.py:136-142)."""
        return ops.sh_encode(in_tensor.reshape(-1, 3), self.levels).reshape(*in_tensor.shape[:-1], self.levels**2)


class MLP(nn.Module):
    def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: Optional[int] = None,
                 skip_connections: Optional[Tuple[int]] = None, activation: Optional[nn.Module] = nn.ReLU(),
                 out_activation: Optional[nn.Module] = None, implementation: str = "torch") -> None:
        super().__init__()
        _check_impl(implementation)
        assert in_dim > 0
        if skip_connections:
            raise NotImplementedError("presight_amd MLP: skip connections are not used on the PreSight path")
        if not isinstance(activation, nn.ReLU):
            raise NotImplementedError("presight_amd MLP: hidden activation must be ReLU")
        if out_activation is not None and not isinstance(out_activation, nn.Sigmoid):
            raise NotImplementedError("presight_amd MLP: output activation must be None or Sigmoid")
        self.in_dim = in_dim
        self.out_dim = out_dim if out_dim is not None else layer_width
        self.num_layers = num_layers
        self.layer_width = layer_width
        self.skip_connections = skip_connections
        self.activation = activation
        self.out_activation = out_activation
        self.tcnn_encoding = None
        dims = [in_dim] + [layer_width] * (num_layers - 1) + [self.out_dim]
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers)])

    def layer_params(self):
        return [(l.weight, l.bias) for l in self.layers]

    def forward(self, in_tensor: Tensor) -> Tensor:
        flat = in_tensor.reshape(-1, self.in_dim)
        y = ops.mlp(flat, self.layer_params(), out_act="sigmoid" if self.out_activation is not None else None)
        return y.view(*in_tensor.shape[:-1], self.out_dim)


class Embedding(nn.Module):
    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.embedding = nn.Embedding(in_dim, out_dim)

    def mean(self, dim=0):
        return self.embedding.weight.mean(dim)

    def forward(self, in_tensor: Tensor) -> Tensor:
        return self.embedding(in_tensor)


class SceneContraction(nn.Module):
    """L-inf contraction.  The fields only use it as a marker (the maths runs inside ps_field_points)."""

    def __init__(self, order=None) -> None:
        super().__init__()
        if order != float("inf"):
            raise NotImplementedError("presight_amd: only SceneContraction(order=inf) is used by PreSight")
        self.order = order

    def forward(self, positions: Tensor) -> Tensor:
        # stand-alone use only (never on the hot path, where ps_field_points fuses normalise+contract+mask)
        mag = positions.abs().amax(dim=-1, keepdim=True)
        return torch.where(mag < 1, positions, (2 - (1 / mag)) * (positions / mag))


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply
