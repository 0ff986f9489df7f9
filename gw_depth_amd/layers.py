"""Parameter containers with the reference's state-dict names and kernel-native storage.

Convolution weights live in HBM as (Cout, KH, KW, Cin) — the layout the implicit-GEMM kernels
read — and are converted to / from the reference's (Cout, Cin, KH, KW) only inside
state_dict() / load_state_dict(), so reference checkpoints load unchanged
(/root/reference/src/main_glassrgbd.py:104-157).
"""
import math

import torch
from torch import nn

from . import ops


class Conv(nn.Module):
    """nn.Conv2d stand-in (weight [+ bias]); forward is done by the owner through ops.conv2d."""

    def __init__(self, cin, cout, k, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, k, k, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        fan_in, fan_out = cin * k * k, cout * k * k
        bound = math.sqrt(6.0 / (fan_in + fan_out))          # xavier_uniform, as the reference's _init_weights
        nn.init.uniform_(self.weight, -bound, bound)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        destination[prefix + "weight"] = destination[prefix + "weight"].permute(0, 3, 1, 2).contiguous()

    def _load_from_state_dict(self, state_dict, prefix, *args):
        key = prefix + "weight"
        if key in state_dict and state_dict[key].dim() == 4:
            state_dict[key] = state_dict[key].permute(0, 2, 3, 1).contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args)


class Linear(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        nn.init.trunc_normal_(self.weight, std=0.02)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def forward(self, x, act=ops.ACT_NONE, residual=None, mult=None, fan=False, defer=False, in_gate=ops.ACT_NONE, gate_src=None):
        return ops.linear(x, self.weight, self.bias, act, residual=residual, mult=mult, fanout=fan, defer=defer, in_gate=in_gate, gate_src=gate_src)


class LayerNorm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))

    def forward(self, x, gelu=False, residual=None, fan=False, in_gate=0):
        """fan: (y, x') with x' = x for the input's second consumer (the skip of a pre-norm block), see ops.layer_norm."""
        return ops.layer_norm(x, self.weight, self.bias, gelu, residual=residual, fanout=fan, in_gate=in_gate)


class FrozenBN(nn.Module):
    """FrozenBatchNorm2d (/root/reference/src/models/backbone.py:19-55): four buffers, folded into the
    preceding convolution as a per-channel weight scale and an epilogue shift."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._cache = None

    def _load_from_state_dict(self, state_dict, prefix, *args):
        state_dict.pop(prefix + "num_batches_tracked", None)      # backbone.py:35-43
        if self._cache is not None:
            self._cache = (None,) + self._cache[1:]               # stale: folded() refreshes the two tensors in place
        super()._load_from_state_dict(state_dict, prefix, *args)

    def _apply(self, fn, *a, **k):
        self._cache = None
        return super()._apply(fn, *a, **k)

    def folded(self):
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.device)
        if self._cache is None or self._cache[0] != key:
            scale = self.weight * (self.running_var + 1e-5).rsqrt()
            shift = self.bias - self.running_mean * scale
            old = self._cache
            if old is not None and old[1].device == scale.device and old[1].shape == scale.shape:
                # refreshed IN PLACE: a captured HIP graph reads these two tensors by address (buffers reloaded after a capture)
                old[1].copy_(scale.float())
                old[2].copy_(shift.float())
                self._cache = (key, old[1], old[2])
            else:
                self._cache = (key, scale.float().contiguous(), shift.float().contiguous())
        return self._cache[1], self._cache[2]


GELU_GATE = True     # TEST HOOK (tests/test_hip_kernels.py compares the two forms): MLPs run GELU's backward in fc2's data-gradient epilogue


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (/root/reference/src/models/multiscale_transformerr.py:55-73, drop=0)."""

    def __init__(self, cin, hidden=None, cout=None):
        super().__init__()
        self.fc1 = Linear(cin, hidden or cin)
        self.fc2 = Linear(hidden or cin, cout or cin)

    def forward(self, x, residual=None):
        """fc2(gelu(fc1(x))) [+ residual, added in fc2's GEMM epilogue].  With gradients on, GELU's backward runs in the epilogue of fc2's
        data-gradient GEMM (from fc1's pre-activation tensor) instead of as a pass over the hidden map."""
        if GELU_GATE and torch.is_grad_enabled() and x.requires_grad:
            h, z = self.fc1(x, ops.ACT_GELU, defer=True)
            return self.fc2(h, residual=residual, in_gate=ops.ACT_GELU, gate_src=z)
        return self.fc2(self.fc1(x, ops.ACT_GELU), residual=residual)


class MlpNorm(nn.Module):
    """fc1 -> fc2 -> LayerNorm (/root/reference/src/models/multiscale_transformerr.py:75-102, no act)."""

    def __init__(self, c, hidden):
        super().__init__()
        self.fc1 = Linear(c, hidden)
        self.fc2 = Linear(hidden, c)
        self.norm = LayerNorm(c)

    def forward(self, x):
        return self.norm(self.fc2(self.fc1(x)))


class Seq(nn.Module):
    """Children registered under explicit string names ('0', '2', ...) to reproduce nn.Sequential keys."""

    def __init__(self, **named):
        super().__init__()
        for k, v in named.items():
            self.add_module(k.lstrip("_"), v)

    def __getitem__(self, i):
        return self._modules[str(i)]
