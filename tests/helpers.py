import json
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relative_position_index(ws=7):
    """The fixed (49,49) int64 buffer of /root/reference/src/models/multiscale_transformerr.py:236-247."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def reference_state_shapes():
    """A zero-filled state dict with the reference's 970 key names / shapes / dtypes (fixture:
    tests/golden/state_dict_spec.json, dumped from the reference's build_model by the harness)."""
    spec = json.load(open(os.path.join(GOLDEN, "state_dict_spec.json")))
    sd = {}
    for k, (shape, dtype) in spec.items():
        if k.endswith("relative_position_index"):
            sd[k] = relative_position_index()
        else:
            sd[k] = torch.zeros(shape, dtype=getattr(torch, dtype))
    return sd


def param_sets():
    return json.load(open(os.path.join(GOLDEN, "param_sets.json")))
