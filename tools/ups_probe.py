#!/usr/bin/env python
"""Stage-by-stage probe of the collapsed upsampled data gradient (each stage in its own process: tools/ups_probe.py N)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gw_depth_amd import hip, ops
stage = int(sys.argv[1])
torch.manual_seed(0)
B, Hi, Wi, Cin, Cout = (int(v) for v in (sys.argv[2:7] if len(sys.argv) > 6 else (2, 24, 32, 64, 64)))
w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05
if stage == 1:
    wk = ops._upsampled_dgrad_weight(w, torch.bfloat16)
    torch.cuda.synchronize()
    print("stage 1 ok", tuple(wk.shape), float(wk.float().abs().sum()))
    sys.exit(0)
A = torch.tensor([[0., 0., 1.], [0., 1., 1.], [1., 1., 0.], [1., 0., 0.]])
wk = torch.einsum("tk,sl,cklN->Ntsc", A, A, w.cpu().float()).to(torch.bfloat16).contiguous().cuda()
gy = torch.randn(B, 2 * Hi, 2 * Wi, Cout, device="cuda").bfloat16()
gx = torch.empty(B, Hi, Wi, Cin, device="cuda", dtype=torch.bfloat16)
lib = hip.library()
dims = (B, 2 * Hi, 2 * Wi, Cout, Hi, Wi, Cin, 4, 4)
if stage == 2:
    lib.conv_forward(gy, wk, gx, dims, stride=2, pad=1)
elif stage == 3:
    gate = torch.randn(B, Hi, Wi, Cin, device="cuda").bfloat16()
    lib.conv_forward(gy, wk, gx, dims, stride=2, pad=1, gate=gate, gate_act=hip.ACT_ELU)
torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(gy.float().permute(0, 3, 1, 2), wk.float().permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
if stage == 3:
    r = gate.float()
    ref = ref * torch.where(r > 0, torch.ones_like(r), r + 1)
print("stage", stage, "ok, rel err", float((gx.float() - ref).norm() / ref.norm()))
