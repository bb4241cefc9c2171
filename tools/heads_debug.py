"""Stage-by-stage check of ocn_heads_fused against the torch modules (fp64): python tools/heads_debug.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ocn_amd.model as M  # noqa: E402
from ocn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def modules64(pred, x1, x2, xij):
    p = pred.double()
    with torch.no_grad():
        alpha = torch.sigmoid(p.alpha).cumprod(-1)
        z = alpha[0] * p.xcn1lin(x1.double()) + alpha[1] * p.xcn2lin(x2.double()) + p.beta * p.xijlin(xij.double())
        out = p.lin(z)
    pred.float()
    return out


def run(H, B, ln, tailact, mode, seed=0):
    torch.manual_seed(seed)
    pred = M.predictor_dict["cn5"](H, H, 1, 3, 0.0, 0.0, ln, tailact=tailact).to(dev).eval()
    with torch.no_grad():
        for p in pred.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        if mode == "xij_only":
            pred.xcn1lin[7].weight.zero_(); pred.xcn1lin[7].bias.zero_(); pred.xcn2lin[7].weight.zero_(); pred.xcn2lin[7].bias.zero_()
        if mode == "a_only":
            pred.xcn2lin[7].weight.zero_(); pred.xcn2lin[7].bias.zero_(); pred.beta.zero_()
        if mode == "b_only":
            pred.xcn1lin[7].weight.zero_(); pred.xcn1lin[7].bias.zero_(); pred.beta.zero_()
    x1, x2, xij = (torch.randn(B, H, device=dev) for _ in range(3))
    ref = modules64(pred, x1, x2, xij)
    pred._drop_caches()
    with torch.no_grad():
        got = pred._heads_fused(x1, x2, xij, None)
    torch.cuda.synchronize()
    err = (got.double() - ref).abs().max().item()
    print(f"H={H:3d} B={B:5d} ln={int(ln)} tailact={int(tailact)} {mode:9s} max|err| = {err:.3e}  (|ref| max {ref.abs().max().item():.3f})", flush=True)


for H in (32, 64, 256):
    for mode in ("xij_only", "a_only", "b_only", "all"):
        run(H, 37, False, True, mode)
    run(H, 37, True, True, "all")
    run(H, 37, True, False, "all")
    run(H, 300, True, False, "all")
