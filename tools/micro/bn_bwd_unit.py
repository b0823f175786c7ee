"""Isolated BatchNorm+SiLU backward launches (cvx_bn_silu_bwd_nhwc) for rocprofv3 --kernel-trace: the one-launch gated kernel against the
two passes (tuning library: CVX_BN_FUSED=1 enables the gated kernel), no other stream at work.   python tools/micro/bn_bwd_unit.py [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from computervision.pytorch_amd import _lib as L  # noqa: E402

SHAPES = [(32, 20 * 20, 128), (32, 20 * 20, 256), (32, 40 * 40, 64), (32, 40 * 40, 128), (32, 80 * 80, 32), (32, 80 * 80, 64), (32, 160 * 160, 16)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
lib = L.load()
for B, hw, C in SHAPES:
    g = torch.Generator().manual_seed(0)
    xh = torch.randn(B * hw, C, generator=g).half().to(dev)
    go = torch.randn(B * hw, C, generator=g).half().to(dev)
    gamma, beta, invstd = [torch.rand(C, generator=g).add(0.5).to(dev) for _ in range(3)]
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.empty_like(xh)
    for _ in range(reps):
        L.check(lib.cvx_bn_silu_bwd_nhwc(L.ptr(xh), L.ptr(go), B, hw, C, L.ptr(gamma), L.ptr(beta), L.ptr(invstd), 1.0, L.ptr(dgamma), L.ptr(dbeta), L.ptr(dy),
                                         None, 0, None), "bn bwd")
    torch.cuda.synchronize()
    print(B, hw, C, float(dy.float().abs().mean()), float(dgamma.sum()))
