"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle and the golden fixtures.

Tolerances (stated once, used below):
* integer / index work (NMS kept indices, rows, task-aligned assignment): bit-exact;
* fp32 kernels given identical inputs (loss values, weight-gradient GEMM, fp32 stem): 1e-5 relative;
* fp16-storage kernels given identical inputs: 5e-4 relative L2 (one fp16 rounding of the output);
* whole-network forward vs the reference's FP32 CPU outputs (fixtures captured from the reference itself):
  **<= 1e-3 relative L2 per Detect level** -- the bar BASELINE.json's north_star states ("CPU-reference parity within 1e-3
  rel on logits").  The engine stores MFMA weights and activations in fp16 like the reference's CUDA autocast path; the stem
  runs in fp32 and every BatchNorm normalises the fp32 accumulators, which is what it took to get below the bar (DESIGN.md
  section 2 has the ablation).  ``LEVEL_TOL`` below; the measured numbers are printed by the tests (pytest -s);
* parameter gradients: backward kernels fed the oracle's d(loss)/d(pred): <= 6e-2 global relative L2 vs fp32 (the
  fp16-storage emulation of the oracle itself deviates by ~4e-2); fully end to end (own loss, discrete assignment): <= 1.2e-1;
  per-layer activation gradients vs the fp16-storage emulation: see test_per_layer_backward_parity.
"""
import contextlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from computervision.pytorch_amd import _lib as L  # noqa: E402
from computervision.pytorch_amd import engine as E  # noqa: E402
from oracle import nms_ref, synth  # noqa: E402
from oracle import yolov8_ref as O  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


LEVEL_TOL = 1e-3     # north_star: logits within 1e-3 rel of the reference's CPU path, per Detect level


def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def level_report(tag, outs, refs):
    """rel-L2 per level (whole (B, 64+nc, H, W) tensor, and its box / class halves separately) -- printed, returned."""
    r = [rel(o, t) for o, t in zip(outs, refs)]
    rb = [rel(o[:, :64], t[:, :64]) for o, t in zip(outs, refs)]
    rc = [rel(o[:, 64:], t[:, 64:]) for o, t in zip(outs, refs)]
    print(f"[parity] {tag}: level rel-L2 " + " ".join(f"{v:.2e}" for v in r) + " | box " + " ".join(f"{v:.2e}" for v in rb) +
          " | cls " + " ".join(f"{v:.2e}" for v in rc))
    return r


def new_model(dev, seed=0, scale="n"):
    from computervision.pytorch_amd.model import Yolo8
    torch.manual_seed(seed)
    return Yolo8(scale, 80).to(dev)


class fp16_storage:
    def __enter__(self):
        O.FP16_STORAGE[0] = True

    def __exit__(self, *a):
        O.FP16_STORAGE[0] = False


# ---- single kernels ------------------------------------------------------------------------------------
CONV_CASES = [(2, 16, 16, 16, 16, 3, 1), (2, 20, 20, 8, 16, 3, 2), (1, 24, 24, 32, 64, 3, 2), (2, 10, 10, 64, 144, 3, 1),
              (2, 12, 12, 48, 32, 1, 1), (1, 20, 20, 384, 256, 1, 1), (3, 7, 9, 80, 80, 3, 1), (1, 13, 13, 256, 512, 3, 2),
              (1, 5, 5, 16, 16, 3, 1), (2, 33, 17, 96, 64, 1, 1), (2, 40, 40, 144, 64, 3, 1), (1, 80, 80, 80, 80, 3, 1),
              (1, 80, 80, 128, 32, 3, 1), (4, 20, 20, 128, 128, 3, 1), (1, 160, 160, 32, 32, 1, 1), (2, 20, 20, 512, 256, 1, 1),
              # big-channel shapes (VGG 19x19 512->512, 38x38 256->512, ResNet-101 1x1 1024->256, a stride-2 3x3, ragged tiles)
              (8, 19, 19, 512, 512, 3, 1), (2, 38, 38, 256, 512, 3, 1), (4, 33, 33, 1024, 256, 1, 1), (8, 40, 40, 128, 256, 3, 2),
              (3, 27, 31, 160, 200, 3, 1)]
GEMM_CASES = [(8, 19, 19, 512, 512, 3, 1, 1), (2, 38, 38, 256, 512, 3, 1, 1), (4, 33, 33, 1024, 256, 1, 1, 1), (8, 40, 40, 128, 256, 3, 2, 1),
              (3, 27, 31, 160, 200, 3, 1, 1), (2, 33, 33, 256, 256, 3, 1, 2), (1, 9, 9, 32, 68, 3, 1, 1), (40, 38, 38, 512, 512, 3, 1, 1),
              (2, 33, 33, 304, 256, 3, 1, 1), (3, 20, 20, 80, 144, 3, 1, 1), (2, 24, 24, 8, 128, 1, 1, 1),  # Cin % 32 != 0: ragged last chunk of every tap
              (2, 19, 19, 512, 24, 3, 1, 1), (1, 38, 38, 256, 16, 3, 1, 1), (2, 10, 10, 512, 88, 3, 1, 1)]  # SSD's loc / conf heads: deep K, few outputs


@pytest.mark.parametrize("B,H,W,C,f", [(2, 12, 10, 64, 2), (1, 7, 9, 24, 2), (2, 6, 5, 64, 4), (1, 5, 4, 16, 8), (1, 3, 3, 520, 2), (3, 24, 24, 256, 2)])
def test_depthwise_transposed_conv_fwd_bwd(dev, B, H, W, C, f):
    """cvx_dwconvt_nhwc / cvx_dwconvt_bwd_nhwc (IDAUp.up_i: ConvTranspose2d(c, c, 2f, stride=f, padding=f//2, groups=c, bias=False),
    centernet_model.py:256) against torch fp32 on fp16-valued operands: forward one fp16 rounding, data gradient likewise (plain and
    accumulate), weight gradient 1e-5 (accumulated into a non-zero arena with inv_scale).  f = 2 takes the one-pass backward (weight +
    data gradient together), f = 4 / 8 and C > 512 the tap-group kernel with the separate data gradient; C = 24 has 3 channel groups
    (256 threads do not divide evenly)."""
    lib = L.load()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(B * 100 + H * 10 + C + f)
    x16 = torch.randn(B, C, H, W, generator=g).half()
    wt = torch.randn(C, 1, 2 * f, 2 * f, generator=g) * 0.3
    xr, wr = x16.float().requires_grad_(True), wt.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, None, stride=f, padding=f // 2, groups=C)
    assert tuple(ref.shape) == (B, C, H * f, W * f)
    go16 = torch.randn(B, C, H * f, W * f, generator=g).half()
    ref.backward(go16.float())
    xd, wd = _nhwc(x16).to(dev), wt.reshape(C, 2 * f, 2 * f).contiguous().to(dev)
    out = torch.empty(B, H * f, W * f, C, dtype=torch.float16, device=dev)
    L.check(lib.cvx_dwconvt_nhwc(L.ptr(xd), B, H, W, C, f, L.ptr(wd), L.ptr(out), st), "dwconvt")
    assert (out.float().permute(0, 3, 1, 2).cpu() - ref.detach()).abs().max() <= 1e-3 * max(1.0, float(ref.abs().max()))
    base16 = torch.randn(B, C, H, W, generator=g).half()
    dw0 = torch.randn(C, 2 * f, 2 * f, generator=g)
    inv_scale = 0.25
    for acc in (0, 1):
        gin = _nhwc(base16).to(dev).clone()
        dw = dw0.to(dev).clone()
        L.check(lib.cvx_dwconvt_bwd_nhwc(L.ptr(xd), L.ptr(_nhwc(go16).to(dev)), B, H, W, C, f, L.ptr(wd), L.ptr(gin), acc, L.ptr(dw), inv_scale, st),
                "dwconvt bwd")
        want = xr.grad + (base16.float() if acc else 0)
        assert (gin.float().permute(0, 3, 1, 2).cpu() - want).abs().max() <= 2e-3 * max(1.0, float(want.abs().max())), acc
        want_dw = dw0 + inv_scale * wr.grad.reshape(C, 2 * f, 2 * f)
        assert rel(dw.cpu(), want_dw) < 1e-5, (acc, rel(dw.cpu(), want_dw))


@pytest.mark.parametrize("B,H,W,Ci,Co,k,s", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, B, H, W, Ci, Co, k, s):
    lib = L.load()
    g = torch.Generator().manual_seed(B * 1000 + H + Ci + Co)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).half()
    xr, wr = x16.float().requires_grad_(True), w16.float().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, s, k // 2)
    OH, OW = ref.shape[2:]
    dy16 = torch.randn(B, Co, OH, OW, generator=g).half()
    ref.backward(dy16.float())
    st = L.stream_ptr(dev)
    xd = x16.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w16.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty(B, OH, OW, Co, dtype=torch.float16, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, s, k // 2, 1, 0, None, None, L.ptr(out), st), "conv")
    assert rel(out.float().permute(0, 3, 1, 2), ref.detach()) < 5e-4
    dyd = dy16.permute(0, 2, 3, 1).contiguous().to(dev)
    wtd = w16.permute(1, 2, 3, 0).contiguous().to(dev)
    dx = torch.zeros(B, H, W, Ci, dtype=torch.float16, device=dev)
    L.check(lib.cvx_conv2d_dgrad_nhwc(L.ptr(dyd), B, H, W, Ci, L.ptr(wtd), Co, k, s, k // 2, 1, L.ptr(dx), st), "dgrad")
    assert rel(dx.float().permute(0, 3, 1, 2), xr.grad) < 5e-4
    need = lib.cvx_conv2d_wgrad_workspace_bytes(B, OH, OW, Ci, Co, k)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device=dev)
    L.check(lib.cvx_conv2d_wgrad_nhwc(L.ptr(xd), L.ptr(dyd), B, H, W, Ci, Co, k, s, k // 2, 1, L.ptr(dw), L.ptr(ws), need, st), "wgrad")
    assert rel(dw.permute(0, 3, 1, 2), wr.grad) < 1e-5


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 20, 20, 64, 36), (1, 40, 40, 32, 68), (3, 9, 13, 144, 36)])
def test_default_dispatch_with_output_channels_that_are_multiples_of_4_only(dev, B, H, W, Ci, Co):
    """cvx_conv2d_nhwc / cvx_conv2d_dgrad_nhwc accept Cout % 4 == 0; the row-band kernel's staged epilogue works on 8 channels per lane and
    must not be dispatched to for 36 / 68 output channels on the small maps it otherwise takes (it read scale / shift past their end and
    wrote 4 channels into the next pixel: ADVICE r04) -- forward with folded BN + SiLU and the data gradient against CPU fp32."""
    lib = L.load()
    g = torch.Generator().manual_seed(B * 100 + Co)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).half()
    scale, shift = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    xr, wr = x16.float().requires_grad_(True), w16.float()
    ref = F.conv2d(xr, wr, None, 1, 1)
    want = F.silu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    st = L.stream_ptr(dev)
    xd, wd = _nhwc(x16).to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
    sc, sh = scale.to(dev), shift.to(dev)
    guard = 64                                                                         # elements behind the output that must stay untouched
    out = torch.full((B * H * W * Co + guard,), 7.0, dtype=torch.float16, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, 1, L.ptr(sc), L.ptr(sh), L.ptr(out), st), "conv")
    assert torch.all(out[-guard:] == 7.0), "the kernel wrote past the end of the output"
    got = out[:-guard].reshape(B, H, W, Co).float().permute(0, 3, 1, 2).cpu()
    assert rel(got, want.detach()) < 1e-3
    # the data gradient's launch has the forward's INPUT channels as its outputs: a 36- / 68-channel input (dy itself has Ci, a multiple of 8)
    g2 = torch.Generator().manual_seed(B * 100 + Co + 1)
    w2 = (torch.randn(Ci, Co, 3, 3, generator=g2) / (Co * 9) ** 0.5).half()        # conv Co -> Ci
    x2 = torch.zeros(B, Co, H, W, requires_grad=True)
    dy2 = torch.randn(B, Ci, H, W, generator=g2).half()
    F.conv2d(x2, w2.float(), None, 1, 1).backward(dy2.float())
    dyd, wtd = _nhwc(dy2).to(dev), w2.permute(1, 2, 3, 0).contiguous().to(dev)
    dx = torch.full((B * H * W * Co + guard,), 7.0, dtype=torch.float16, device=dev)
    L.check(lib.cvx_conv2d_dgrad_nhwc(L.ptr(dyd), B, H, W, Co, L.ptr(wtd), Ci, 3, 1, 1, 1, L.ptr(dx), st), "dgrad")
    assert torch.all(dx[-guard:] == 7.0), "the data-gradient kernel wrote past the end of its output"
    assert rel(dx[:-guard].reshape(B, H, W, Co).float().permute(0, 3, 1, 2), x2.grad) < 5e-4


# round 5: the fat-workgroup 3x3 kernel (conv_wgrad_k3.hip: every configuration A..F, column blocks, ragged last row block / column block, chunk
# ranges that end inside an image, channel padding 8 -> 16 and 72 -> 80, co blocks with an empty tail) and the streaming 1x1 kernel
# (conv_wgrad_stream.hip: K-step waves, odd ci-tile counts, ragged last chunk) -- beyond what CONV_CASES holds
WGRAD_R05_CASES = [(2, 13, 160, 16, 16, 3), (3, 11, 80, 32, 32, 3), (2, 40, 40, 64, 64, 3), (5, 20, 20, 64, 64, 3), (2, 9, 37, 64, 64, 3), (1, 80, 80, 64, 144, 3),
                   (2, 21, 40, 128, 144, 3), (3, 20, 20, 128, 128, 3), (2, 23, 19, 80, 80, 3), (2, 20, 20, 256, 144, 3), (3, 6, 250, 16, 16, 3),
                   (2, 17, 29, 64, 32, 3), (2, 12, 12, 8, 16, 3), (1, 7, 7, 72, 80, 3), (2, 30, 30, 32, 24, 3),
                   (40, 160, 160, 32, 32, 1), (36, 160, 160, 48, 32, 1), (34, 157, 163, 16, 32, 1), (33, 160, 160, 64, 16, 1)]


@pytest.mark.parametrize("B,H,W,Ci,Co,k", WGRAD_R05_CASES)
def test_weight_gradient_kernels_of_round_5(dev, B, H, W, Ci, Co, k):
    """cvx_conv2d_wgrad_nhwc on the shapes the two round-5 kernels take, against fp32 F.conv2d backward on the CPU (1x1 cases: a matmul
    in fp64 -- they are 800k+ pixels because the streaming kernel only takes such layers)."""
    lib = L.load()
    g = torch.Generator().manual_seed(B * 1000 + H + Ci + Co + k)
    x16 = (torch.randn(B, H, W, Ci, generator=g) * 0.5).half()
    dy16 = (torch.randn(B, H, W, Co, generator=g) * 0.5).half()
    if k == 1:
        want = (dy16.reshape(-1, Co).double().t() @ x16.reshape(-1, Ci).double()).float().reshape(Co, 1, 1, Ci)
    else:
        xr = x16.permute(0, 3, 1, 2).float()
        wr = torch.zeros(Co, Ci, k, k, requires_grad=True)
        F.conv2d(xr, wr, None, 1, k // 2).backward(dy16.permute(0, 3, 1, 2).float())
        want = wr.grad.permute(0, 2, 3, 1)
    st = L.stream_ptr(dev)
    need = lib.cvx_conv2d_wgrad_workspace_bytes(B, H, W, Ci, Co, k)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device=dev)
    xd, dyd = x16.to(dev), dy16.to(dev)
    L.check(lib.cvx_conv2d_wgrad_nhwc(L.ptr(xd), L.ptr(dyd), B, H, W, Ci, Co, k, 1, k // 2, 1, L.ptr(dw), L.ptr(ws), need, st), "wgrad")
    assert rel(dw.cpu(), want) < 1e-5, rel(dw.cpu(), want)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 64, 64, 16, 32), (3, 36, 320, 16, 32), (2, 50, 38, 16, 32), (1, 25, 31, 16, 32), (2, 80, 80, 32, 64), (3, 22, 160, 32, 64),
                                         (2, 33, 47, 32, 64), (5, 8, 8, 16, 32), (1, 6, 250, 32, 64)])
def test_stride2_weight_gradient_in_phase_planes(dev, B, H, W, Ci, Co):
    """conv_wgrad_k3.hip on 3x3 / stride 2 / pad 1 (model.1: 16 -> 32, model.3: 32 -> 64): the input's four phase planes side by side in LDS,
    every tap a constant offset inside one plane.  Even and odd maps (the last input row / column is then read by kh, kw = 1 only), maps
    wide enough for several column blocks, maps smaller than one chunk; against fp32 F.conv2d backward on the CPU."""
    lib = L.load()
    g = torch.Generator().manual_seed(B * 1000 + H + W + Ci)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x16 = (torch.randn(B, H, W, Ci, generator=g) * 0.5).half()
    dy16 = (torch.randn(B, OH, OW, Co, generator=g) * 0.5).half()
    wr = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x16.permute(0, 3, 1, 2).float(), wr, None, 2, 1).backward(dy16.permute(0, 3, 1, 2).float())
    want = wr.grad.permute(0, 2, 3, 1)
    st = L.stream_ptr(dev)
    need = lib.cvx_conv2d_wgrad_workspace_bytes(B, OH, OW, Ci, Co, 3)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    dw = torch.empty(Co, 3, 3, Ci, dtype=torch.float32, device=dev)
    xd, dyd = x16.to(dev), dy16.to(dev)
    L.check(lib.cvx_conv2d_wgrad_nhwc(L.ptr(xd), L.ptr(dyd), B, H, W, Ci, Co, 3, 2, 1, 1, L.ptr(dw), L.ptr(ws), need, st), "wgrad stride 2")
    assert rel(dw.cpu(), want) < 1e-5, rel(dw.cpu(), want)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 16, 24, 32, 64), (3, 20, 20, 64, 128), (1, 40, 8, 16, 32), (2, 12, 12, 128, 256), (2, 10, 14, 24, 96)])
def test_stride2_data_gradient_as_one_pixel_shuffle_gemm(dev, B, H, W, Ci, Co):
    """cvx_conv2d_dgrad_nhwc takes the engine's route for 3x3 / stride 2 / pad 1 on even maps: the 2 x 2 window of dy as a stride-1 GEMM with
    the four output phases as channel blocks and a pixel-shuffle store (ConvParams::ps_cin) -- against torch's conv2d backward on the CPU,
    incl. 16 and 24 input channels (4 x 16 = 64 outputs: the smallest the route takes) and maps whose last row / column of dy has no
    neighbour below / right (every map: the window's second tap is out of range there)."""
    lib = L.load()
    g = torch.Generator().manual_seed(H * 100 + Ci)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).half()
    xr, wr = x16.float().requires_grad_(True), w16.float()
    ref = F.conv2d(xr, wr, None, 2, 1)
    dy16 = torch.randn(*ref.shape, generator=g).half()
    ref.backward(dy16.float())
    dyd = dy16.permute(0, 2, 3, 1).contiguous().to(dev)
    wtd = w16.permute(1, 2, 3, 0).contiguous().to(dev)
    dx = torch.full((B, H, W, Ci), 7.0, dtype=torch.float16, device=dev)     # every pixel must be written
    L.check(lib.cvx_conv2d_dgrad_nhwc(L.ptr(dyd), B, H, W, Ci, L.ptr(wtd), Co, 3, 2, 1, 1, L.ptr(dx), L.stream_ptr(dev)), "dgrad")
    assert rel(dx.float().permute(0, 3, 1, 2), xr.grad) < 5e-4


@pytest.mark.parametrize("B,H,W,Ci,Co,k,s,dil", GEMM_CASES)
def test_gemm_shaped_conv_kernel_all_epilogues(dev, B, H, W, Ci, Co, k, s, dil):
    """conv_gemm.hip run directly (mode | 0x100) -- the dispatcher only prefers it on the largest layers: plain fp16 store, folded BN + SiLU,
    bias -> fp32, and the training epilogue (raw fp32 + per-channel statistics), on VGG / ResNet-101 / atrous shapes, ragged pixel and
    channel tiles (9x9 pixels, 68 / 200 channels), the batch-40 38x38 512->512 layer that takes the 256 x 256 macro tile, and input
    channel counts that are not multiples of the 32-deep ring chunk (304: DeepLab's decoder concat; 80; 8)."""
    lib = L.load()
    g = torch.Generator().manual_seed(B + H + Ci + Co + k)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).half()
    pad = dil * (k // 2)
    xd, wd = x16.permute(0, 2, 3, 1).contiguous().to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
    # the reference convolution: torch fp32 on the CPU for the small cases (every epilogue is then checked against a CPU result at least
    # once per parametrisation set), torch-ROCm's own fp32 convolution for the large ones (a CPU conv of the 512-channel cases takes seconds each)
    on_cpu = B * H * W * Ci * Co * k * k <= 2.5e9
    conv = F.conv2d(x16.float(), w16.float(), None, s, pad, dil).to(dev) if on_cpu else F.conv2d(x16.float().to(dev), w16.float().to(dev), None, s, pad, dil)
    OH, OW = conv.shape[2:]
    st = L.stream_ptr(dev)
    sc, sh = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
    out = torch.empty(B, OH, OW, Co, dtype=torch.float16, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, s, pad, dil, 0x100, None, None, L.ptr(out), st), "plain")
    assert rel(out.float().permute(0, 3, 1, 2), conv) < 5e-4
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, s, pad, dil, 0x101, L.ptr(sc), L.ptr(sh), L.ptr(out), st), "affine")
    assert rel(out.float().permute(0, 3, 1, 2), F.silu(conv * sc[None, :, None, None] + sh[None, :, None, None])) < 5e-4
    out32 = torch.empty(B, OH, OW, Co, dtype=torch.float32, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, s, pad, dil, 0x102, L.ptr(sh), None, L.ptr(out32), st), "bias")
    assert rel(out32.permute(0, 3, 1, 2), conv + sh[None, :, None, None]) < 1e-5
    R = 16 if Co <= 32 else 8 if Co <= 64 else 4 if Co <= 128 else 2 if Co <= 256 else 1  # cvx_stat_replicas (csrc/bn_act.h:8)
    slab = torch.zeros(R, Co, 2, 2, dtype=torch.int64, device=dev)  # [replica][channel][sum, sum of squares][coarse 2^-6, fine 2^-40]
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, s, pad, dil, 0x103, None, L.ptr(slab), L.ptr(out32), st), "stats")
    assert rel(out32.permute(0, 3, 1, 2), conv) < 1e-5
    tot = slab.sum(0).double()
    sums = tot[..., 0] / 64.0 + tot[..., 1] / 2.0 ** 40
    n = B * OH * OW
    assert (sums[:, 0] / n - conv.double().mean((0, 2, 3))).abs().max() < 1e-5
    assert rel(sums[:, 1] / n, (conv.double() ** 2).mean((0, 2, 3))) < 1e-5


TILE_CASES = [  # B, H, W, Cin, Cout: 3x3 stride 1 pad 1
    (2, 40, 40, 64, 64), (3, 20, 20, 128, 128), (2, 80, 80, 32, 32), (1, 40, 40, 128, 144), (2, 20, 20, 256, 144), (2, 40, 40, 80, 80),
    (1, 20, 20, 144, 256), (2, 13, 17, 32, 40), (1, 7, 5, 96, 24), (3, 33, 9, 48, 200), (1, 80, 80, 64, 144), (2, 2, 2, 64, 16), (1, 3, 50, 40, 8), (32, 20, 20, 128, 128), (32, 40, 40, 64, 64), (16, 40, 40, 128, 144),
]


@pytest.mark.parametrize("B,H,W,Ci,Co", TILE_CASES)
def test_row_band_conv_kernel_all_epilogues(dev, B, H, W, Ci, Co):
    """conv_tile.hip run directly (mode | 0x2000): plain fp16 store (+ accumulate is covered by the engine's data-gradient tests), folded
    BN + SiLU, bias -> fp32, and the training epilogue (raw fp32 + per-channel statistics) on YOLOv8-n's small-map shapes, ragged row bands
    (13 x 17, 33 x 9), maps smaller than one pixel group, channel counts that are not powers of two (80, 144, 96, 48, 40: no swizzle, padded
    K-steps) and output channel counts that leave the last channel block ragged (200, 24, 8) -- against torch's fp32 convolution ON THE
    CPU of the same fp16-valued operands (modules.py:19-33)."""
    lib = L.load()
    g = torch.Generator().manual_seed(B + H * 3 + W + Ci + Co)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).half()
    xd, wd = x16.permute(0, 2, 3, 1).contiguous().to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
    conv = F.conv2d(x16.float(), w16.float(), None, 1, 1)  # CPU fp32
    st = L.stream_ptr(dev)
    sc, sh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    scd, shd = sc.to(dev), sh.to(dev)
    out = torch.empty(B, H, W, Co, dtype=torch.float16, device=dev)
    T = 0x2000
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, T | 0, None, None, L.ptr(out), st), "plain")
    assert rel(out.float().permute(0, 3, 1, 2).cpu(), conv) < 5e-4
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, T | 1, L.ptr(scd), L.ptr(shd), L.ptr(out), st), "affine")
    assert rel(out.float().permute(0, 3, 1, 2).cpu(), F.silu(conv * sc[None, :, None, None] + sh[None, :, None, None])) < 5e-4
    out32 = torch.empty(B, H, W, Co, dtype=torch.float32, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, T | 2, L.ptr(shd), None, L.ptr(out32), st), "bias")
    assert rel(out32.permute(0, 3, 1, 2).cpu(), conv + sh[None, :, None, None]) < 1e-5
    R = 16 if Co <= 32 else 8 if Co <= 64 else 4 if Co <= 128 else 2 if Co <= 256 else 1  # cvx_stat_replicas (csrc/bn_act.h:8)
    slab = torch.zeros(R, Co, 2, 2, dtype=torch.int64, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, T | 3, None, L.ptr(slab), L.ptr(out32), st), "stats")
    assert rel(out32.permute(0, 3, 1, 2).cpu(), conv) < 1e-5
    tot = slab.sum(0).double().cpu()
    sums = tot[..., 0] / 64.0 + tot[..., 1] / 2.0 ** 40
    n = B * H * W
    assert (sums[:, 0] / n - conv.double().mean((0, 2, 3))).abs().max() < 1e-5
    assert rel(sums[:, 1] / n, (conv.double() ** 2).mean((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7])
def test_gemm_shaped_conv_kernel_every_variant(dev, variant):
    """Each variant of conv_gemm.hip (macro tile 256x256 ... 128x128, 3 / 4 ring slots, 32- / 64-deep chunks: kVariants) forced through
    mode bits 9..12, whatever the cost model would pick: folded BN + SiLU output on a VGG shape, ragged pixel / channel tiles, a dilated
    conv, and input channel counts that leave the last chunk of a tap ragged at both chunk depths (304, 160, 8)."""
    lib = L.load()
    st = L.stream_ptr(dev)
    for (B, H, W, Ci, Co, k, s, dil) in [(8, 19, 19, 512, 512, 3, 1, 1), (3, 27, 31, 160, 200, 3, 1, 1), (2, 33, 33, 304, 256, 3, 1, 2), (2, 24, 24, 8, 128, 1, 1, 1),
                                         (4, 40, 40, 128, 256, 3, 2, 1), (1, 33, 33, 96, 132, 1, 1, 1)]:
        g = torch.Generator().manual_seed(variant * 100 + Ci + Co)
        x16 = torch.randn(B, Ci, H, W, generator=g).half()
        w16 = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).half()
        pad = dil * (k // 2)
        xd, wd = x16.permute(0, 2, 3, 1).contiguous().to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
        conv = F.conv2d(x16.float().to(dev), w16.float().to(dev), None, s, pad, dil)
        OH, OW = conv.shape[2:]
        sc, sh = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
        out = torch.empty(B, OH, OW, Co, dtype=torch.float16, device=dev)
        L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, s, pad, dil, 0x101 | (variant << 9), L.ptr(sc), L.ptr(sh), L.ptr(out), st), "affine")
        assert rel(out.float().permute(0, 3, 1, 2), F.silu(conv * sc[None, :, None, None] + sh[None, :, None, None])) < 5e-4, (variant, Ci, Co)
        out32 = torch.empty(B, OH, OW, Co, dtype=torch.float32, device=dev)
        R = 16 if Co <= 32 else 8 if Co <= 64 else 4 if Co <= 128 else 2 if Co <= 256 else 1
        slab = torch.zeros(R, Co, 2, 2, dtype=torch.int64, device=dev)
        L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, k, s, pad, dil, 0x103 | (variant << 9), None, L.ptr(slab), L.ptr(out32), st), "stats")
        assert rel(out32.permute(0, 3, 1, 2), conv) < 1e-5, (variant, Ci, Co)
        tot = slab.sum(0).double()
        sums = tot[..., 0] / 64.0 + tot[..., 1] / 2.0 ** 40
        assert (sums[:, 0] / (B * OH * OW) - conv.double().mean((0, 2, 3))).abs().max() < 1e-5, (variant, Ci, Co)


@pytest.mark.parametrize("B,H,W,Co,s,k", [(2, 40, 72, 16, 1, 7), (2, 37, 53, 64, 2, 7), (1, 64, 200, 16, 2, 7), (3, 19, 130, 64, 1, 7), (1, 8, 8, 16, 1, 7),
                                          (2, 30, 70, 32, 1, 3), (1, 75, 129, 32, 1, 3)])
def test_first_layer_kernel(dev, B, H, W, Co, s, k):
    """conv_stem7.hip (DLA-34's base layer: 7x7 stride 1, 16 channels; ResNet's conv1: 7x7 stride 2, 64 channels; YOLOv7's first layer: 3x3,
    32 channels -- on the image padded to 8 channels): the dispatcher takes these launches there.  Folded BN + SiLU output and the training epilogue (raw fp32 + per-channel
    statistics) against torch fp32; ragged tiles in both directions, an image smaller than one tile."""
    lib = L.load()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(B * 7 + H + W + Co + s)
    x16 = torch.zeros(B, 8, H, W, dtype=torch.float16)
    x16[:, :3] = torch.rand(B, 3, H, W, generator=g).half()
    w16 = torch.zeros(Co, 8, k, k, dtype=torch.float16)
    w16[:, :3] = (torch.randn(Co, 3, k, k, generator=g) / (3 * k * k) ** 0.5).half()
    xd, wd = x16.permute(0, 2, 3, 1).contiguous().to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
    conv = F.conv2d(x16.float().to(dev), w16.float().to(dev), None, s, k // 2)
    OH, OW = conv.shape[2:]
    sc, sh = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
    out = torch.empty(B, OH, OW, Co, dtype=torch.float16, device=dev)
    wide = 15 << 9  # mode bits 9..12 = 15: also the 64-channel variant, which the dispatcher leaves to the GEMM-shaped kernel (measured)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, 8, L.ptr(wd), Co, k, s, k // 2, 1, 1 | wide, L.ptr(sc), L.ptr(sh), L.ptr(out), st), "affine")
    assert rel(out.float().permute(0, 3, 1, 2), F.silu(conv * sc[None, :, None, None] + sh[None, :, None, None])) < 5e-4
    out32 = torch.empty(B, OH, OW, Co, dtype=torch.float32, device=dev)
    R = 16 if Co <= 32 else 8
    slab = torch.zeros(R, Co, 2, 2, dtype=torch.int64, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, 8, L.ptr(wd), Co, k, s, k // 2, 1, 3 | wide, None, L.ptr(slab), L.ptr(out32), st), "stats")
    assert rel(out32.permute(0, 3, 1, 2), conv) < 1e-5
    tot = slab.sum(0).double()
    sums = tot[..., 0] / 64.0 + tot[..., 1] / 2.0 ** 40
    n = B * OH * OW
    assert (sums[:, 0] / n - conv.double().mean((0, 2, 3))).abs().max() < 1e-5
    assert rel(sums[:, 1] / n, (conv.double() ** 2).mean((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("B,H,W,Ci,Co,k,s", [(2, 8, 8, 256, 128, 3, 1), (3, 4, 9, 256, 136, 1, 1), (1, 12, 40, 64, 256, 3, 2), (5, 9, 11, 48, 512, 3, 1),
                                             (2, 16, 24, 264, 128, 1, 1), (1, 64, 8, 160, 128, 3, 1)])
def test_gemm_shaped_weight_gradient_edges(dev, B, H, W, Ci, Co, k, s):
    """conv_wgrad_gemm.hip at the edges of what it takes: output maps 8 pixels wide (four row wraps inside one 32-pixel chunk) and 4 rows
    high (an image wrap per chunk), pixel counts that are not multiples of the chunk, 136 / 264 channels (ragged channel and column tiles,
    cin_pad16 > Cin), stride 2, and both tiles (128 x 128, 256 x 256) -- against torch fp32 on fp16-valued operands."""
    lib = L.load()
    g = torch.Generator().manual_seed(B * 31 + H + W + Ci + Co)
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w = torch.zeros(Co, Ci, k, k, requires_grad=True)
    ref = F.conv2d(x16.float(), w, None, s, k // 2)
    OH, OW = ref.shape[2:]
    dy16 = torch.randn(B, Co, OH, OW, generator=g).half()
    ref.backward(dy16.float())
    st = L.stream_ptr(dev)
    xd, dyd = x16.permute(0, 2, 3, 1).contiguous().to(dev), dy16.permute(0, 2, 3, 1).contiguous().to(dev)
    need = lib.cvx_conv2d_wgrad_workspace_bytes(B, OH, OW, Ci, Co, k)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device=dev)
    L.check(lib.cvx_conv2d_wgrad_nhwc(L.ptr(xd), L.ptr(dyd), B, H, W, Ci, Co, k, s, k // 2, 1, L.ptr(dw), L.ptr(ws), need, st), "wgrad")
    assert rel(dw.permute(0, 3, 1, 2), w.grad) < 1e-5


def test_conv_epilogues_affine_silu_and_bias(dev):
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    B, H, W, Ci, Co = 2, 9, 11, 32, 80
    x16 = torch.randn(B, Ci, H, W, generator=g).half()
    w16 = (torch.randn(Co, Ci, 3, 3, generator=g) / 17).half()
    sc, sh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    conv = F.conv2d(x16.float(), w16.float(), None, 1, 1)
    xd, wd = x16.permute(0, 2, 3, 1).contiguous().to(dev), w16.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty(B, H, W, Co, dtype=torch.float16, device=dev)
    scd, shd = sc.to(dev), sh.to(dev)               # keep the device copies alive across the call
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, 1, L.ptr(scd), L.ptr(shd), L.ptr(out),
                                L.stream_ptr(dev)), "conv")
    assert rel(out.float().permute(0, 3, 1, 2), F.silu(conv * sc[None, :, None, None] + sh[None, :, None, None])) < 5e-4
    out32 = torch.empty(B, H, W, Co, dtype=torch.float32, device=dev)
    L.check(lib.cvx_conv2d_nhwc(L.ptr(xd), B, H, W, Ci, L.ptr(wd), Co, 3, 1, 1, 1, 2, L.ptr(shd), None, L.ptr(out32),
                                L.stream_ptr(dev)), "conv")
    assert rel(out32.permute(0, 3, 1, 2), conv + sh[None, :, None, None]) < 1e-5


# ---- streaming kernels, one by one, vs plain torch fp32 --------------------------------------------------------------
def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,C,res", [(2, 16, 16, 16, False), (3, 9, 7, 80, True), (1, 40, 40, 144, False), (2, 5, 5, 256, True),
                                          (4, 20, 20, 64, True), (1, 3, 3, 640, False)])
def test_bn_silu_train_and_backward_unit(dev, B, H, W, C, res):
    """bn_silu_apply / bn_bwd_reduce / bn_bwd_apply (bn_act.hip) against torch autograd of
    silu(batch_norm(y)) (+ residual) in fp32 (modules.py:29-30, 134-135)."""
    lib = L.load()
    g = torch.Generator().manual_seed(B * 100 + C)
    y = (torch.randn(B, C, H, W, generator=g) * (torch.rand(C, generator=g) * 3 + 0.2).view(1, C, 1, 1) + torch.randn(C, generator=g).view(1, C, 1, 1) * 2)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    r16 = torch.randn(B, C, H, W, generator=g).half()
    gout16 = torch.randn(B, C, H, W, generator=g).half()
    rm0, rv0 = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    # reference
    yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = r16.float().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    out = F.silu(F.batch_norm(yr, rm, rv, gr, br, True, 0.03, 1e-3))
    if res:
        out = out + rr
    out.backward(gout16.float())
    # engine kernels
    st = L.stream_ptr(dev)
    yd = _nhwc(y).to(dev)
    gd, bd, rmd, rvd = gamma.to(dev), beta.to(dev), rm0.clone().to(dev), rv0.clone().to(dev)
    resd = _nhwc(r16).to(dev) if res else None
    outd = torch.empty(B, H, W, C, dtype=torch.float16, device=dev)
    xh = torch.empty_like(outd)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    L.check(lib.cvx_bn_silu_train_nhwc(L.ptr(yd), B, H * W, C, L.ptr(gd), L.ptr(bd), 1e-3, 0.03, L.ptr(rmd), L.ptr(rvd), L.ptr(resd),
                                       L.ptr(outd), L.ptr(xh), L.ptr(mean), L.ptr(invstd), st), "bn fwd")
    assert rel(outd.float().permute(0, 3, 1, 2), out.detach()) < 5e-4
    y64 = y.double()
    n = y64.numel() / C
    mu64, var64 = y64.mean((0, 2, 3)), y64.var((0, 2, 3), unbiased=False)
    np.testing.assert_allclose(mean.cpu().numpy(), mu64.numpy(), rtol=1e-5, atol=1e-6)
    # E[y^2] - E[y]^2 from fp32 block partial sums: the cases with |mean| ~ 30 sigma lose ~900 x 6e-8 of the variance
    np.testing.assert_allclose(invstd.cpu().numpy(), (var64 + 1e-3).rsqrt().numpy(), rtol=1e-4)
    # running statistics: momentum 0.03, unbiased variance (fp64 reference: torch's fp32 variance itself is off by ~3e-5)
    np.testing.assert_allclose(rmd.cpu().numpy(), (0.97 * rm0.double() + 0.03 * mu64).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), (0.97 * rv0.double() + 0.03 * var64 * n / (n - 1)).numpy(), rtol=1e-4, atol=1e-6)
    xh_ref = (y - y.mean((0, 2, 3), keepdim=True)) * (y.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-3).rsqrt()
    assert rel(xh.float().permute(0, 3, 1, 2), xh_ref) < 5e-4
    goutd = _nhwc(gout16).to(dev)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.empty_like(outd)
    gres = torch.full((B, H, W, C), 0.5, dtype=torch.float16, device=dev) if res else None
    L.check(lib.cvx_bn_silu_bwd_nhwc(L.ptr(xh), L.ptr(goutd), B, H * W, C, L.ptr(gd), L.ptr(bd), L.ptr(invstd), 1.0, L.ptr(dgam), L.ptr(dbet),
                                     L.ptr(dy), L.ptr(gres), 1, st), "bn bwd")
    # dy is stored in fp16 and computed from the fp16 xhat: one rounding of the operand, one of the result
    assert rel(dy.float().permute(0, 3, 1, 2), yr.grad) < 2e-3
    assert rel(dgam, gr.grad) < 1e-3 and rel(dbet, br.grad) < 1e-3
    if res:                                              # residual branch: gradient passes through, accumulated onto 0.5
        assert rel(gres.float().permute(0, 3, 1, 2), rr.grad + 0.5) < 5e-4


@pytest.mark.parametrize("B,H,W,C,act,res_pre", [(2, 20, 20, 64, 1, 0), (2, 9, 11, 256, 1, 1), (1, 16, 16, 2048, 1, 1), (3, 7, 5, 1024, 2, 0),
                                                   (2, 8, 8, 512, 2, 1), (16, 1, 1, 256, 1, 0), (2, 10, 10, 256, 0, 1), (1, 20, 20, 512, 0, 1)])
def test_bn_act_train_and_backward_unit(dev, B, H, W, C, act, res_pre):
    """The generalised BatchNorm passes (bn_act.hip): ReLU / no activation, the residual inside the activation (Bottleneck.forward,
    resnet.py:139-141), up to 2048 channels (layer4) and down to a 1 x 1 map (ASPPPooling, deeplabv3plus.py:29-33), against torch
    autograd in fp32."""
    lib = L.load()
    g = torch.Generator().manual_seed(B * 100 + C + act)
    y = (torch.randn(B, C, H, W, generator=g) * (torch.rand(C, generator=g) * 3 + 0.2).view(1, C, 1, 1) + torch.randn(C, generator=g).view(1, C, 1, 1) * 2)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    r16 = torch.randn(B, C, H, W, generator=g).half()
    gout16 = torch.randn(B, C, H, W, generator=g).half()
    rm0, rv0 = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = r16.float().requires_grad_(True)
    z = F.batch_norm(yr, rm0.clone(), rv0.clone(), gr, br, True, 0.1, 1e-5)
    if res_pre:
        z = z + rr
    out = F.relu(z) if act == 1 else (F.silu(z) if act == 0 else z)      # act 0 + res_pre: YOLOv7's RepConv, silu(bn(a) + bn(b))
    out.backward(gout16.float())
    st = L.stream_ptr(dev)
    yd = _nhwc(y).to(dev)
    gd, bd, rmd, rvd = gamma.to(dev), beta.to(dev), rm0.clone().to(dev), rv0.clone().to(dev)
    resd = _nhwc(r16).to(dev) if res_pre else None
    outd = torch.empty(B, H, W, C, dtype=torch.float16, device=dev)
    xh = torch.empty_like(outd)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    L.check(lib.cvx_bn_act_train_nhwc(L.ptr(yd), B, H * W, C, L.ptr(gd), L.ptr(bd), 1e-5, 0.1, L.ptr(rmd), L.ptr(rvd), L.ptr(resd), act, res_pre,
                                      L.ptr(outd), L.ptr(xh), L.ptr(mean), L.ptr(invstd), st), "bn fwd")
    assert rel(outd.float().permute(0, 3, 1, 2), out.detach()) < 5e-4
    y64 = y.double()
    n = y64.numel() / C
    mu64, var64 = y64.mean((0, 2, 3)), y64.var((0, 2, 3), unbiased=False)
    np.testing.assert_allclose(rmd.cpu().numpy(), (0.9 * rm0.double() + 0.1 * mu64).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), (0.9 * rv0.double() + 0.1 * var64 * n / (n - 1)).numpy(), rtol=1e-4, atol=1e-6)
    goutd = _nhwc(gout16).to(dev)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.empty_like(outd)
    gres = torch.full((B, H, W, C), 0.5, dtype=torch.float16, device=dev) if res_pre else None
    fwd_operand = resd if (act == 0 and res_pre) else outd                # ReLU: its own output (mask); SiLU + pre-residual: the residual
    L.check(lib.cvx_bn_act_bwd_nhwc(L.ptr(xh), L.ptr(goutd), L.ptr(fwd_operand), B, H * W, C, L.ptr(gd), L.ptr(bd), L.ptr(invstd), act, res_pre, 1.0,
                                    L.ptr(dgam), L.ptr(dbet), L.ptr(dy), L.ptr(gres), 1, st), "bn bwd")
    assert rel(dy.float().permute(0, 3, 1, 2), yr.grad) < 2e-3
    assert rel(dgam, gr.grad) < 1e-3 and rel(dbet, br.grad) < 1e-3
    if res_pre:                                          # the identity branch receives the pre-activation gradient, accumulated onto 0.5
        assert rel(gres.float().permute(0, 3, 1, 2), rr.grad + 0.5) < 5e-4
    # refused combinations fail loudly
    with pytest.raises(L.CvxError):
        L.check(lib.cvx_bn_act_train_nhwc(L.ptr(yd), B, H * W, C, L.ptr(gd), L.ptr(bd), 1e-5, 0.1, L.ptr(rmd), L.ptr(rvd), L.ptr(outd), 1, 0,
                                          L.ptr(outd), L.ptr(xh), L.ptr(mean), L.ptr(invstd), st), "bn fwd")


def test_bn_statistics_do_not_overflow_at_large_magnitudes(dev):
    """Sums of squares far beyond the +-8.6e9 range of the first fixed-point format: pre-BN values
    of magnitude 3e3 over 400k rows -> sum of squares ~4e12 per channel."""
    lib = L.load()
    B, H, W, C = 4, 320, 320, 16
    g = torch.Generator().manual_seed(1)
    y = torch.randn(B, H, W, C, generator=g) * 3000.0 + 500.0
    yd = y.to(dev)
    out, xh = torch.empty(B, H, W, C, dtype=torch.float16, device=dev), torch.empty(B, H, W, C, dtype=torch.float16, device=dev)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ones, zeros = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    L.check(lib.cvx_bn_silu_train_nhwc(L.ptr(yd), B, H * W, C, L.ptr(ones), L.ptr(zeros), 1e-3, 0.03, L.ptr(rm), L.ptr(rv), None, L.ptr(out),
                                       L.ptr(xh), L.ptr(mean), L.ptr(invstd), L.stream_ptr(dev)), "bn fwd")
    y64 = y.double().reshape(-1, C)
    np.testing.assert_allclose(mean.cpu().numpy(), y64.mean(0).numpy(), rtol=1e-5)
    np.testing.assert_allclose(invstd.cpu().numpy(), (y64.var(0, unbiased=False) + 1e-3).rsqrt().numpy(), rtol=1e-5)


@pytest.mark.parametrize("B,H,W,C", [(2, 20, 20, 128), (1, 5, 7, 8), (3, 4, 4, 256), (1, 33, 2, 16)])
def test_maxpool5_and_upsample2_units(dev, B, H, W, C):
    """SPPF's 5x5/s1/p2 max pool (modules.py:312-318) and nn.Upsample(x2, nearest) (yolo_v8.py:39), forward and backward,
    against torch (exact: the ops only move fp16 values; ties resolved like torch's first-max scan)."""
    lib = L.load()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(H * W + C)
    x16 = (torch.randint(-8, 9, (B, C, H, W), generator=g).float() / 4).half()     # many exact ties
    go16 = torch.randn(B, C, H, W, generator=g).half()
    xr = x16.float().requires_grad_(True)
    ref = F.max_pool2d(xr, 5, 1, 2)
    ref.backward(go16.float())
    xd = _nhwc(x16).to(dev)
    out = torch.empty_like(xd)
    idx = torch.empty(B, H, W, C, dtype=torch.uint8, device=dev)
    L.check(lib.cvx_maxpool5_nhwc(L.ptr(xd), B, H, W, C, L.ptr(out), L.ptr(idx), st), "pool")
    assert torch.equal(out.float().permute(0, 3, 1, 2).cpu(), ref.detach())
    gin = torch.empty_like(xd)
    L.check(lib.cvx_maxpool5_bwd_nhwc(L.ptr(_nhwc(go16).to(dev)), L.ptr(idx), B, H, W, C, L.ptr(gin), 0, st), "pool bwd")
    assert rel(gin.float().permute(0, 3, 1, 2), xr.grad) < 5e-4            # up to 25 fp16 terms summed in fp32, rounded once
    # upsample
    up = torch.empty(B, 2 * H, 2 * W, C, dtype=torch.float16, device=dev)
    L.check(lib.cvx_upsample2_nhwc(L.ptr(xd), B, H, W, C, L.ptr(up), st), "up")
    assert torch.equal(up.float().permute(0, 3, 1, 2).cpu(), F.interpolate(x16.float(), scale_factor=2.0, mode="nearest"))
    gu16 = torch.randn(B, C, 2 * H, 2 * W, generator=g).half()
    xr2 = x16.float().requires_grad_(True)
    F.interpolate(xr2, scale_factor=2.0, mode="nearest").backward(gu16.float())
    gin2 = torch.full((B, H, W, C), 1.0, dtype=torch.float16, device=dev)
    L.check(lib.cvx_upsample2_bwd_nhwc(L.ptr(_nhwc(gu16).to(dev)), B, H, W, C, L.ptr(gin2), 1, st), "up bwd")
    assert rel(gin2.float().permute(0, 3, 1, 2), xr2.grad + 1.0) < 5e-4


@pytest.mark.parametrize("B,H,W,C", [(2, 75, 75, 16), (1, 38, 51, 8), (3, 7, 5, 64), (1, 1, 9, 24)])
def test_inference_pool_resize_norm_units(dev, B, H, W, C):
    """The pooling / resampling / normalisation kernels of the DLA, ResNet + DeepLab and VGG + SSD graphs against torch on odd
    sizes: 2x2/2 max pool in floor and ceil mode (ssd_model.py:16-18), 3x3 pad-1 max pool at stride 1 and 2 (ssd_model.py:30,
    resnet.py:163), global average pool, bilinear resize with align_corners=False up, down and from 1x1
    (deeplabv3plus.py:38,117-122,147), L2Normalize (ssd_model.py:113-128).  Max pools exact; the others to one fp16 rounding."""
    lib = L.load()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(H * 131 + W * 7 + C)
    x16 = torch.randn(B, C, H, W, generator=g).half()
    xd = _nhwc(x16).to(dev)
    xf = x16.float()

    def run_pool(k, s, ceil, ref):
        out = torch.empty(B, ref.shape[2], ref.shape[3], C, dtype=torch.float16, device=dev)
        L.check(lib.cvx_maxpool_nhwc(L.ptr(xd), B, H, W, C, k, s, ceil, L.ptr(out), st), f"maxpool {k}/{s}/{ceil}")
        assert torch.equal(out.float().permute(0, 3, 1, 2).cpu(), ref), (k, s, ceil)

    if H >= 2 and W >= 2:
        run_pool(2, 2, 0, F.max_pool2d(xf, 2, 2))
    run_pool(2, 2, 1, F.max_pool2d(xf, 2, 2, ceil_mode=True))
    run_pool(3, 1, 0, F.max_pool2d(xf, 3, 1, 1))
    run_pool(3, 2, 0, F.max_pool2d(xf, 3, 2, 1))
    avg = torch.empty(B, 1, 1, C, dtype=torch.float16, device=dev)
    L.check(lib.cvx_avgpool_global_nhwc(L.ptr(xd), B, H * W, C, L.ptr(avg), st), "avgpool")
    assert rel(avg.float().permute(0, 3, 1, 2).cpu(), F.adaptive_avg_pool2d(xf, 1)) < 1e-3
    for (oh, ow) in ((2 * H + 1, 3 * W), (max(H // 2, 1), max(W // 3, 1)), (H, W), (4 * H - 3, 4 * W - 3)):
        out = torch.empty(B, oh, ow, C, dtype=torch.float16, device=dev)
        L.check(lib.cvx_resize_bilinear_nhwc(L.ptr(xd), B, H, W, C, oh, ow, L.ptr(out), st), "resize")
        ref = F.interpolate(xf, size=(oh, ow), mode="bilinear", align_corners=False)
        assert (out.float().permute(0, 3, 1, 2).cpu() - ref).abs().max() < 4e-3, (oh, ow)        # one fp16 rounding of values up to ~4
    one = _nhwc(x16[:, :, :1, :1]).to(dev)
    out = torch.empty(B, 5, 6, C, dtype=torch.float16, device=dev)
    L.check(lib.cvx_resize_bilinear_nhwc(L.ptr(one), B, 1, 1, C, 5, 6, L.ptr(out), st), "broadcast")
    assert torch.equal(out.cpu(), _nhwc(x16[:, :, :1, :1]).expand(B, 5, 6, C))
    wgt = (torch.rand(C, generator=g) * 30 + 1).to(dev)
    out = torch.empty_like(xd)
    L.check(lib.cvx_l2norm_nhwc(L.ptr(xd), L.ptr(wgt), B, H * W, C, L.ptr(out), st), "l2norm")
    ref = wgt.cpu().view(1, C, 1, 1) * (xf / (xf.pow(2).sum(1, keepdim=True).sqrt() + 1e-10))
    assert rel(out.float().permute(0, 3, 1, 2).cpu(), ref) < 5e-4


@pytest.mark.parametrize("B,H,W,C", [(2, 33, 33, 16), (1, 38, 51, 8), (3, 7, 5, 64), (2, 1, 1, 24)])
def test_training_pool_resize_dropout_units(dev, B, H, W, C):
    """Backward kernels of the ResNet / DeepLab ops against torch autograd (fp32): 3x3 pad-1 max pool at stride 2 and 1 (ties
    broken like torch: first maximum in scan order), global average pool, bilinear resize (align_corners = False) up, down,
    identity and from 1 x 1, each also in accumulate mode; dropout: mask statistics, scale, and backward = same mask."""
    lib = L.load()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(H * 131 + W * 7 + C)
    x16 = (torch.randint(-8, 9, (B, C, H, W), generator=g).float() / 4).half()     # many exact ties for the pools
    xd = _nhwc(x16).to(dev)
    base16 = torch.randn(B, C, H, W, generator=g).half()
    for stride in (2, 1):
        xr = x16.float().requires_grad_(True)
        ref = F.max_pool2d(xr, 3, stride, 1)
        oh, ow = ref.shape[2:]
        go16 = torch.randn(B, C, oh, ow, generator=g).half()
        ref.backward(go16.float())
        out = torch.empty(B, oh, ow, C, dtype=torch.float16, device=dev)
        am = torch.empty(B, oh, ow, C, dtype=torch.uint8, device=dev)
        L.check(lib.cvx_maxpool3_train_nhwc(L.ptr(xd), B, H, W, C, stride, L.ptr(out), L.ptr(am), st), "maxpool3 train")
        assert torch.equal(out.float().permute(0, 3, 1, 2).cpu(), ref.detach())
        for acc in (0, 1):
            gin = _nhwc(base16).to(dev).clone()
            L.check(lib.cvx_maxpool3_bwd_nhwc(L.ptr(_nhwc(go16).to(dev)), L.ptr(am), B, H, W, C, stride, L.ptr(gin), acc, st), "maxpool3 bwd")
            want = xr.grad + (base16.float() if acc else 0)
            assert (gin.float().permute(0, 3, 1, 2).cpu() - want).abs().max() <= 2e-3 * max(1.0, float(want.abs().max()))
    for ceil in ((0, 1) if H >= 2 and W >= 2 else (1,)):              # 2x2 / stride 2 (YOLOv7 Transition_Block, SSD's ceil-mode pool)
        xr = x16.float().requires_grad_(True)
        ref = F.max_pool2d(xr, 2, 2, ceil_mode=bool(ceil))
        go16 = torch.randn(*ref.shape, generator=g).half()
        ref.backward(go16.float())
        for acc in (0, 1):
            gin = _nhwc(base16).to(dev).clone()
            L.check(lib.cvx_maxpool2_bwd_nhwc(L.ptr(xd), L.ptr(_nhwc(go16).to(dev)), B, H, W, C, ceil, L.ptr(gin), acc, st), "maxpool2 bwd")
            want = xr.grad + (base16.float() if acc else 0)
            assert (gin.float().permute(0, 3, 1, 2).cpu() - want).abs().max() <= 2e-3 * max(1.0, float(want.abs().max())), (ceil, acc)
    xf16 = torch.randn(B, C, H, W, generator=g).half()
    # global average pool
    xr = xf16.float().requires_grad_(True)
    go16 = torch.randn(B, C, 1, 1, generator=g).half()
    F.adaptive_avg_pool2d(xr, 1).backward(go16.float())
    for acc in (0, 1):
        gin = _nhwc(base16).to(dev).clone()
        L.check(lib.cvx_avgpool_global_bwd_nhwc(L.ptr(_nhwc(go16).to(dev)), B, H * W, C, L.ptr(gin), acc, st), "avgpool bwd")
        want = xr.grad + (base16.float() if acc else 0)
        assert (gin.float().permute(0, 3, 1, 2).cpu() - want).abs().max() <= 2e-3 * max(1.0, float(want.abs().max()))
    # bilinear resize
    for (oh, ow) in ((2 * H + 1, 3 * W), (max(H // 2, 1), max(W // 3, 1)), (H, W), (4 * H - 3, 4 * W - 3), (33, 33)):
        xr = xf16.float().requires_grad_(True)
        go16 = torch.randn(B, C, oh, ow, generator=g).half()
        F.interpolate(xr, size=(oh, ow), mode="bilinear", align_corners=False).backward(go16.float())
        for acc in (0, 1):
            gin = _nhwc(base16).to(dev).clone()
            L.check(lib.cvx_resize_bilinear_bwd_nhwc(L.ptr(_nhwc(go16).to(dev)), B, H, W, C, oh, ow, L.ptr(gin), acc, st), "resize bwd")
            want = xr.grad + (base16.float() if acc else 0)
            assert (gin.float().permute(0, 3, 1, 2).cpu() - want).abs().max() <= 2e-3 * max(1.0, float(want.abs().max())), (oh, ow, acc)
    # dropout
    n = B * H * W * C
    ones = torch.ones(B, H, W, C, dtype=torch.float16, device=dev)
    out = torch.empty_like(ones)
    L.check(lib.cvx_dropout_nhwc(L.ptr(ones), B, H * W, C, 0.1, 1234, L.ptr(out), 0, st), "dropout")
    vals = out.float().cpu().flatten()
    kept = vals > 0
    assert torch.all((vals[kept] - 1 / 0.9).abs() < 1e-3)
    assert abs(float(kept.float().mean()) - 0.9) < 4 * (0.09 / n) ** 0.5 + 1e-9
    out2 = torch.empty_like(ones)
    L.check(lib.cvx_dropout_nhwc(L.ptr(ones), B, H * W, C, 0.1, 1234, L.ptr(out2), 0, st), "dropout again")
    assert torch.equal(out, out2)                                                       # same seed, same mask (what the backward relies on)
    L.check(lib.cvx_dropout_nhwc(L.ptr(ones), B, H * W, C, 0.1, 1235, L.ptr(out2), 0, st), "dropout other seed")
    assert n < 64 or not torch.equal(out, out2)
    prev = out2.clone()
    L.check(lib.cvx_dropout_nhwc(L.ptr(ones), B, H * W, C, 0.0, 7, L.ptr(out2), 1, st), "dropout p=0 accumulate")
    assert torch.equal(out2, (prev.float() + 1).half())                                 # p = 0 keeps everything; accumulate adds


@pytest.mark.parametrize("B,H,W,Co", [(2, 64, 64, 16), (1, 32, 96, 32), (3, 16, 16, 48), (1, 128, 128, 80),
                                      (2, 640, 640, 16), (1, 72, 200, 16), (2, 96, 160, 32)])   # round 5: the bench's width (10 tiles per row), maps that end inside a tile
def test_fp32_stem_unit(dev, B, H, W, Co):
    """stem.hip (model.0 = Conv(3, c, 3, 2), yolo_v8.py:28) against torch fp32: train pass (batch statistics, running
    update, xhat), eval pass (folded scale / shift) and the weight gradient."""
    lib = L.load()
    st = L.stream_ptr(dev)
    g = torch.Generator().manual_seed(Co + H)
    x = torch.rand(B, 3, H, W, generator=g)
    w = (torch.rand(Co, 3, 3, 3, generator=g) - 0.5) * 0.4
    gamma, beta = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.3
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(x, wr, None, 2, 1)
    rm, rv = torch.zeros(Co), torch.full((Co,), 0.9)
    ref = F.silu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.03, 1e-3))
    xd, wd = x.to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev)          # [cout][kh][kw][ci]
    out = torch.empty(B, H // 2, W // 2, Co, dtype=torch.float16, device=dev)
    xh = torch.empty_like(out)
    mean, invstd = torch.empty(Co, device=dev), torch.empty(Co, device=dev)
    rmd, rvd = torch.zeros(Co, device=dev), torch.full((Co,), 0.9, device=dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    L.check(lib.cvx_stem_train_nchw(L.ptr(xd), B, H, W, L.ptr(wd), Co, L.ptr(gd), L.ptr(bd), 1e-3, 0.03, L.ptr(rmd), L.ptr(rvd), L.ptr(out),
                                    L.ptr(xh), L.ptr(mean), L.ptr(invstd), st), "stem train")
    assert rel(out.float().permute(0, 3, 1, 2), ref.detach()) < 5e-4          # one fp16 rounding of the stored activation
    np.testing.assert_allclose(mean.cpu().numpy(), y.detach().mean((0, 2, 3)).numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=2e-5, atol=1e-6)
    sc, sh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    scd, shd = sc.to(dev), sh.to(dev)
    L.check(lib.cvx_stem_eval_nchw(L.ptr(xd), B, H, W, L.ptr(wd), Co, L.ptr(scd), L.ptr(shd), L.ptr(out), st), "stem eval")
    assert rel(out.float().permute(0, 3, 1, 2), F.silu(y.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))) < 5e-4
    dy16 = torch.randn(B, Co, H // 2, W // 2, generator=g).half()
    y.backward(dy16.float(), retain_graph=True)
    dw = torch.empty(Co, 3, 3, 3, device=dev)
    L.check(lib.cvx_stem_wgrad_nchw(L.ptr(xd), B, H, W, L.ptr(_nhwc(dy16).to(dev)), Co, L.ptr(dw), st), "stem wgrad")
    assert rel(dw.permute(0, 3, 1, 2), wr.grad) < 1e-5
    # the fused backward the engine runs: BatchNorm + SiLU backward and the weight gradient in one pass, from gout
    L.check(lib.cvx_stem_train_nchw(L.ptr(xd), B, H, W, L.ptr(wd), Co, L.ptr(gd), L.ptr(bd), 1e-3, 0.03, L.ptr(rmd), L.ptr(rvd), L.ptr(out),
                                    L.ptr(xh), L.ptr(mean), L.ptr(invstd), st), "stem train")
    wr2, gr2, br2 = w.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref2 = F.silu(F.batch_norm(F.conv2d(x, wr2, None, 2, 1), None, None, gr2, br2, True, 0.03, 1e-3))
    ref2.backward(dy16.float())
    dgam, dbet = torch.zeros(Co, device=dev), torch.zeros(Co, device=dev)
    L.check(lib.cvx_stem_backward_nchw(L.ptr(xd), B, H, W, L.ptr(xh), L.ptr(_nhwc(dy16).to(dev)), Co, L.ptr(gd), L.ptr(bd), L.ptr(invstd), 1.0,
                                       L.ptr(dgam), L.ptr(dbet), L.ptr(dw), st), "stem backward")
    assert rel(dgam, gr2.grad) < 1e-3 and rel(dbet, br2.grad) < 1e-3
    assert rel(dw.permute(0, 3, 1, 2), wr2.grad) < 3e-3       # dy passes through one fp16 rounding (as bn_bwd_apply stores it) and xhat is fp16
    # ... and the route the engine takes on every YOLOv8 training shape: ONE pass, xhat recomputed from the images (no xhat argument), the
    # weight-gradient sums on fp16 MFMAs.  (Shapes with a single pixel split take the route above in the engine too.)
    if B * (H // 2) * (W // 2) > 1024 and W % 4 == 0:
        dgam2, dbet2, dw2 = torch.zeros(Co, device=dev), torch.zeros(Co, device=dev), torch.empty(Co, 3, 3, 3, device=dev)
        L.check(lib.cvx_stem_backward_recompute_nchw(L.ptr(xd), B, H, W, L.ptr(wd), L.ptr(_nhwc(dy16).to(dev)), Co, L.ptr(gd), L.ptr(bd), L.ptr(mean), L.ptr(invstd),
                                                     1.0, L.ptr(dgam2), L.ptr(dbet2), L.ptr(dw2), st), "stem backward (one pass, recomputed xhat)")
        assert rel(dgam2, gr2.grad) < 1e-3 and rel(dbet2, br2.grad) < 1e-3
        assert rel(dw2.permute(0, 3, 1, 2), wr2.grad) < 3e-3


# ---- whole network -----------------------------------------------------------------------------------------
def test_forward_matches_golden_and_oracle(dev, gold):
    g = gold("yolov8n_fwd_128.npz")
    x = torch.from_numpy(g["x"])
    m = new_model(dev).train()
    with torch.no_grad():
        outs = m(x.to(dev))
    assert [tuple(o.shape) for o in outs] == [(2, 144, 16, 16), (2, 144, 8, 8), (2, 144, 4, 4)]
    r = level_report("128x128 fixture vs reference fp32", outs, [torch.from_numpy(g[f"train{i}"]) for i in range(3)])
    assert max(r) < LEVEL_TOL, r                                  # the reference's own (fp32 CPU) outputs
    with fp16_storage():
        sd = O.init_state_dict("n", 80, seed=0)
        emu = O.forward(sd, x, "n", 80, training=True)
    r = level_report("128x128 fixture vs fp16-storage emulation", outs, [e.detach() for e in emu])
    assert max(r) < LEVEL_TOL, r                                  # same arithmetic, rounding points emulated on the CPU
    # BN running statistics after one training forward (momentum 0.03, unbiased variance)
    for k in g.files:
        if k.startswith("bn:"):
            # the stem runs in fp32 (statistics to fp32 round-off); the deepest head BN sees 30 layers of fp16 activations
            assert rel(m.state_dict()[k[3:]], torch.from_numpy(g[k])) < (1e-5 if "model.0." in k else 2e-2), k
    assert int(m.state_dict()["model.0.bn.num_batches_tracked"]) == 2
    # eval mode: (y, feats); compare the head logits and the decoded output
    m.eval()
    with torch.no_grad():
        y, feats = m(x.to(dev))
    assert tuple(y.shape) == (2, 84, 336) and len(feats) == 3
    # decoded output: boxes in pixels (|v| <= 128 here) and class probabilities; the logits behind them carry <= 1e-3 (LEVEL_TOL), the DFL
    # softmax expectation and the sigmoid do not amplify it: 2e-3 (relative for the box coordinates, absolute for the probabilities)
    np.testing.assert_allclose(y.cpu().numpy(), g["eval_y"], rtol=2e-3, atol=2e-3)


def test_forward_640_subsample(dev, gold):
    g = gold("yolov8n_fwd_640_sub.npz")
    m = new_model(dev).train()
    with torch.no_grad():
        outs = m(synth.images(1, 640, 640, seed=1).to(dev))
    r = level_report("640x640 sub-sample vs reference fp32", [o.flatten()[::97].reshape(1, -1, 1, 1) for o in outs],
                     [torch.from_numpy(g[f"lvl{i}"]).reshape(1, -1, 1, 1) for i in range(3)])
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"lvl{i}"])
        assert rel(o.flatten()[::97], ref) < LEVEL_TOL, (i, r)
        assert abs(float(o.norm()) / float(g["norms"][i]) - 1) < 1e-3


def test_model_scale_s_runs_and_matches_oracle(dev):
    x = synth.images(2, 128, 128, seed=4)
    m = new_model(dev, scale="s").train()
    with torch.no_grad():
        outs = m(x.to(dev))
    ref32 = O.forward(O.init_state_dict("s", 80, seed=0), x, "s", 80, training=True)
    r = level_report("YOLOv8-s 128x128 vs oracle fp32", outs, [t.detach() for t in ref32])
    assert max(r) < LEVEL_TOL, r


def test_non_square_input_matches_oracle(dev):
    """96 x 160 (3 x 5 cells at stride 32): tile edges in both directions, widths that are not multiples of 16."""
    x = torch.rand(3, 3, 96, 160, generator=torch.Generator().manual_seed(11))
    m = new_model(dev).train()
    with torch.no_grad():
        outs = m(x.to(dev))
    assert [tuple(o.shape) for o in outs] == [(3, 144, 12, 20), (3, 144, 6, 10), (3, 144, 3, 5)]
    ref32 = O.forward(O.init_state_dict("n", 80, seed=0), x, "n", 80, training=True)
    r = level_report("96x160 vs oracle fp32", outs, [t.detach() for t in ref32])
    assert max(r[:2]) < LEVEL_TOL, r   # P3, P4: the stated bar
    # P5 is 3 x 5 cells: 45 samples per BatchNorm channel at the deepest level, where a batch statistic over so few samples amplifies the
    # fp16 operand rounding the reference's own CUDA autocast path has too.  The yardstick there is the oracle with exactly the engine's
    # rounding points emulated (O.FP16_STORAGE): the engine may not exceed it by more than a quarter (and never 1.3e-3).
    O.FP16_STORAGE[0] = True
    try:
        emu = O.forward(O.init_state_dict("n", 80, seed=0), x, "n", 80, training=True)
    finally:
        O.FP16_STORAGE[0] = False
    r_emu = float((emu[2] - ref32[2]).norm() / ref32[2].norm())
    print("P5: engine", r[2], "fp16-rounding emulation of the reference", r_emu)
    assert r[2] < max(LEVEL_TOL, 1.25 * r_emu) and r[2] < 1.3e-3, (r, r_emu)
    # and a fused training step on it stays finite (data / weight gradients at the same odd tile edges)
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m, lr=1e-3))
    batch = {k: v.to(dev) for k, v in synth.targets(3, seed=5).items()}
    losses = [float(step(x.to(dev), batch).sum()) for _ in range(4)]
    assert all(np.isfinite(losses)) and bool(torch.isfinite(m.flat_params).all())


# ---- loss ------------------------------------------------------------------------------------------------------
def _loss_case(B, H, seed):
    hw = [(H // s, H // s) for s in (8, 16, 32)]
    A = sum(a * b for a, b in hw)
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(B, A, 144, generator=g)
    pred[..., 64:] = pred[..., 64:] * 2 - 3
    return pred, synth.targets(B, seed=seed), hw


def _oracle_loss(pred, batch, hw):
    pr = pred.clone().requires_grad_(True)
    feats, off = [], 0
    for (h, w) in hw:
        feats.append(pr[:, off:off + h * w].permute(0, 2, 1).reshape(pred.shape[0], 144, h, w))
        off += h * w
    aux = {}
    loss, items = O.v8_loss(feats, batch, 80, aux=aux)
    loss.backward()
    return items, pr.grad, aux


@pytest.mark.parametrize("B,H,seed", [(2, 128, 3), (4, 160, 4), (3, 320, 5), (1, 64, 6)])
def test_loss_value_and_gradient(dev, B, H, seed):
    from computervision.pytorch_amd.train import flatten_targets
    pred, batch, hw = _loss_case(B, H, seed)
    items_ref, grad_ref, aux = _oracle_loss(pred, batch, hw)
    items, dpred = E.V8LossOp(80)(pred.to(dev), flatten_targets(batch, dev), hw, (8, 16, 32), 256.0)
    np.testing.assert_allclose(items.cpu().numpy(), items_ref.numpy(), rtol=2e-5)
    assert rel(dpred.float() / 256.0, grad_ref) < 5e-4


def test_loss_edge_cases(dev):
    from computervision.pytorch_amd.train import flatten_targets
    pred, batch, hw = _loss_case(3, 128, 9)
    # (a) no targets at all
    empty = {"batch_idx": torch.zeros(0), "cls": torch.zeros(0, 1), "bboxes": torch.zeros(0, 4)}
    it_ref, g_ref, _ = _oracle_loss(pred, empty, hw)
    it, dp = E.V8LossOp(80)(pred.to(dev), flatten_targets(empty, dev), hw, (8, 16, 32), 64.0)
    np.testing.assert_allclose(it.cpu().numpy(), it_ref.numpy(), rtol=2e-5)
    assert rel(dp.float() / 64.0, g_ref) < 5e-4
    # (b) ragged: image 1 has no boxes, image 2 has many overlapping ones, one degenerate (zero-size) box
    bi = torch.tensor([0., 0., 2., 2., 2., 2., 2., 2.])
    cls = torch.tensor([[1.], [5.], [7.], [7.], [9.], [3.], [3.], [79.]])
    bb = torch.tensor([[.5, .5, .4, .4], [.3, .3, .2, .3], [.5, .5, .5, .5], [.52, .5, .5, .5], [.5, .52, .45, .5], [.7, .7, .2, .2],
                       [.2, .8, .3, .2], [0., 0., 0., 0.]])
    ragged = {"batch_idx": bi, "cls": cls, "bboxes": bb}
    it_ref, g_ref, aux = _oracle_loss(pred, ragged, hw)
    it, dp = E.V8LossOp(80)(pred.to(dev), flatten_targets(ragged, dev), hw, (8, 16, 32), 64.0)
    np.testing.assert_allclose(it.cpu().numpy(), it_ref.numpy(), rtol=5e-5)
    assert rel(dp.float() / 64.0, g_ref) < 5e-4
    # (c) targets given out of image order are regrouped (stable) by the wrapper
    perm = torch.tensor([2, 0, 3, 1, 4, 5, 6, 7])
    shuffled = {"batch_idx": bi[perm], "cls": cls[perm], "bboxes": bb[perm]}
    it2, _ = E.V8LossOp(80)(pred.to(dev), flatten_targets(shuffled, dev), hw, (8, 16, 32), 64.0)
    np.testing.assert_allclose(it2.cpu().numpy(), it.cpu().numpy(), rtol=1e-6)


def _logit(p):
    return torch.log(p) - torch.log1p(-p)


def test_assigner_fixture_through_the_loss(dev, gold):
    """The reference-captured TaskAlignedAssigner case (tests/golden/tal_assign.npz: padded target rows, two heavily
    overlapping targets -> anchors claimed twice) through the HIP loss: logits are built whose sigmoid / DFL expectation
    reproduce the fixture's scores and boxes, then the per-anchor assignment of ``loss_v8.hip`` is read back and compared
    with the reference's fg_mask / target_gt_idx BIT-EXACTLY, its normalised target scores and classes to fp32 round-off."""
    g = gold("tal_assign.npz")
    pd_scores, pd_bboxes, anc = (torch.from_numpy(g[k]) for k in ("pd_scores", "pd_bboxes", "anc"))
    gt_labels, gt_bboxes = torch.from_numpy(g["gt_labels"]), torch.from_numpy(g["gt_bboxes"])
    B, A, nc = pd_scores.shape
    G = gt_bboxes.shape[1]
    hw, strides = [(16, 16), (8, 8), (4, 4)], (8.0, 16.0, 32.0)
    stride_t = torch.cat([torch.full((h * w,), s) for (h, w), s in zip(hw, strides)])
    # DFL logits whose softmax expectation is the fixture's ltrb distance (two neighbouring bins carry the mass)
    ltrb = torch.cat((anc[None] - pd_bboxes[..., :2], pd_bboxes[..., 2:] - anc[None]), -1) / stride_t[None, :, None]
    assert float(ltrb.min()) >= 0 and float(ltrb.max()) < 15
    lo = ltrb.floor()
    frac = (ltrb - lo).double()
    box = torch.full((B, A, 4, 16), -80.0, dtype=torch.float64)
    box.scatter_(3, lo.long().unsqueeze(-1), torch.log1p(-frac).unsqueeze(-1).clamp_min(-80.0))
    box.scatter_(3, (lo.long() + 1).unsqueeze(-1), torch.log(frac.clamp_min(1e-35)).unsqueeze(-1).clamp_min(-80.0))
    pred = torch.cat((box.reshape(B, A, 64).float(), _logit(pd_scores.double()).float()), 2).contiguous()
    # every fixture row as a target (zero rows included: the kernel must treat them like the reference's padding)
    wh = 128.0
    rows = []
    for b in range(B):
        for k in range(G):
            x1, y1, x2, y2 = gt_bboxes[b, k].tolist()
            rows.append([b, float(gt_labels[b, k, 0]), (x1 + x2) / 2 / wh, (y1 + y2) / 2 / wh, (x2 - x1) / wh, (y2 - y1) / wh])
    targets = torch.tensor(rows, dtype=torch.float32, device=dev)
    op = E.V8LossOp(nc)
    op(pred.to(dev), targets, hw, strides, 1.0)
    gt_index, norm = op.assignment(B, A, len(rows))
    gt_index, norm = gt_index.cpu(), norm.cpu()
    fg_ref, idx_ref = torch.from_numpy(g["fg"]), torch.from_numpy(g["gt_idx"])
    fg = gt_index >= 0
    assert torch.equal(fg, fg_ref), f"fg mask differs at {int((fg != fg_ref).sum())} anchors"
    local = gt_index - torch.arange(B).view(B, 1) * G                 # target row -> index inside its image
    assert torch.equal(local[fg], idx_ref[fg])
    assert int(fg.sum()) > 20 and bool((idx_ref[fg_ref] < G).all())
    np.testing.assert_allclose(norm[fg].numpy(), g["t_scores_sum"][fg_ref.numpy()], rtol=2e-5, atol=1e-7)
    cls_of = gt_labels[..., 0].long()
    assert torch.equal(cls_of.gather(1, local.clamp_min(0))[fg], torch.from_numpy(g["t_cls"])[fg])


def test_pack_targets_kernel(dev):
    """cvx_pack_targets against the ORACLE's Loss.preprocess (core/algorithms/yolo_v8.py:51-65, oracle build_targets): image j's rows of the
    packed (N, 6) list, in order, are row j of the reference's (B, Gmax, 5) tensor (class, then the boxes the oracle converts to pixel
    corners); the host path of flatten_targets must give the same rows."""
    from computervision.pytorch_amd.train import flatten_targets
    gen = torch.Generator().manual_seed(3)
    n, B, H, W = 777, 9, 96.0, 160.0
    batch = {"batch_idx": torch.randint(0, B, (n,), generator=gen).float(), "cls": torch.randint(0, 80, (n, 1), generator=gen).float(),
             "bboxes": torch.rand(n, 4, generator=gen)}
    ref = O.build_targets(batch, B, H, W)                               # (B, Gmax, 5): cls, x1, y1, x2, y2 in pixels
    for rows in (flatten_targets({k: v.to(dev) for k, v in batch.items()}, dev).cpu(), flatten_targets(batch, "cpu")):
        assert rows.shape == (n, 6) and bool((rows[1:, 0] >= rows[:-1, 0]).all())
        for j in range(B):
            mine = rows[rows[:, 0] == j]
            k = mine.shape[0]
            assert k == int((batch["batch_idx"] == j).sum())
            assert torch.equal(mine[:, 1], ref[j, :k, 0]) and bool((ref[j, k:] == 0).all())
            cxcywh = mine[:, 2:6] * torch.tensor([W, H, W, H])
            xyxy = torch.cat((cxcywh[:, :2] - cxcywh[:, 2:] / 2, cxcywh[:, :2] + cxcywh[:, 2:] / 2), 1)
            assert torch.equal(xyxy, ref[j, :k, 1:])
    assert flatten_targets({"batch_idx": torch.zeros(0), "cls": torch.zeros(0, 1), "bboxes": torch.zeros(0, 4)}, dev).shape == (0, 6)


# ---- full train step -----------------------------------------------------------------------------------------
def test_train_step_gradients_and_adam(dev, gold):
    from computervision.pytorch_amd.train import FlatAdam, V8DetectionLoss, flatten_targets
    from configs import Yolo8DetConfig
    g = gold("yolov8n_train_160.npz")
    x = torch.from_numpy(g["x"])
    batch = {"batch_idx": torch.from_numpy(g["batch_idx"]), "cls": torch.from_numpy(g["cls"]), "bboxes": torch.from_numpy(g["bboxes"])}
    m = new_model(dev).train()
    crit = V8DetectionLoss(Yolo8DetConfig(), m)
    opt = FlatAdam(m, lr=1e-3)
    eng = m.engine_for(160, 160)
    pred = m._run_forward(x.to(dev), training=True)
    items, dpred = crit.op(pred, flatten_targets(batch, dev), m.level_shapes(160, 160), (8, 16, 32), crit.loss_scale)
    m.flat_grads.zero_()
    eng.backward(dpred, crit.loss_scale)
    m.attach_grads()
    # loss vs the reference's own numbers (fp32 CPU), captured in the fixture
    assert abs(float(items.sum() * 4) / float(g["loss"][0]) - 1) < 2e-3
    np.testing.assert_allclose(items.cpu().numpy(), g["items"][0], rtol=3e-3)
    named = dict(m.named_parameters())
    keys = [str(k) for k in g["keys"]]
    # (1) end to end (own forward, own loss, own backward) vs the fp32 oracle.  The task-aligned top-10 assignment is
    #     a discrete function of the fp16-perturbed predictions, so this bound is loose; (1b) removes that effect.
    sd32 = O.init_state_dict("n", 80, seed=0)
    leaves = {k: sd32[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd32)
    work.update(leaves)
    feats = O.forward(work, x, "n", 80, training=True)
    for f in feats:
        f.retain_grad()
    loss32, _ = O.v8_loss(feats, batch, 80)
    loss32.backward()
    g32 = {k: leaves[k].grad for k in keys}
    ref_all = torch.cat([g32[k].flatten() for k in keys])
    mine = torch.cat([named[k].grad.flatten().cpu() for k in keys])
    assert rel(mine, ref_all) < 1.2e-1
    np.testing.assert_allclose(named["model.22.cv3.0.2.bias"].grad.cpu().numpy(), g["g_headb"], rtol=5e-2, atol=5e-4)
    # (1b) backward kernels alone: feed the ORACLE's d(loss)/d(pred) to cvx_engine_backward.  What remains is fp16
    #     storage of activations/gradients; the fp16-storage emulation of the oracle deviates from fp32 by the same
    #     ~4e-2 on this batch (tests/test_oracle_golden.py::test_fp16_storage_emulation_gap), so 6e-2 is the bar.
    dpred_ref = torch.cat([f.grad.reshape(4, 144, -1) for f in feats], 2).permute(0, 2, 1).contiguous()
    ls = 1024.0
    m._run_forward(x.to(dev), training=True)
    m.flat_grads.zero_()
    eng.backward((dpred_ref * ls).half().to(dev).contiguous(), ls)
    m.attach_grads()
    mine = torch.cat([named[k].grad.flatten().cpu() for k in keys])
    assert rel(mine, ref_all) < 6e-2
    for k, tol in (("model.22.cv3.0.2.bias", 2e-3), ("model.22.cv3.1.2.weight", 1e-2), ("model.22.cv3.0.2.weight", 2.5e-2),
                   ("model.15.cv2.conv.weight", 4e-2), ("model.0.conv.weight", 7e-2)):
        assert rel(named[k].grad, g32[k]) < tol, k
    # (2) Adam: one fused step on the flat arenas vs the oracle's update from the SAME gradients
    before = {k: named[k].detach().clone().cpu() for k in ("model.0.conv.weight", "model.4.m.1.cv2.bn.weight", "model.22.cv2.1.2.weight")}
    gk = {k: named[k].grad.detach().clone().cpu() for k in before}
    opt.step(zero_grad=True)
    O.adam_step(before, gk, {}, 1e-3)
    for k in before:
        assert rel(named[k].detach(), before[k]) < 1e-6, k
    assert float(m.flat_grads.abs().max()) == 0.0


def test_fused_steps_track_the_reference_loss_curve(dev, gold):
    """Three fused steps (forward, loss, backward, Adam) on the fixture batch: the loss must follow the
    reference's own first two recorded steps within 1%."""
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    g = gold("yolov8n_train_160.npz")
    x = torch.from_numpy(g["x"]).to(dev)
    batch = {"batch_idx": torch.from_numpy(g["batch_idx"]), "cls": torch.from_numpy(g["cls"]), "bboxes": torch.from_numpy(g["bboxes"])}
    m = new_model(dev).train()
    step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m, lr=1e-3))
    for s in range(2):
        items = step(x, batch)
        assert abs(float(items.sum() * 4) / float(g["loss"][s]) - 1) < 1e-2, s


def test_fifty_fused_steps_follow_the_oracle_loss_trajectory(dev, gold):
    """50 steps of engine (fp16 operands, fused step) vs oracle (fp32 CPU restatement of the reference's train_loop, yolo8_train.py:93-111)
    from the same seed-0 initialisation on one repeated 128x128 batch; the oracle's two curves (fp32, and with the engine's rounding
    points emulated) come from tests/golden/yolov8n_traj_128.npz (oracle/make_traj_fixture.py).  Adam at lr 1e-3 on a 2-image batch is
    chaotic step by step (the oracle's OWN curve jumps by 10-25 % between consecutive steps; its fp16 emulation leaves its fp32 run by up
    to 47 % on single steps and 24 % on 10-step window means), so the yardstick is the emulation: the engine's 10-step window means must
    stay as close to the fp32 curve as 1.5 x the emulation does (+ 5 %).  Hard bounds on top: the first two steps (before anything can
    diverge) within 2 %, the curve falls by > 6 x, the last window within 20 %."""
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    g = gold("yolov8n_traj_128.npz")
    ref, emu = g["fp32"], g["fp16_emulation"]
    x, batch = synth.images(2, 128, 128, seed=1), synth.targets(2, seed=2)
    m = new_model(dev).train()
    step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m, lr=1e-3))
    xd = x.to(dev)
    mine = np.array([float(step(xd, batch).sum()) * 2 for _ in range(50)])   # items are per-image means; the reference's loss is items.sum() * batch
    win = lambda c: c.reshape(5, 10).mean(1)  # noqa: E731
    d_eng, d_emu = np.abs(win(mine) / win(ref) - 1), np.abs(win(emu) / win(ref) - 1)
    print("10-step window means: engine", np.round(win(mine), 3), "oracle fp32", np.round(win(ref), 3), "oracle fp16-emulation", np.round(win(emu), 3),
          "| engine vs fp32", np.round(d_eng, 3), "emulation vs fp32", np.round(d_emu, 3))
    assert np.isfinite(mine).all() and win(mine)[-1] < win(mine)[0] / 6
    assert np.abs(mine[:2] / ref[:2] - 1).max() < 2e-2
    assert d_eng.max() < 1.5 * d_emu.max() + 0.05, (d_eng, d_emu)
    assert d_eng[-1] < 0.2


def test_twenty_fused_steps_follow_the_reference_loss_curve(dev, gold):
    """The REAL reference's loss curve (tests/golden/yolov8n_traj_160.npz, written by oracle/make_golden.py yolov8_traj: the imported model,
    Loss and torch.optim.Adam of core/trainer/yolo8_train.py:93-111 for 50 steps on one repeated 160x160 batch of 8 at lr 1e-4) against the
    fused engine step from the same seed-0 initialisation.  At this setting the dynamics are smooth for ~24 steps (the oracle's fp32
    restatement follows the reference to 7.5e-5 per step over steps 0-19; afterwards an assigner flip separates even the two fp32 runs), so
    the pinned window is steps 0-19.  fp16 rounding noise is amplified by Adam's first steps (m / sqrt(v) is close to sign(g) there): the
    oracle with the engine's fp16 rounding points emulated is already 1.6 % off at step 3 and 4.6 % at its worst step (2.4 % on the window
    mean).  The engine is another fp16 pipeline (other summation orders), so its bar is twice the emulation's worst step on every step
    (measured 5.6-7.0 %, depending on the run: the stat atomics' order is not fixed) and 3 % on the window mean."""
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    g = gold("yolov8n_traj_160.npz")
    n = int(g["pinned_steps"])
    ref, emu = g["reference"][:n], g["fp16_emulation"][:n]
    x, batch = synth.images(8, 160, 160, seed=11), synth.targets(8, seed=12)
    m = new_model(dev).train()
    step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m, lr=float(g["lr"])))
    xd = x.to(dev)
    mine = np.array([float(step(xd, batch).sum()) * 8 for _ in range(n)])   # items are per-image means; the reference's loss is items.sum() * batch
    d_eng, d_emu = np.abs(mine / ref - 1), np.abs(emu / ref - 1)
    print("engine vs reference per step", np.round(d_eng, 4), "max", d_eng.max(), "mean-curve", abs(mine.mean() / ref.mean() - 1),
          "| emulation vs reference max", d_emu.max(), "mean-curve", abs(emu.mean() / ref.mean() - 1))
    assert np.abs(mine[:2] / ref[:2] - 1).max() < 5e-3
    assert d_eng.max() < 2 * d_emu.max(), (d_eng, d_emu)
    assert abs(mine.mean() / ref.mean() - 1) < 0.03


def test_autograd_compat_path_matches_fused_path(dev):
    """model(x) -> criterion -> loss.backward() (the reference's train_loop spelling) fills the same gradient arena."""
    from computervision.pytorch_amd.train import V8DetectionLoss, flatten_targets
    from configs import Yolo8DetConfig
    x, batch = synth.images(2, 128, 128, seed=1).to(dev), synth.targets(2, seed=2)
    m = new_model(dev).train()
    crit = V8DetectionLoss(Yolo8DetConfig(), m)
    preds = m(x)
    loss, items = crit(preds, batch)
    loss.backward()
    g_compat = m.flat_grads.clone()
    assert dict(m.named_parameters())["model.3.conv.weight"].grad is not None
    m2 = new_model(dev).train()
    crit2 = V8DetectionLoss(Yolo8DetConfig(), m2)
    pred = m2._run_forward(x, training=True)
    it2, dpred = crit2.op(pred, flatten_targets(batch, dev), m2.level_shapes(128, 128), (8, 16, 32), crit2.loss_scale)
    m2.flat_grads.zero_()
    m2.engine_for(128, 128).backward(dpred, crit2.loss_scale)
    assert abs(float(loss) - float(it2.sum() * 2)) < 1e-3 * abs(float(loss))
    assert rel(g_compat, m2.flat_grads) < 2e-3


def test_dp_simulation_on_the_engine_matches_the_reference(dev, gold):
    """Data-parallel parity on hardware without N GPUs: each 'rank' is one sequential pass of the ENGINE over its shard
    (same weights, own BN batch statistics, own loss normaliser); the mean of the per-shard gradient arenas is what the
    RCCL exchange delivers and must match the fixture the real reference produced shard by shard
    (tests/golden/dp_sim_96.npz; the exchange loop itself runs under gloo in tests/test_dp_gloo_cpu.py)."""
    from computervision.pytorch_amd.train import V8DetectionLoss, flatten_targets
    from configs import Yolo8DetConfig
    f = gold("dp_sim_96.npz")
    x, bi, cls, bb = (torch.from_numpy(f[k]) for k in ("x", "batch_idx", "cls", "bboxes"))
    keys = [str(k) for k in f["keys"]]
    for world in (2, 4):
        per = 8 // world
        acc = None
        for r in range(world):
            m = new_model(dev).train()
            crit = V8DetectionLoss(Yolo8DetConfig(), m)
            sel = (bi >= r * per) & (bi < (r + 1) * per)
            sb = {"batch_idx": bi[sel] - r * per, "cls": cls[sel], "bboxes": bb[sel]}
            pred = m._run_forward(x[r * per:(r + 1) * per].to(dev), training=True)
            _, dpred = crit.op(pred, flatten_targets(sb, dev), m.level_shapes(96, 96), (8, 16, 32), crit.loss_scale)
            m.flat_grads.zero_()
            m.engine_for(96, 96).backward(dpred, crit.loss_scale)
            torch.cuda.synchronize()
            g = m.flat_grads.clone()
            acc = g if acc is None else acc + g
        mean = m.layout.views((acc / world).cpu())
        flat = torch.cat([mean[k].flatten() for k in keys])
        ref = torch.from_numpy(f[f"w{world}_sub"])
        err = float((flat[::211] - ref).norm() / ref.norm())
        print(f"[parity] DP simulation world {world}: gradient rel-L2 vs reference shard mean {err:.3e}")
        assert err < 1.6e-1, (world, err)       # end to end incl. the discrete top-10 assignment on fp16-perturbed logits (1.2e-1 / 2.8e-2 measured)
        assert abs(float(flat.norm()) / float(f[f"w{world}_norm"]) - 1) < 5e-2


def test_yolov8s_against_the_reference_fixture(dev, gold):
    """BASELINE config 3's per-rank model (YOLOv8-s): forward logits, loss and gradient norms at 160x160 against numbers
    captured from the reference itself (oracle/make_golden.py section 8); bit-identical initialisation under the seed."""
    from computervision.pytorch_amd.train import V8DetectionLoss, flatten_targets
    from configs import Yolo8DetConfig
    g = gold("yolov8s_train_160.npz")
    cfg = Yolo8DetConfig()
    cfg.arch.model_type = "s"
    m = new_model(dev, scale="s").train()
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    x = torch.from_numpy(g["x"])
    batch = {"batch_idx": torch.from_numpy(g["batch_idx"]), "cls": torch.from_numpy(g["cls"]), "bboxes": torch.from_numpy(g["bboxes"])}
    crit = V8DetectionLoss(cfg, m)
    pred = m._run_forward(x.to(dev), training=True)
    outs, off = [], 0
    for (h, w) in m.level_shapes(160, 160):
        outs.append(pred[:, off:off + h * w].permute(0, 2, 1).reshape(2, 144, h, w))
        off += h * w
    r = level_report("YOLOv8-s 160x160 fixture vs reference fp32", outs, [torch.from_numpy(g[f"train{i}"]) for i in range(3)])
    assert max(r) < LEVEL_TOL, r
    items, dpred = crit.op(pred, flatten_targets(batch, dev), m.level_shapes(160, 160), (8, 16, 32), crit.loss_scale)
    np.testing.assert_allclose(items.cpu().numpy(), g["items"], rtol=3e-3)
    assert abs(float(items.sum() * 2) / float(g["loss"]) - 1) < 2e-3
    m.flat_grads.zero_()
    m.engine_for(160, 160).backward(dpred, crit.loss_scale)
    m.attach_grads()
    named = dict(m.named_parameters())
    mine = np.array([float(named[str(k)].grad.norm()) for k in g["keys"]])
    ref = g["grad_norms"]
    big = ref > 0.05 * ref.max()                                  # per-tensor norms of the tensors that carry the gradient
    assert np.all(np.abs(mine[big] / ref[big] - 1) < 0.15), float(np.abs(mine[big] / ref[big] - 1).max())
    assert abs(np.linalg.norm(mine) / np.linalg.norm(ref) - 1) < 5e-2
    np.testing.assert_allclose(named["model.22.cv3.0.2.bias"].grad.cpu().numpy(), g["g_headb"], rtol=5e-2, atol=5e-4)


def test_yolov8s_full_size_fused_steps(dev):
    """YOLOv8-s at BASELINE config 3's per-rank size (batch 32, 640x640): fused steps stay finite, the loss goes down,
    the gradient arena is left zeroed, eval is per-image independent."""
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    cfg = Yolo8DetConfig()
    cfg.arch.model_type = "s"
    m = new_model(dev, scale="s").train()
    step = FusedTrainStep(m, V8DetectionLoss(cfg, m), FlatAdam(m, lr=1e-3))
    x, batch = synth.images(32, 640, 640, seed=1).to(dev), synth.targets(32, seed=2)
    losses = [float(step(x, batch).sum()) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert float(m.flat_grads.abs().max()) == 0.0 and bool(torch.isfinite(m.flat_params).all())


# ---- eval tail -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["tv0141_cuda", "tv0141_cpu", "offset", "vanilla"])
def test_nms_bit_exact_against_the_oracle(dev, gold, variant):
    """Every torchvision-0.14.1 batched_nms strategy (oracle/nms_ref.py) -- kept anchor indices AND rows bit-exact -- on the
    clustered fixture (8400 anchors: conf 0.25 -> ~900 candidates = coordinate-offset path of the library's switch,
    conf 0.001 -> 8400 candidates = per-class path) and on the borderline fixture (pairs at IoU = thr +- 1e-6 in high classes,
    where the two strategies keep different sets: tests/test_oracle_golden.py counts them)."""
    g = gold("nms_tail.npz")
    pred = synth.nms_pred(int(g["seed"]))
    for p_, conf, max_det in ((pred, 0.25, 300), (pred, 0.001, 300), (synth.nms_pred_borderline(3), 0.25, 1024)):
        rows, index, counts = E.nms(torch.from_numpy(p_).to(dev), conf, 0.7, max_det, variant=variant)
        ref = nms_ref.non_max_suppression(p_, conf, 0.7, max_det, variant=variant)
        for b in range(p_.shape[0]):
            k = int(counts[b])
            assert k == len(ref[b][1])
            assert np.array_equal(index[b, :k].cpu().numpy().astype(np.int64), ref[b][1])
            assert np.array_equal(rows[b, :k].cpu().numpy(), ref[b][0])          # bit-exact rows
    assert np.array_equal(E.nms(torch.from_numpy(pred).to(dev), 0.25, 0.7, 300)[1][0, :300].cpu().numpy().astype(np.int64), g["keep0"])


def test_nms_edge_cases(dev):
    empty = torch.zeros(2, 84, 8400, device=dev)
    rows, index, counts = E.nms(empty, 0.25, 0.7, 300)
    assert counts.tolist() == [0, 0]
    p = np.zeros((1, 84, 64), np.float32)
    p[0, :4, 3] = p[0, :4, 9] = (100, 100, 50, 50)
    p[0, 4 + 7, 3] = p[0, 4 + 7, 9] = 0.9                                      # identical boxes, tied scores
    rows, index, counts = E.nms(torch.from_numpy(p).to(dev), 0.25, 0.7, 300)
    assert counts.tolist() == [1] and int(index[0, 0]) == 3                     # lower anchor index wins the tie
    p[0, 4 + 7, 9] = 0
    p[0, 4 + 8, 9] = 0.8                                                         # other class: both survive
    rows, index, counts = E.nms(torch.from_numpy(p).to(dev), 0.25, 0.7, 300)
    assert counts.tolist() == [2] and index[0, :2].tolist() == [3, 9]
    with pytest.raises(AssertionError):
        E.nms(torch.from_numpy(p).to(dev), 1.5, 0.7)
    # idempotence at full size: NMS of the survivors keeps all of them
    pred = synth.nms_pred(11, b=1)
    rows, index, counts = E.nms(torch.from_numpy(pred).to(dev), 0.25, 0.7, 300)
    k = int(counts[0])
    sub = pred[:, :, index[0, :k].cpu().numpy()]
    r2, i2, c2 = E.nms(torch.from_numpy(np.ascontiguousarray(sub)).to(dev), 0.25, 0.7, 300)
    assert int(c2[0]) == k and i2[0, :k].tolist() == list(range(k))


@pytest.mark.parametrize("nc,slack", [(80, 0), (80, 4), (3, 0), (20, 3), (91, 0), (200, 0)])
def test_decode_kernel_on_ragged_tiles_and_row_strides(dev, nc, slack):
    """cvx_decode_strided (modules.py:434-446) against the oracle's decode_eval: anchor counts that end inside a 64-anchor tile, tiles that
    straddle two images, class counts off the 16-byte path (3, 91), rows with slack behind them, and a class count whose tile does not
    fit LDS (200: the per-row kernel).  fp32 on identical inputs; the only difference is expf (1e-6 relative on a DFL distance of up to 15
    cells, times a stride of up to 32 pixels): boxes to 1e-3 pixel, class probabilities to 1e-6."""
    g = torch.Generator().manual_seed(nc + slack)
    hw, strides, B = [(5, 7), (3, 4), (2, 2)], (8.0, 16.0, 32.0), 3
    feats = [torch.randn(B, 64 + nc, h, w, generator=g) * 3 for h, w in hw]
    want = O.decode_eval(feats, strides, nc)
    rows = torch.cat([f.reshape(B, 64 + nc, -1) for f in feats], 2).permute(0, 2, 1).contiguous()   # (B, A, no)
    A = rows.shape[1]
    buf = torch.full((B, A, 64 + nc + slack), float("nan"))
    buf[:, :, :64 + nc] = rows
    pred = buf.to(dev)[:, :, :64 + nc]
    got = E.decode(pred, nc, hw, strides).cpu()
    assert got.shape == (B, 4 + nc, A)
    assert torch.isfinite(got).all()
    assert torch.allclose(got[:, :4], want[:, :4], rtol=1e-5, atol=1e-3), float((got[:, :4] - want[:, :4]).abs().max())
    assert torch.allclose(got[:, 4:], want[:, 4:], rtol=0, atol=1e-6), float((got[:, 4:] - want[:, 4:]).abs().max())


def test_decode_box_through_the_plugin_api(dev):
    import builder
    cfg, algo_cls, _ = builder.export_from_registry("yolo8_det")
    algo = algo_cls(cfg, dev)
    pred = synth.nms_pred(5, b=1)
    boxes, conf, cls = algo.decode_box((torch.from_numpy(pred).to(dev), None), 480, 640)
    ref_rows, _ = nms_ref.non_max_suppression(pred, cfg.decode.conf_threshold, cfg.decode.nms_threshold, cfg.decode.max_det)[0]
    rb, rc, rk = nms_ref.decode_box(ref_rows, (640, 640), (480, 640), True)
    np.testing.assert_allclose(boxes, rb, rtol=1e-5, atol=1e-3)
    assert np.array_equal(cls, rk) and np.array_equal(conf, rc)

def test_eval_forward_at_the_bench_dispatch_point_bs32_against_the_oracle(dev):
    """The headline's own shapes: the bs-32 640 x 640 synthetic batch through the EVAL-mode forward (tile plans depend on the batch), images
    0 and 31 against the CPU oracle's eval forward of those two images -- 1e-3 per Detect level (an image's eval-mode outputs do not
    depend on the rest of the batch).  The running statistics are calibrated first (one oracle training forward at momentum 1 on the two
    images): at the constructor's mean 0 / var 0.9 the activations decay layer by layer and every logit is its bias."""
    x = synth.images(32, 640, 640, seed=1)
    sd = O.init_state_dict("n", 80, seed=0)
    old = O.BN_MOMENTUM
    O.BN_MOMENTUM = 1.0
    try:
        with torch.no_grad():
            O.forward(sd, x[[0, 31]], "n", 80, training=True)
    finally:
        O.BN_MOMENTUM = old
    with torch.no_grad():
        _, ref = O.forward(sd, x[[0, 31]], "n", 80, training=False)
    m = new_model(dev)
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        y, feats = m(x.to(dev))
    for lvl in range(3):
        got = feats[lvl][[0, 31]].float().cpu()
        r = rel(got, ref[lvl])
        spread = float(ref[lvl].std())
        print(f"[bs32 eval] level {lvl}: rel {r:.2e} (logit spread {spread:.2f})")
        assert spread > 0.5 and r < LEVEL_TOL, (lvl, r, spread)


def test_full_size_properties_bs32(dev):
    """BASELINE size (bs=32, 640x640): size-independent checks -- finite outputs, loss decreases over fused
    steps on a fixed batch, gradient arena zeroed by the fused Adam, per-image independence of the eval path."""
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    m = new_model(dev).train()
    step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m, lr=1e-3))
    x, batch = synth.images(32, 640, 640, seed=1).to(dev), synth.targets(32, seed=2)
    losses = [float(step(x, batch).sum()) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert float(m.flat_grads.abs().max()) == 0.0
    m.eval()
    with torch.no_grad():
        y_all = m._run_forward(x[:4], training=False)
        y_one = m._run_forward(x[2:3], training=False)
    assert torch.equal(y_all[2:3], y_one)                 # eval BN: images do not interact, bit-identical


def test_overlapped_exchange_is_bit_identical_to_plain_backward(dev):
    """Data-parallel path with one rank (RCCL group of size 1): backward cut into 4 op ranges, each range's arena slice
    folded and all-reduced on a side stream while the next range runs, must give exactly the plain step's parameters."""
    import torch.distributed as dist
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    x, batch = synth.images(2, 128, 128, seed=1).to(dev), {k: v.to(dev) for k, v in synth.targets(2, seed=2).items()}
    created = False
    if not dist.is_initialized():
        try:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1, device_id=dev)
            created = True
        except Exception as exc:                                    # no RCCL on this box: nothing to compare
            pytest.skip(f"RCCL process group unavailable: {exc}")
    try:
        runs = []
        for distributed in (False, True):
            m = new_model(dev).train()
            step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m, lr=1e-3), n_buckets=4)
            step.distributed = distributed
            losses = [step(x, batch).clone() for _ in range(3)]
            torch.cuda.synchronize()
            runs.append((torch.stack(losses).cpu(), m.flat_params.clone().cpu()))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    finally:
        if created:
            dist.destroy_process_group()


_DP_COST_SCRIPT = r"""
import sys, time, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from computervision.pytorch_amd.model import Yolo8
from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
from configs import Yolo8DetConfig
from oracle import synth
dev = torch.device("cuda:0")
x, batch = synth.images(32, 640, 640, seed=1).to(dev), {k: v.to(dev) for k, v in synth.targets(32, seed=2).items()}
try:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1, device_id=dev)
except Exception as exc:
    print("DP_COST_SKIP", exc)
    sys.exit(0)
ms = []
for distributed in (False, True):
    torch.manual_seed(0)
    m = Yolo8("n", 80).to(dev).train()
    step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m, lr=1e-3), n_buckets=5)
    step.distributed = distributed
    for _ in range(4):
        step(x, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step(x, batch)
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) * 100.0)
    del step, m
dist.destroy_process_group()
print(f"DP_COST {ms[0]:.3f} {ms[1]:.3f}")
"""


def test_data_parallel_step_costs_what_the_single_gpu_step_costs(dev):
    """The exchange path of ``bench.py --gpus N`` with a 1-rank RCCL group, at the bench's own shape (batch 32, 640 x 640), against the plain
    step: with the exchange on a stream of its own (a torch pool stream) this was 16.0 against 6.5 ms -- a fifth stream at work beside the
    engine's four -- and is 6.5 against 6.5 with the exchange on the engine's reduction stream.  A 2.45x cliff is
    what this guards against: the bound is 1.3x.
    In a child process with the hardware-queue setting a rank of a multi-GPU job runs under (WORLD_SIZE > 1: neither the package nor
    bench.py touches GPU_MAX_HW_QUEUES, the runtime's default of 4 stays).  Under the SINGLE-process default of one queue per priority the
    same path measured 11.4 against 6.5 ms (round 4): ProcessGroupNCCL's stream shares the main stream's priority, so the collective and
    its wait for the weight gradients sit in the main chain's in-order queue -- the reason that default is not applied to ranks."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="4")
    r = subprocess.run([sys.executable, "-c", _DP_COST_SCRIPT, root], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if "DP_COST_SKIP" in r.stdout:
        pytest.skip("RCCL process group unavailable: " + r.stdout[-300:])
    ms = [float(v) for v in r.stdout.split("DP_COST")[1].split()[:2]]
    print(f"[dp] plain step {ms[0]:.2f} ms, 1-rank RCCL step {ms[1]:.2f} ms")
    assert ms[1] < 1.3 * ms[0], ms


@pytest.mark.parametrize("scale,B,H,W", [("n", 4, 160, 160), ("n", 2, 96, 224), ("s", 2, 128, 128)])
def test_per_layer_backward_against_fp64_on_the_engines_own_operands(dev, scale, B, H, W):
    """Layer by layer through the whole backward pass (cvx_engine_debug_copy: every BN conv's xhat, its completed output
    gradient g and the gradient dy it hands to the data- / weight-gradient kernels): the BatchNorm+SiLU backward of
    modules.py:29-30 in fp64 on those operands --  dz = g * silu'(gamma*xhat+beta),  dgamma = sum dz*xhat,
    dbeta = sum dz,  dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat))  -- must give the engine's dgamma / dbeta
    (2e-6: fp32 block partials, exact fixed-point totals) and its fp16 dy (one rounding).  End-to-end gradient
    comparisons cannot be this tight: a last-bit change of one layer's sums flips fp16 roundings of dy, and sums of
    sign-alternating terms amplify that from layer to layer (1.5e-3 over the network, measured)."""
    from computervision.pytorch_amd.train import V8DetectionLoss
    from configs import Yolo8DetConfig
    x, batch = synth.images(B, H, W, seed=5).to(dev), {k: v.to(dev) for k, v in synth.targets(B, seed=6).items()}
    m = new_model(dev, scale=scale).train()
    crit = V8DetectionLoss(Yolo8DetConfig(), m)
    eng = m.engine_for(H, W)
    m.flat_grads.zero_()
    loss, _ = crit(m(x), batch)
    loss.backward()
    torch.cuda.synchronize()
    checked = 0
    for i, o in enumerate(eng.graph.ops):
        if o["type"] != L.OP_CONV or o.get("act", 0) != L.ACT_BN_SILU or o["name"] == "0":
            continue                                      # (the stem keeps no fp16 dy: stem_bwd forms it in registers)
        cs = m.layout.convs[o["name"]]
        C = cs.cout_eng
        xh = eng.read_layer(i, B, "xhat32").double().reshape(-1, C)     # (as the backward passes use it: round 5's raw-fp16 layers normalise on the fly)
        dy = eng.read_layer(i, B, "dy").double().reshape(-1, C)
        ob = o["out"]
        gout = eng.read_buffer(ob[0], B, grad=True).double().reshape(-1, eng.graph.bufs[ob[0]][2])[:, ob[1]:ob[1] + ob[2]]
        ga, be = m.flat_params[cs.gamma_off:cs.gamma_off + C].double(), m.flat_params[cs.beta_off:cs.beta_off + C].double()
        z = xh * ga + be
        sg = torch.sigmoid(z)
        dz = gout * (sg * (1 + z * (1 - sg)))
        want_g, want_b = (dz * xh).sum(0) / crit.loss_scale, dz.sum(0) / crit.loss_scale
        got_g, got_b = m.flat_grads[cs.gamma_off:cs.gamma_off + C].double(), m.flat_grads[cs.beta_off:cs.beta_off + C].double()
        if float(want_g.norm()) == 0.0:
            continue
        assert rel(got_g, want_g) < 2e-6 and rel(got_b, want_b) < 2e-6, (o["name"], rel(got_g, want_g), rel(got_b, want_b))
        # invstd from the running-variance update is not exposed; recover gamma*invstd per channel from dy itself (least squares)
        core = dz - dz.mean(0) - xh * (dz * xh).mean(0)
        gi = (dy * core).sum(0) / (core * core).sum(0).clamp_min(1e-300)
        # fp16 dy: half an ulp relative (5e-4) for normal values, half the subnormal spacing (3e-8) absolute below 6e-5 --
        # the head's gradients at loss scale 1024 are that small
        rms = float(dy.pow(2).mean().sqrt())
        assert rel(dy, core * gi) < 1e-3 + 6e-8 / max(rms, 1e-30), (o["name"], rel(dy, core * gi), rms)
        checked += 1
    assert checked >= 50


def test_dynamic_loss_scale_skips_overflow_and_backs_off(dev):
    """GradScaler semantics of the reference's mixed-precision loop (yolo8_train.py:99-104): a step with non-finite
    gradients leaves parameters and Adam state untouched and halves the scale; finite steps at equal scale are
    bit-identical to the static-scale path."""
    from computervision.pytorch_amd.train import DynamicLossScale, FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    x, batch = synth.images(2, 128, 128, seed=1).to(dev), {k: v.to(dev) for k, v in synth.targets(2, seed=2).items()}
    # (a) same scale, dynamic vs static: identical
    runs = []
    for dynamic in (False, True):
        m = new_model(dev).train()
        crit = V8DetectionLoss(Yolo8DetConfig(), m)
        sc = DynamicLossScale(dev, init_scale=crit.loss_scale, growth_interval=1000) if dynamic else None
        step = FusedTrainStep(m, crit, FlatAdam(m, lr=1e-3), scaler=sc)
        for _ in range(3):
            step(x, batch)
        torch.cuda.synchronize()
        runs.append(m.flat_params.clone().cpu())
    assert torch.equal(runs[0], runs[1])
    # (b) absurd scale: fp16 gradients overflow -> skipped steps, scale backs off until a step goes through
    m = new_model(dev).train()
    sc = DynamicLossScale(dev, init_scale=2.0 ** 40, growth_interval=1000)
    opt = FlatAdam(m, lr=1e-3)
    step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), opt, scaler=sc)
    p0 = m.flat_params.clone()
    step(x, batch)
    torch.cuda.synchronize()
    assert torch.equal(m.flat_params, p0)                          # skipped on the device
    sc.poll()
    assert sc.scale == 2.0 ** 39 and sc.skipped == 1
    for _ in range(60):
        step(x, batch)
        torch.cuda.synchronize()
    sc.poll()
    assert sc.scale < 2.0 ** 30 and not torch.equal(m.flat_params, p0)
    assert bool(torch.isfinite(m.flat_params).all()) and bool(torch.isfinite(opt._m).all())


@pytest.mark.parametrize("scale", ["m", "x"])
def test_wider_scales_train_through_the_engine(dev, scale):
    """YOLOv8-m/-x reach 576/640-channel layers (BatchNorm passes up to 1024 channels, kernels fall back where a fast path
    has no instantiation): a few fused steps must run, stay finite and reduce the loss."""
    from computervision.pytorch_amd.model import Yolo8
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    cfg = Yolo8DetConfig()
    cfg.arch.model_type = scale
    torch.manual_seed(0)
    m = Yolo8(scale, 80, loss_scale=1024.0).to(dev).train()
    step = FusedTrainStep(m, V8DetectionLoss(cfg, m), FlatAdam(m, lr=1e-3))
    x, batch = synth.images(2, 128, 128, seed=1).to(dev), {k: v.to(dev) for k, v in synth.targets(2, seed=2).items()}
    losses = [float(step(x, batch).sum()) for _ in range(6)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and bool(torch.isfinite(m.flat_params).all())
    assert min(losses[3:]) < losses[0]


@pytest.mark.parametrize("nc", [20, 3, 91])
def test_other_class_counts_match_the_oracle(dev, nc):
    """VOC (20 classes, configs/dataset_cfg.py), a count that is not even a multiple of 4, and 91 (COCO category ids: wider than the
    64-channel hidden layers of the class branch, so those are zero-padded to 96 as well): the engine pads the class
    columns of pred to a multiple of 8 (zero weight rows); forward, loss value and one fused step must follow the oracle."""
    from computervision.pytorch_amd.model import Yolo8
    from computervision.pytorch_amd.train import FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    cfg = Yolo8DetConfig()
    cfg.dataset.num_classes = nc
    x = synth.images(2, 128, 128, seed=3)
    b = synth.targets(2, seed=4)
    b["cls"] = b["cls"] % nc
    torch.manual_seed(0)
    m = Yolo8("n", nc, loss_scale=1024.0).to(dev).train()
    sd = O.init_state_dict("n", nc, seed=0)
    for k, v in m.state_dict().items():                      # same seed, same draw order: identical initialisation
        assert torch.equal(v.cpu(), sd[k]), k
    with torch.no_grad():
        outs = m(x.to(dev))
    assert [tuple(o.shape) for o in outs] == [(2, 64 + nc, 16, 16), (2, 64 + nc, 8, 8), (2, 64 + nc, 4, 4)]
    with fp16_storage():
        ref = O.forward(sd, x, "n", nc, training=True)
    for o, r in zip(outs, ref):
        # per part: the box logits are O(1) and carry the fp16-storage noise of 30 layers (3-5e-3, the same for every class
        # count -- and as far from the fp32 oracle as the emulation itself); the class logits (bias ~ -7) are tight
        assert rel(o[:, :64], r[:, :64].detach()) < 8e-3 and rel(o[:, 64:], r[:, 64:].detach()) < 2e-3
    # loss on the oracle's own head outputs: value parity (same inputs to both)
    crit = V8DetectionLoss(cfg, m)
    feats = [r.detach().to(dev) for r in ref]
    loss, items = crit(feats, {k: v.to(dev) for k, v in b.items()})
    _, items_ref = O.v8_loss([r.detach() for r in ref], b, nc)
    np.testing.assert_allclose(items.cpu().numpy(), items_ref.numpy(), rtol=5e-5)
    # fused steps: finite, loss goes down, eval path decodes to (B, 4 + nc, A)
    m2 = Yolo8("n", nc, loss_scale=1024.0).to(dev).train()
    step = FusedTrainStep(m2, V8DetectionLoss(cfg, m2), FlatAdam(m2, lr=1e-3))
    bd = {k: v.to(dev) for k, v in b.items()}
    losses = [float(step(x.to(dev), bd).sum()) for _ in range(6)]
    assert all(np.isfinite(losses)) and min(losses[3:]) < losses[0]
    m2.eval()
    with torch.no_grad():
        y, _ = m2(x.to(dev))
    assert tuple(y.shape) == (2, 4 + nc, 336) and bool(torch.isfinite(y).all())


# ---- CenterNet DLA-34 inference + heat-map decode (BASELINE config 4) ------------------------------------------
def _centernet(dev, nc=80):
    from computervision.pytorch_amd.dla import CenterNetDLA34
    torch.manual_seed(0)
    return CenterNetDLA34(nc).to(dev).eval()


def test_centernet_state_dict_is_the_references(dev):
    """Same keys, order and shapes as the reference's CenterNet(cfg).state_dict(), and -- under the same seed -- the same
    values bit for bit (oracle.init_state_dict is asserted equal to the reference's in oracle/make_golden.py section 9)."""
    from oracle import centernet_ref as C
    m = _centernet(dev)
    ref = C.init_state_dict(80, seed=0)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 326
    for k, v in ref.items():
        assert sd[k].shape == v.shape and torch.equal(sd[k].cpu(), v), k
    assert sum(p.numel() for p in m.parameters()) == 18474476


def test_centernet_forward_matches_the_oracle(dev, gold):
    """DLA-34 eval forward on the engine vs the fp32 oracle (pinned to the reference), with BatchNorm running statistics that
    normalise (calibrated on the batch: at the constructor's mean 0 / var 1 a 40-layer ReLU chain decays to nothing and every
    output is its head bias).  Per head: relative L2 of the logits / regressions <= 3e-3 (fp16 weights and activations,
    fp32 accumulation; the 7x7 stem reads the image as fp16)."""
    from oracle import centernet_ref as C
    g = gold("centernet_fwd_128.npz")
    x = torch.from_numpy(g["x"])
    sd = C.init_state_dict(80, seed=0)
    old = C.BN_MOMENTUM
    C.BN_MOMENTUM = 1.0                                               # running statistics := this batch's statistics
    try:
        C.forward(sd, x, 80, training=True)
    finally:
        C.BN_MOMENTUM = old
    with torch.no_grad():
        ref = C.forward(sd, x, 80, training=False)
    m = _centernet(dev)
    m.load_state_dict(sd)
    with torch.no_grad():
        out = m(x.to(dev)).cpu()
    assert tuple(out.shape) == (2, 32, 32, 84)
    parts = {"heatmap": slice(0, 80), "wh": slice(80, 82), "reg": slice(82, 84)}
    C.FP16_STORAGE[0] = True
    try:
        with torch.no_grad():
            emu = C.forward(sd, x, 80, training=False)
    finally:
        C.FP16_STORAGE[0] = False
    errs = {k: rel(out[..., sl], ref[..., sl]) for k, sl in parts.items()}
    errs_emu = {k: rel(out[..., sl], emu[..., sl]) for k, sl in parts.items()}
    print("[parity] CenterNet DLA-34 128x128 eval vs oracle fp32: " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()) +
          " | vs fp16-storage emulation: " + " ".join(f"{k} {v:.2e}" for k, v in errs_emu.items()))
    # same arithmetic (fp16 conv operands, fp32 accumulation) emulated on the CPU: only the summation order differs, and a
    # rounding that flips propagates like the rounding itself (3.8e-3 / 4.5e-3 / 4.8e-3 measured on this random-init network,
    # whose calibrated BatchNorms have folded scales up to 6)
    assert max(errs_emu.values()) < 8e-3, errs_emu
    # against fp32: what fp16 operands cost on this 50-conv ReLU network at random init (the emulation itself is 1.1e-2 /
    # 1.3e-2 / 1.4e-2 away from fp32 on this input)
    assert max(errs.values()) < 2.5e-2, errs
    # the reference's own eval output at the constructor's statistics (fixture): bias-dominated, must agree as well
    m2 = _centernet(dev)
    sd0 = C.init_state_dict(80, seed=0)
    C.forward(sd0, x, 80, training=True)
    m2.load_state_dict(sd0)
    with torch.no_grad():
        out0 = m2(x.to(dev)).cpu()
    assert rel(out0.flatten()[::7], torch.from_numpy(g["eval_sub"])) < 3e-3         # bias-dominated outputs


@pytest.mark.parametrize("tag", ["synth", "net"])
def test_centernet_decode_matches_the_reference_fixture(dev, gold, tag):
    """cvx_centernet_decode on head tensors whose decode the REFERENCE itself produced (fixture): classes exact, scores
    bit-exact, boxes to fp32 round-off, survivors of the DIoU-NMS identical.  'synth' has real peaks; 'net' is a random-init
    network output (scores ~0.5 everywhere, 100 of 100 kept)."""
    import builder
    g = gold("centernet_fwd_128.npz")
    cfg, algo_cls, _ = builder.export_from_registry("centernet")
    cfg.dataset.num_classes = 80
    cfg.arch.input_size = (3, 128, 128)
    algo = algo_cls(cfg, dev)
    pred = torch.from_numpy(g[tag + "_pred"]).to(dev)
    h, w = (int(v) for v in g[tag + "_hw"])
    boxes, scores, classes = algo.decode_boxes(pred, h, w)
    assert np.array_equal(classes, g[tag + "_classes"])              # same peaks, same order, same DIoU-NMS survivors
    # scores: the reference's torch.sigmoid and the device's 1 / (1 + expf(-x)) agree to the last bit on 98 of the 100 'net'
    # scores and on all 'synth' ones; the rest differ by one unit in the last place of expf
    ulp = np.abs(scores.view(np.int32).astype(np.int64) - g[tag + "_scores"].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp == 0).mean() >= 0.9, (int(ulp.max()), float((ulp == 0).mean()))
    np.testing.assert_allclose(boxes, g[tag + "_boxes"], rtol=1e-6, atol=1e-5)


def test_centernet_full_size_properties(dev):
    """BASELINE config 4's size (512x512; batch 8 here, 64 in bench.py): finite outputs, per-image independence of forward and
    decode (image b of a batch == the same image alone, bit for bit: eval BatchNorm, per-image decode), sparse heat-maps."""
    import builder
    m = _centernet(dev)
    x = synth.images(8, 512, 512, seed=3).to(dev)
    with torch.no_grad():
        raw = m.forward_raw(x)
        one = m.forward_raw(x[5:6])
    assert bool(torch.isfinite(raw).all()) and torch.equal(raw[5:6], one)
    cfg, algo_cls, _ = builder.export_from_registry("centernet")
    cfg.dataset.num_classes = 80
    cfg.arch.input_size = (3, 512, 512)
    algo = algo_cls(cfg, dev)
    d_all, d_one = algo.decode_raw(raw, 128, 128), algo.decode_raw(one, 128, 128)
    n = int(d_one["counts"][0])
    assert n == int(d_all["counts"][5]) and n >= 0
    assert torch.equal(d_all["topk_index"][5], d_one["topk_index"][0]) and torch.equal(d_all["keep"][5, :n], d_one["keep"][0, :n])
    # a heat-map with 7 peaks only: the top-100 list is cut short, exactly those 7 come back
    sparse = torch.full((1, 128 * 128, 96), -200.0, device=dev)      # sigmoid(-200) == 0 exactly in fp32
    sparse[..., 80:] = 0.5
    peaks = [(3, 5, 7), (100, 64, 0), (127, 127, 79), (64, 64, 40), (10, 120, 33), (90, 2, 1), (50, 77, 60)]
    for y, x_, c in peaks:
        sparse[0, y * 128 + x_, c] = 2.0 + 0.01 * c
    d = algo.decode_raw(sparse, 128, 128)
    assert int(d["counts"][0]) == 7 and sorted(d["topk_index"][0, :7].tolist()) == sorted((y * 128 + x_) * 80 + c for y, x_, c in peaks)
    assert d["topk_index"][0, 7:].tolist() == [-1] * 93
    # logits spread over several octaves of score (0.007 ... 0.995: every digit of the radix select sees many occupied bins, the lanes of
    # a wave land in different bins; the top stays below the saturated range where fp32 sigmoids tie) against torch: sigmoid, the
    # reference's 3 x 3 pool over (x, class), top-100 of the surviving peaks in descending order
    gq = torch.Generator().manual_seed(11)
    wide = torch.zeros(2, 128 * 128, 96)
    wide[..., :80] = torch.randn(2, 128 * 128, 80, generator=gq)
    dw = algo.decode_raw(wide.to(dev), 128, 128)
    heat = torch.sigmoid(wide[..., :80]).reshape(2, 128, 128, 80)
    peak = heat * (F.max_pool2d(heat, 3, 1, 1) == heat)                      # (B, H, W, C) pooled as if it were NCHW: over (x, class)
    want = torch.topk(peak.reshape(2, -1), 100)
    assert int((want.values[:, 1:] == want.values[:, :-1]).sum()) == 0       # no ties in this draw: the order is unique
    assert torch.equal(dw["topk_index"].cpu().long(), want.indices)


# ---- DeepLabv3+ ResNet-101, inference --------------------------------------------------------------------
def _deeplab(dev, gold_file=None):
    from computervision.pytorch_amd.deeplab import DeepLabV3PlusR101
    torch.manual_seed(0)
    m = DeepLabV3PlusR101(21)
    if gold_file is not None:                              # the calibrated running statistics of the fixture run
        sd = m.state_dict()
        keys, vals, off = [str(k) for k in gold_file["stat_keys"]], gold_file["stat_vals"], 0
        with torch.no_grad():
            for k in sd:                                   # residual branches scaled down, as in the fixture run (make_golden.py, section 10)
                if k.endswith(".bn3.weight"):
                    sd[k].fill_(float(gold_file["bn3_gamma"]))
            for k in keys:
                n = sd[k].numel()
                sd[k].copy_(torch.from_numpy(vals[off:off + n].copy()))
                off += n
    return m.to(dev).eval()


def test_deeplab_state_dict_is_the_references(dev):
    """674 keys in the reference's order, seed-0 values bit-identical to DeeplabV3Plus(21, 16, pretrained_backbone=False)
    (resnet.py:150-178, deeplabv3plus.py:99-110) -- the oracle's init is pinned to the reference's in make_golden.py."""
    from oracle import deeplab_ref as D
    m = _deeplab(dev)
    ref = D.init_state_dict(21, seed=0)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 674
    for k, v in ref.items():
        assert sd[k].shape == v.shape and torch.equal(sd[k].cpu(), v), k


def test_deeplab_forward_matches_the_reference_fixture(dev, gold):
    """Eval forward of the calibrated random-init network on the fixture batch (2 x 3 x 193 x 225, odd sizes as the
    reference's 513): logits at the decoder's resolution and the final NCHW tensor against the REAL reference's outputs.
    104 convolutions with fp16 operands in a row: the yardstick is what the fp16-operand emulation of the oracle gives on the
    CPU against the reference (printed); the engine must be within 1.5x of it."""
    from oracle import deeplab_ref as D
    g = gold("deeplab_fwd_193x225.npz")
    m = _deeplab(dev, g)
    x = torch.from_numpy(g["x"]).to(dev)
    out = m(x)
    rows = m.last_rows[..., :21].reshape(2, 49, 57, 21).float().cpu()
    ref_rows = torch.from_numpy(g["rows"])
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    D.FP16_STORAGE[0] = True
    try:
        with torch.no_grad():
            emu_out, emu_rows = D.forward(sd, x.cpu(), 21, return_rows=True)
    finally:
        D.FP16_STORAGE[0] = False
    e_ref, e_emu, emu_ref = rel(rows, ref_rows), rel(rows, emu_rows), rel(emu_rows, ref_rows)
    print(f"deeplab rows: engine vs reference {e_ref:.3e}, engine vs fp16 emulation {e_emu:.3e}, emulation vs reference {emu_ref:.3e}")
    assert e_emu < 1.5e-2 and e_ref < max(1.5 * emu_ref, 1e-2)
    assert tuple(out.shape) == (2, 21, 193, 225)
    assert rel(out.flatten()[::11].cpu(), torch.from_numpy(g["out_sub"])) < max(1.5 * emu_ref, 1e-2)
    # argmax agreement (what the plugin's post-processing consumes, segmentation_2d.py:28)
    agree = (out.argmax(1).cpu() == emu_out.argmax(1)).float().mean().item()
    assert agree > 0.97, agree


def test_deeplab_full_size_through_the_plugin_api(dev):
    """export_from_registry("deeplabv3plus") at the reference's 513 x 513 (odd sizes all the way down: 257, 129, 65, 33):
    logits shape, finiteness, determinism, the colour post-processing (segmentation_2d.py:20-30) and the training guard."""
    import builder
    from computervision.pytorch_amd import _lib as LL
    cfg, algo_cls, trainer_cls = builder.export_from_registry("deeplabv3plus")
    algo = algo_cls(cfg, dev)
    torch.manual_seed(0)
    model, name = algo.build_model()
    assert name == "deeplabv3plus"
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith(".bn3.weight"):
                v.fill_(0.1)
    model = model.to(dev).eval()
    x = synth.images(2, 513, 513, seed=3).to(dev)
    out = model(x)
    assert tuple(out.shape) == (2, 21, 513, 513) and out.dtype == torch.float32 and torch.isfinite(out).all()
    assert torch.equal(out, model(x))
    col = algo.predict_tensor(model, x)
    assert tuple(col.shape) == (2, 513, 513, 3) and int(col.max()) <= 192
    assert type(algo.build_loss()).__name__ == "SegLoss"
    cfg.loss.loss_type = "dice"
    with pytest.raises(LL.CvxError):
        algo_cls(cfg, dev).build_loss()


def test_deeplab_trainer_at_full_size(dev):
    """export_from_registry("deeplabv3plus") -> DeeplabV3PlusTrainer at the reference's 513 x 513 (BASELINE configs[5]; batch 4
    here to keep the test short): a few fused steps of the reference's train_loop (segmentation_trainer.py:114-131) with dropout
    active and GradScaler's initial scale; losses finite and falling on a repeated batch, every parameter tensor moved, BatchNorm
    statistics updated, no overflow skip; then evaluate_loop's metrics (:133-159)."""
    import builder
    from core.trainer.segmentation_trainer import SyntheticSegmentationLoader
    cfg, _, trainer_cls = builder.export_from_registry("deeplabv3plus")
    cfg.train.batch_size = 4
    torch.manual_seed(0)
    loader = SyntheticSegmentationLoader(4, (513, 513), 21, length=2, seed=3)
    tr = trainer_cls(cfg, dev, dataloader=loader)
    assert tr.model.dropout_p == 0.1 and tr.model.loss_scale == 65536.0
    p0 = tr.model.flat_params.clone()
    rv0 = tr.model.flat_stats.clone()
    batch = next(iter(loader))
    tr.model.train()
    losses = [float(tr.train_loop(batch, None)[0]) for _ in range(6)]
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in losses), losses
    assert losses[-1] < losses[0], losses
    assert not tr.criterion.bad_targets()
    tr._step.scaler.poll()
    assert tr._step.scaler.skipped == 0 and tr.optimizer.device_step() == 6
    moved = [k for k, v in tr.model.layout.views(tr.model.flat_params - p0).items() if float(v.abs().max()) == 0.0 and "classifier.3" not in k]
    assert not moved, moved[:5]
    assert not torch.equal(tr.model.flat_stats, rv0) and int(tr.model.state_dict()["backbone.bn1.num_batches_tracked"]) == 6
    ev = tr.evaluate_loop()
    assert set(ev) == {"Loss", "Overall Acc", "Mean Acc", "FreqW Acc", "Mean IoU"} and np.isfinite(ev["Loss"]) and 0.0 <= ev["Mean IoU"] <= 1.0


def test_deeplab_torch_side_loss_reaches_the_same_gradients(dev, gold):
    """The drop-in contract of the reference's loop: ``preds = model(images)`` is an ordinary (B, 21, H, W) tensor, and ANY torch
    loss on it (here the reference's FocalLoss formula, focal_loss.py:14-22, written with torch ops) back-propagates into the engine
    through the adjoint of the final resize.  Same parameter gradients as the fused SegLoss path, to fp16 rounding of the
    logits gradient (the two paths round dLoss/drows to fp16 at the same point); dropout masks repeat with the seed."""
    from computervision.pytorch_amd.deeplab import SegLoss
    g = gold("deeplab_train_97x129.npz")
    x, t = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["target"].astype(np.int64)).to(dev)
    grads = []
    for path in ("fused", "torch"):
        m = _deeplab_train_model(dev, g, dropout_p=0.1)
        m.seed = 11
        out = m(x)
        if path == "fused":
            loss = SegLoss("focal")(out, t)
        else:
            ce = F.cross_entropy(out, t, ignore_index=-100, reduction="none")
            loss = (0.25 * (1 - torch.exp(-ce)) ** 2 * ce).mean()
        loss.backward()
        grads.append((float(loss.detach()), m.flat_grads.clone(), m.flat_stats.clone()))
    assert abs(grads[0][0] - grads[1][0]) < 1e-5 * grads[0][0]
    assert torch.equal(grads[0][2], grads[1][2])                       # identical forward passes (same dropout mask)
    assert rel(grads[1][1], grads[0][1]) < 5e-3, rel(grads[1][1], grads[0][1])


# ---- YOLOv7-l, inference + decode + NMS -----------------------------------------------------------
def _yolov7(dev, g=None):
    from computervision.pytorch_amd.yolov7 import Yolo7L
    torch.manual_seed(0)
    m = Yolo7L(20)
    if g is not None:
        sd = m.state_dict()
        with torch.no_grad():
            for i, h in enumerate(("yolo_head_P3", "yolo_head_P4", "yolo_head_P5")):
                sd[h + ".bias"].copy_(torch.from_numpy(g["head_bias"][i].copy()))
            keys, vals, off = [str(k) for k in g["stat_keys"]], g["stat_vals"], 0
            for k in keys:
                n = sd[k].numel()
                sd[k].copy_(torch.from_numpy(vals[off:off + n].copy()))
                off += n
    return m.to(dev).eval()


def test_yolov7_state_dict_is_the_references(dev):
    """558 keys in the reference's order, seed-0 values bit-identical to Yolo7(cfg) (yolov7_model.py:355-458)."""
    from oracle import yolov7_ref as Y
    m = _yolov7(dev)
    ref = Y.init_state_dict(20, seed=0)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 558
    for k, v in ref.items():
        assert sd[k].shape == v.shape and torch.equal(sd[k].cpu(), v), k


def test_yolov7_forward_decode_nms_match_the_reference_fixture(dev, gold):
    """Eval forward of the calibrated network on the fixture batch against the REAL reference's three outputs (within 1.5x of
    what fp16 operands cost the reference's own arithmetic, measured with the oracle's emulation); the anchor
    decode (cvx_yolo7_decode) of those engine rows against the oracle's decode of the same rows (1e-6: expf); the per-class
    NMS (cvx_nms_variant VANILLA on objectness * class scores) against the oracle's on the same decoded tensor: kept rows,
    order (class ascending, score descending) and the 7-column rows exactly."""
    import builder
    from oracle import yolov7_ref as Y
    g = gold("yolov7_fwd_160x224.npz")
    m = _yolov7(dev, g)
    x = torch.from_numpy(g["x"]).to(dev)
    outs = m(x)
    refs = [torch.from_numpy(g["out0"]), torch.from_numpy(g["out1"])]
    errs = [rel(outs[i].cpu(), refs[i]) for i in range(2)] + [rel(outs[2].flatten()[::5].cpu(), torch.from_numpy(g["out2_sub"]))]
    # yardstick: the oracle with fp16 conv operands / stored activations on the CPU.  At N(0, 0.02) init with calibrated
    # BatchNorms this network turns one fp16 rounding into 3-5 % on the logits in the reference's own arithmetic
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    Y.FP16_STORAGE[0] = True
    try:
        with torch.no_grad():
            emu = Y.forward(sd, x.cpu())
    finally:
        Y.FP16_STORAGE[0] = False
    e_emu = [rel(outs[i].cpu(), emu[i]) for i in range(3)]
    emu_ref = [rel(emu[i], refs[i]) for i in range(2)] + [rel(emu[2].flatten()[::5], torch.from_numpy(g["out2_sub"]))]
    print("yolov7 logits (P5, P4, P3): engine vs reference", " ".join(f"{e:.3e}" for e in errs), "| engine vs fp16 emulation",
          " ".join(f"{e:.3e}" for e in e_emu), "| emulation vs reference", " ".join(f"{e:.3e}" for e in emu_ref))
    assert all(errs[i] < 1.5 * emu_ref[i] for i in range(3)) and max(e_emu) < 0.7 * max(emu_ref), (errs, e_emu, emu_ref)
    assert [tuple(o.shape) for o in outs] == [(2, 75, 5, 7), (2, 75, 10, 14), (2, 75, 20, 28)]
    cfg, algo_cls, _ = builder.export_from_registry("yolo7")
    algo = algo_cls(cfg, dev)
    algo.input_image_size = [160, 224]
    dec, y = algo.decode_rows(m, m.last_rows)
    want = Y.decode(tuple(o.cpu() for o in outs), 20, (160, 224))
    assert tuple(dec.shape) == tuple(want.shape) == (2, 3 * 735, 25)
    np.testing.assert_allclose(dec.cpu().numpy(), want.numpy(), rtol=2e-6, atol=1e-7)
    oracle = Y.nms(dec.cpu(), 20, float(g["conf"]), float(g["nms_thr"]))
    got = algo.nms_device(y, dec, float(g["conf"]))
    for b in range(2):
        rows, keep = oracle[b]
        det, idx = got[b]
        assert np.array_equal(idx.cpu().numpy(), keep), b
        np.testing.assert_allclose(det.cpu().numpy(), rows, rtol=1e-6, atol=1e-7)
        assert len(keep) > 100
    # the reference-shaped entry point: letterbox inverse to a 120 x 200 image
    res = algo.decode_box(outs, 120, 200, model=m)
    assert len(res) == 2 and res[0].shape[1] == 7 and np.isfinite(res[0]).all()


def test_yolov7_nms_returns_more_than_the_first_row_block(dev):
    """The reference's _nms keeps every surviving box (yolo_v7.py:348-415); the device tail first asks cvx_nms for 1024 rows per image and
    retries with room for more when a block comes back full: 3000 disjoint boxes of one class all survive and come back in score order;
    the same holds for the oracle on the same input."""
    import builder
    from oracle import yolov7_ref as Y
    cfg, algo_cls, _ = builder.export_from_registry("yolo7")
    algo = algo_cls(cfg, dev)
    nc, A = 20, 3000
    g = torch.Generator().manual_seed(5)
    dec = torch.zeros(1, A, 5 + nc)
    ii = torch.arange(A)
    dec[0, :, 0] = (ii % 60).float() / 60 + 0.004        # 60 x 50 grid of disjoint boxes (centre x, centre y, w, h), normalised
    dec[0, :, 1] = (ii // 60).float() / 50 + 0.004
    dec[0, :, 2] = 0.006
    dec[0, :, 3] = 0.006
    dec[0, :, 4] = 0.5 + 0.5 * torch.rand(A, generator=g)  # objectness
    dec[0, :, 5 + 3] = 0.9                                 # one class
    dec = dec.to(dev)
    # the NMS input the decode kernel would have produced from these rows: (B, 4 + nc, A), score = objectness * class probability
    y = torch.cat((dec[:, :, :4], dec[:, :, 4:5] * dec[:, :, 5:]), 2).permute(0, 2, 1).contiguous()
    det, idx = algo.nms_device(y, dec, 0.3)[0]
    assert det.shape == (A, 7) and idx.shape == (A,)
    rows, keep = Y.nms(dec.cpu(), nc, 0.3, algo.nms_threshold)[0]
    assert len(keep) == A and np.array_equal(np.sort(idx.cpu().numpy()), np.sort(keep))
    assert bool((det[:-1, 4] * det[:-1, 5] >= det[1:, 4] * det[1:, 5] - 1e-7).all())  # one class: scores descending


def test_yolov7_full_size_through_the_plugin_api(dev):
    """export_from_registry("yolo7") at 640 x 640: output shapes (coarsest level first), determinism, decode of 25200 anchors
    (beyond the 16384 the NMS kernel's sort holds -- allowed, candidates are what counts), the guards."""
    import builder
    from computervision.pytorch_amd import _lib as LL
    cfg, algo_cls, trainer_cls = builder.export_from_registry("yolo7")
    cfg.train.pretrained = False
    algo = algo_cls(cfg, dev)
    torch.manual_seed(0)
    model, name = algo.build_model()
    assert name == "YOLOv7"
    model = model.to(dev).eval()
    x = synth.images(2, 640, 640, seed=4).to(dev)
    outs = model(x)
    assert [tuple(o.shape) for o in outs] == [(2, 75, 20, 20), (2, 75, 40, 40), (2, 75, 80, 80)] and all(torch.isfinite(o).all() for o in outs)
    again = model(x)
    assert all(torch.equal(a, b) for a, b in zip(outs, again))
    dec, y = algo.decode_rows(model, model.last_rows)
    assert tuple(dec.shape) == (2, 25200, 25) and tuple(y.shape) == (2, 24, 25200)
    assert float(dec[..., :2].min()) > -0.1 and float(dec[..., :2].max()) < 1.1 and float(dec[..., 4:].min()) >= 0 and float(dec[..., 4:].max()) <= 1
    res = algo.predict_tensor(model, x, 480, 640, conf_threshold=0.9)
    assert len(res) == 2
    assert type(algo.build_loss()).__name__ == "Yolo7Loss" and trainer_cls.__name__ == "Yolo7Trainer"


# ---- SSD300 VGG16-BN, inference + decode -----------------------------------------------------------------
def _ssd(dev, g=None):
    from computervision.pytorch_amd.ssd import SSD300VGG
    torch.manual_seed(0)
    m = SSD300VGG(20)
    if g is not None:
        sd = m.state_dict()
        keys, vals, off = [str(k) for k in g["stat_keys"]], g["stat_vals"], 0
        with torch.no_grad():
            for k in keys:
                n = sd[k].numel()
                sd[k].copy_(torch.from_numpy(vals[off:off + n].copy()))
                off += n
    return m.to(dev).eval()


def test_ssd_state_dict_is_the_references(dev):
    from oracle import ssd_ref as SS
    m = _ssd(dev)
    ref = SS.init_state_dict(20, seed=0)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 136
    for k, v in ref.items():
        assert sd[k].shape == v.shape and torch.equal(sd[k].cpu(), v), k


def test_ssd_forward_and_decode_match_the_reference_fixture(dev, gold):
    """Eval forward of the calibrated VGG16-BN SSD (conv bias folded with the BatchNorm, ceil-mode pool, dilated conv6,
    L2Normalize, activation-free extras, NCHW-order flattening) against the REAL reference's (loc, conf); then softmax + prior
    decode + per-class NMS on the reference's synthetic head tensors: boxes 1e-6 (expf), kept (prior, class) pairs and their
    order identical."""
    import builder
    from oracle import ssd_ref as SS
    g = gold("ssd_fwd_300.npz")
    m = _ssd(dev, g)
    x = (torch.from_numpy(g["x_u8"]).float() / 255.0).to(dev)
    loc, conf = m(x)
    assert tuple(loc.shape) == (2, 8732, 4) and tuple(conf.shape) == (2, 8732, 21)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    SS.FP16_STORAGE[0] = True
    try:
        with torch.no_grad():
            eloc, econf = SS.forward(sd, x.cpu(), 20)
    finally:
        SS.FP16_STORAGE[0] = False
    rloc = torch.from_numpy(g["loc"])
    e_ref = (rel(loc.cpu(), rloc), rel(conf.flatten()[::5].cpu(), torch.from_numpy(g["conf_sub"])))
    emu_ref = (rel(eloc, rloc), rel(econf.flatten()[::5], torch.from_numpy(g["conf_sub"])))
    print("ssd (loc, conf): engine vs reference", " ".join(f"{e:.3e}" for e in e_ref), "| fp16 emulation vs reference", " ".join(f"{e:.3e}" for e in emu_ref),
          "| engine vs emulation", f"{rel(loc.cpu(), eloc):.3e} {rel(conf.cpu(), econf):.3e}")
    assert all(e_ref[i] < max(1.5 * emu_ref[i], 2e-3) for i in range(2))
    cfg, algo_cls, _ = builder.export_from_registry("ssd")
    algo = algo_cls(cfg, dev)
    sloc, sconf = torch.from_numpy(g["sloc"]).float().to(dev), torch.from_numpy(g["sconf"]).float().to(dev)
    from computervision.pytorch_amd import engine as E
    boxes, prob = E.ssd_decode(sloc, sconf, torch.from_numpy(algo.anchors).to(dev))
    np.testing.assert_allclose(boxes.cpu().numpy(), torch.stack([SS.parse_loc(sloc[b].cpu(), algo.anchors) for b in range(2)]).numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(prob.cpu().numpy(), torch.softmax(sconf.cpu(), -1).numpy(), rtol=2e-6, atol=1e-9)
    # 81 score columns: rows too long for the LDS-staged form (the direct kernel runs); an anchor count that leaves the last workgroup ragged
    gq = torch.Generator().manual_seed(5)
    wide_conf, some_loc = torch.randn(1, 1000, 81, generator=gq) * 3, torch.randn(1, 1000, 4, generator=gq)
    anc = torch.from_numpy(algo.anchors)[:1000].contiguous()
    b2, p2 = E.ssd_decode(some_loc.to(dev), wide_conf.to(dev), anc.to(dev))
    np.testing.assert_allclose(p2.cpu().numpy(), torch.softmax(wide_conf, -1).numpy(), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(b2.cpu().numpy(), SS.parse_loc(some_loc[0], anc.numpy()).unsqueeze(0).numpy(), rtol=2e-6, atol=1e-7)
    b3, p3, m3 = E.ssd_decode(some_loc[:, :777].contiguous().to(dev), wide_conf[:, :777, :21].contiguous().to(dev), anc[:777].contiguous().to(dev),
                              with_class_max=True)
    np.testing.assert_allclose(p3.cpu().numpy(), torch.softmax(wide_conf[:, :777, :21], -1).numpy(), rtol=2e-6, atol=1e-9)
    assert torch.equal(m3, p3.reshape(-1, 21).amax(0))                       # per-class maxima of the batch: exact (a max of the kernel's own values)
    _, p4, m4 = E.ssd_decode(some_loc.to(dev), wide_conf.to(dev), anc.to(dev), with_class_max=True)
    assert torch.equal(m4, p4.reshape(-1, 81).amax(0))
    got = algo.decode_device((sloc, sconf))
    for b in range(2):
        rows, pairs = got[b]
        # the class loop of the oracle on the device's own boxes / scores (expf differs from torch's in the last bit): identical
        want_rows, want_pairs = SS.nms_per_class(boxes[b].cpu(), prob[b].cpu(), 20, float(g["conf_thr"]), float(g["nms_thr"]))
        assert np.array_equal(pairs.cpu().numpy(), want_pairs), b
        np.testing.assert_allclose(rows.cpu().numpy(), want_rows, rtol=1e-6, atol=1e-7)
        # ... and against what the reference produced from torch's softmax: the same detections but for score ties at 1 ulp
        ref_pairs = set(map(tuple, g[f"pairs{b}"].tolist()))
        assert len(ref_pairs & set(map(tuple, want_pairs.tolist()))) >= 0.99 * len(ref_pairs) and abs(len(want_pairs) - len(ref_pairs)) <= 4
    res = algo.decode_boxes((sloc, sconf), 240, 320)
    assert len(res) == 2 and res[0].shape[1] == 6 and np.isfinite(res[0]).all()
    assert algo.decode_boxes((loc, conf), 240, 320) == [[], []]              # the random-init network passes nothing at 0.7


# ---- DeepLabv3+ training (BASELINE configs[5]) ---------------------------------------------------------------
@pytest.mark.parametrize("B,ih,iw,H,W,nc,mode", [(2, 25, 33, 97, 129, 21, 0), (1, 9, 9, 33, 33, 21, 1), (3, 17, 12, 65, 45, 5, 0), (2, 33, 33, 33, 33, 19, 1),
                                                 (1, 5, 6, 40, 48, 21, 0)])
def test_seg_loss_kernel_against_torch(dev, B, ih, iw, H, W, nc, mode):
    """cvx_seg_loss (loss_seg.hip) against torch autograd in fp32: F.interpolate(bilinear, align_corners=False) of the logits rows
    (deeplabv3plus.py:147), then FocalLoss (focal_loss.py:14-22) or nn.CrossEntropyLoss(mean) (segmentation_2d.py:61), with ignored
    pixels; the gradient w.r.t. the low-resolution rows comes back loss-scaled in fp16.  Also the bad-label flag and the adjoint
    resize entry point on its own."""
    from computervision.pytorch_amd.deeplab import SegLoss
    lib = L.load()
    g = torch.Generator().manual_seed(B * 1000 + H + nc)
    ld = (nc + 7) & ~7
    rows = torch.zeros(B, ih * iw, ld)
    rows[..., :nc] = torch.randn(B, ih * iw, nc, generator=g) * 2
    t = torch.randint(0, nc, (B, H, W), generator=g)
    t[torch.rand(B, H, W, generator=g) < 0.15] = -100
    rr = rows[..., :nc].reshape(B, ih, iw, nc).permute(0, 3, 1, 2).clone().requires_grad_(True)
    logits = F.interpolate(rr, size=(H, W), mode="bilinear", align_corners=False)
    if mode == 0:
        ce = F.cross_entropy(logits, t, ignore_index=-100, reduction="none")
        ref = (0.25 * (1 - torch.exp(-ce)) ** 2 * ce).mean()
    else:
        ref = F.cross_entropy(logits, t, reduction="mean")
    ref.backward()
    want = rr.grad.permute(0, 2, 3, 1).reshape(B, ih * iw, nc)
    crit = SegLoss("focal" if mode == 0 else "ce")
    crit.nc = nc
    scale = 65536.0
    loss, dpred = crit.op(rows.to(dev), t.to(dev), (ih, iw), scale)
    assert abs(float(loss) - float(ref)) < 2e-5 * abs(float(ref)), (float(loss), float(ref))
    got = dpred.float().cpu() / scale
    assert rel(got[..., :nc], want) < 1e-3, rel(got[..., :nc], want)               # one fp16 rounding of the scaled gradient
    assert float(got[..., nc:].abs().max()) == 0.0 if ld > nc else True
    # rows whose stride is not a multiple of 8 floats take the scalar-load path of the pixel kernel: same numbers
    rows_odd = torch.zeros(B, ih * iw, nc + 1)
    rows_odd[..., :nc] = rows[..., :nc]
    loss_o, dpred_o = crit.op(rows_odd.to(dev), t.to(dev), (ih, iw), scale)
    assert abs(float(loss_o) - float(loss)) <= 1e-6 * abs(float(loss)) and rel(dpred_o[..., :nc].float(), dpred[..., :nc].float()) < 1e-3
    # labels outside [0, nc) that are not the ignore index: flagged (torch asserts on the device), the pixel is skipped
    tb = t.clone()
    tb[0, 0, 0] = nc + 3
    with pytest.raises(L.CvxError):
        crit.op(rows.to(dev), tb.to(dev), (ih, iw), scale)
    # the adjoint of the final resize alone (callers with their own loss on the NCHW logits)
    gl = torch.randn(B, nc, H, W, generator=g)
    rin = rr.detach().clone().requires_grad_(True)
    F.interpolate(rin, size=(H, W), mode="bilinear", align_corners=False).backward(gl)
    rr2 = rin.grad
    d2 = torch.empty(B, ih * iw, ld, dtype=torch.float16, device=dev)
    L.check(lib.cvx_resize_bilinear_nchw_grad_to_rows(L.ptr(gl.to(dev)), B, nc, ih, iw, H, W, 0.125, L.ptr(d2), ld, L.stream_ptr(dev)), "adjoint")
    assert rel(d2.float().cpu()[..., :nc] * 8, rr2.permute(0, 2, 3, 1).reshape(B, ih * iw, nc)) < 1e-3


@pytest.mark.parametrize("B,ih,iw,H,W,nc,ld", [(2, 9, 12, 33, 45, 21, 24), (1, 5, 5, 17, 17, 5, 5), (2, 7, 6, 7, 6, 8, 8), (1, 33, 33, 129, 129, 3, 8)])
def test_logit_rows_to_nchw_resize(dev, B, ih, iw, H, W, nc, ld):
    """cvx_resize_bilinear_rows_to_nchw (the final F.interpolate(bilinear, align_corners=False) of the segmentation logits,
    deeplabv3plus.py:147) against torch in fp32: 16-byte row loads with a scalar tail (21 of 24 columns), rows whose stride rules the
    vector loads out (ld = 5), identity size."""
    lib = L.load()
    g = torch.Generator().manual_seed(ih * 7 + nc)
    rows = torch.randn(B, ih * iw, ld, generator=g)
    want = F.interpolate(rows[..., :nc].reshape(B, ih, iw, nc).permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False)
    out = torch.empty(B, nc, H, W, device=dev)
    rd = rows.to(dev)
    L.check(lib.cvx_resize_bilinear_rows_to_nchw(L.ptr(rd), ld, B, nc, ih, iw, H, W, L.ptr(out), L.stream_ptr(dev)), "rows_to_nchw")
    assert float((out.cpu() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


def _deeplab_train_model(dev, g, dropout_p=0.0):
    from computervision.pytorch_amd.deeplab import DeepLabV3PlusR101
    torch.manual_seed(0)
    m = DeepLabV3PlusR101(21, dropout_p=dropout_p)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if k.endswith(".bn3.weight"):
                v.fill_(float(g["bn3_gamma"]))              # the fixture's conditioning (make_golden.py, section 10b)
    return m.to(dev).train()


def test_deeplab_training_step_matches_the_reference_fixture(dev, gold):
    """model.train(); loss = FocalLoss()(model(x), t); loss.backward() on the engine against the REAL reference's step on the
    fixture batch (make_golden.py section 10b): loss value, logits rows, updated running statistics, and all 338 parameter gradients.
    The gradients of a 100-layer ReLU network cannot agree to 1e-3 between an fp16-operand forward and an fp32 one: a forward
    perturbation of 5e-3 flips the ReLU masks of ~0.4 % of the elements in every layer, and each flip changes that element's
    gradient by 100 % -- the reference's own arithmetic with fp16-rounded operands (the oracle's emulation, run here) differs from
    its fp32 gradients by ~14 %.  That difference is the yardstick: the engine must stay within 1.25x of it overall and per
    tensor, and to 1e-2 on the classifier's last layers where no mask has intervened yet.  The exactness of the backward pass
    itself is the business of test_deeplab_per_layer_backward_on_the_engines_own_operands."""
    from computervision.pytorch_amd.deeplab import SegLoss
    from oracle import deeplab_ref as D
    g = gold("deeplab_train_97x129.npz")
    m = _deeplab_train_model(dev, g)
    sd0 = {k: v.cpu().clone() for k, v in m.state_dict().items()}
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["target"].astype(np.int64))
    crit = SegLoss("focal")
    out = m(x.to(dev))
    assert tuple(out.shape) == (2, 21, 97, 129)
    loss = crit(out, t.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4 * float(g["loss"]), (float(loss.detach()), float(g["loss"]))
    # the yardstick, and the full fp32 gradients (the oracle is pinned to the fixture in tests/test_oracle_golden.py)
    _, ref_grads, ref_rows = D.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, t)
    D.FP16_STORAGE[0] = True
    try:
        _, emu_grads, emu_rows = D.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, t)
    finally:
        D.FP16_STORAGE[0] = False
    lh, lw = m._last_engine.graph.level_hw[0]
    rows = m.last_rows[..., :21].reshape(2, lh, lw, 21).cpu()
    assert rel(ref_rows, torch.from_numpy(g["rows"])) < 1e-4          # (another CPU, another summation order)
    e_rows, y_rows = rel(rows, ref_rows), rel(emu_rows, ref_rows)
    assert e_rows < max(1.25 * y_rows, 1e-2), (e_rows, y_rows)
    eg = {k: p.grad.cpu() for k, p in m.named_parameters()}
    assert list(eg.keys()) == [str(k) for k in g["grad_keys"]]

    def total(a, b):
        num = sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in b)
        return (num / sum(float(b[k].double().pow(2).sum()) for k in b)) ** 0.5

    e_tot, y_tot = total(eg, ref_grads), total(emu_grads, ref_grads)
    dot = sum(float((eg[k].double() * ref_grads[k].double()).sum()) for k in eg)
    cos = dot / (sum(float(eg[k].double().pow(2).sum()) for k in eg) * sum(float(ref_grads[k].double().pow(2).sum()) for k in eg)) ** 0.5
    print(f"deeplab train: loss {float(loss.detach()):.6f} (ref {float(g['loss']):.6f}); rows {e_rows:.2e} (yardstick {y_rows:.2e}); "
          f"grads {e_tot:.3e} (yardstick {y_tot:.3e}), cosine {cos:.4f}")
    assert e_tot < 1.25 * y_tot and cos > 0.985, (e_tot, y_tot, cos)
    for k in eg:
        e, y = rel(eg[k], ref_grads[k]), rel(emu_grads[k], ref_grads[k])
        assert e < max(1.6 * y, 2e-2), (k, e, y)
    for k in ("classifier.classifier.3.weight", "classifier.classifier.3.bias", "classifier.classifier.1.weight"):
        assert rel(eg[k], torch.from_numpy(g["g:" + k])) < 1e-2, (k, rel(eg[k], torch.from_numpy(g["g:" + k])))
    sdm = m.state_dict()
    for k in [str(k) for k in g["stat_keys"]]:
        assert rel(sdm[k].cpu(), torch.from_numpy(g["s:" + k])) < 2e-2, k
    assert int(sdm["backbone.bn1.num_batches_tracked"]) == 1


def _check_backward_per_layer(m, B, dpred, min_layers, min_buffers):
    """Every op of the last backward pass of model `m` against fp64 on the ENGINE'S OWN operands (cvx_engine_debug_copy): for each
    Conv + BatchNorm block its xhat, its forward output / residual, the gradient g arriving at its output and the gradient dy it
    hands on -> dgamma / dbeta (2e-6), dy (one fp16 rounding), the weight gradient conv_wgrad(x, dy) (5e-4: fp32 MFMA accumulation of
    fp16 products); for each activation buffer the gradient it ends up with = the sum over ALL its consumers (data gradients,
    residual branches taking dz, pool / resize / upsample / dropout backward), each recomputed in fp64 from that consumer's own
    operands (2e-3: one fp16 rounding per accumulation)."""
    eng, lay, scale = m._last_engine, m.layout, m.loss_scale
    gr = eng.graph
    P, G = m.flat_params.double().cpu(), m.flat_grads.double().cpu()
    no_pad = gr.bufs[gr.pred_buf][2]
    dpred = dpred.double().cpu().reshape(B, gr.anchors, no_pad)

    def act(view, grad=False):                              # (B, h, w, c) fp64 slice of an engine buffer
        b, off, c = view[0], view[1], view[2]
        h, w, cc, _ = gr.bufs[b]
        return eng.read_buffer(b, B, grad=grad).double().cpu().reshape(B, h, w, cc)[..., off:off + c]

    nchw = lambda a: a.permute(0, 3, 1, 2).contiguous()     # noqa: E731
    nhwc = lambda a: a.permute(0, 2, 3, 1).contiguous()     # noqa: E731
    expect, covered, unknown = {}, {}, {}                   # buffer -> summed fp64 input-gradient contributions, per channel

    def add(view, contrib):
        b, off, c = view[0], view[1], view[2]
        if b not in expect:
            h, w, cc, _ = gr.bufs[b]
            expect[b] = torch.zeros(B, h, w, cc, dtype=torch.float64)
            covered[b] = torch.zeros(cc, dtype=torch.bool)
        expect[b][..., off:off + c] += contrib               # consumers may read overlapping slices of a concat buffer (ELAN)
        covered[b][off:off + c] = True

    def spec(name):
        cs = lay.head_first if name == "backbone.heads.0" else lay.convs[name]
        return cs if isinstance(cs, dict) else dict(cout_eng=cs.cout_eng, cin=cs.cin, k=cs.k, w_off=cs.w_off, gamma_off=cs.gamma_off,
                                                    beta_off=cs.beta_off, bias_off=cs.bias_off)

    n_w = n_bn = 0
    for i, o in enumerate(gr.ops):
        typ = o["type"]
        if typ == L.OP_CONV:
            cs = spec(o["name"])
            C, cin, k = cs.get("cout_eng", o["out"][2]), cs["cin"], cs["k"]
            C = o["out"][2]
            xin = act(o["in"])[..., :cin]
            wq = P[cs["w_off"]:cs["w_off"] + C * k * k * cin].reshape(C, k, k, cin).permute(0, 3, 1, 2).float().half().double()
            if o["act"] == L.ACT_BIAS:
                a0 = o["out"][3]
                dy = dpred[:, a0:a0 + o["oh"] * o["ow"], o["out"][1]:o["out"][1] + C].reshape(B, o["oh"], o["ow"], C)
                want_b = dy.reshape(-1, C).sum(0) / scale
                got_b = G[cs["bias_off"]:cs["bias_off"] + C]
                assert rel(got_b, want_b) < 1e-5 or float(want_b.norm()) == 0.0, o["name"]
            elif o["act"] in (L.ACT_BIAS_RELU, L.ACT_BIAS_LINEAR):      # conv + bias (+ ReLU), no BatchNorm: dy = g * [out > 0]
                gout = act(o["out"], grad=True)
                want_dy = gout * (act(o["out"]) > 0) if o["act"] == L.ACT_BIAS_RELU else gout
                dy = eng.read_layer(i, B, "dy").double().cpu().reshape(B, o["oh"], o["ow"], C)
                assert torch.equal(dy, want_dy), o["name"]
                got_b, want_b = G[cs["bias_off"]:cs["bias_off"] + C], dy.reshape(-1, C).sum(0) / scale
                assert rel(got_b, want_b) < 1e-5, (o["name"], rel(got_b, want_b))
            else:
                xh = eng.read_layer(i, B, "xhat").double().cpu().reshape(-1, C)
                dy = eng.read_layer(i, B, "dy").double().cpu().reshape(B, o["oh"], o["ow"], C)
                gout = act(o["out"], grad=True).reshape(-1, C)
                ga, be = P[cs["gamma_off"]:cs["gamma_off"] + C], P[cs["beta_off"]:cs["beta_off"] + C]
                pre_res = "res" in o and (o.get("flags", 0) & L.OPF_RES_PRE_ACT)
                if o["act"] == L.ACT_BN_RELU:
                    dz = gout * (act(o["out"]).reshape(-1, C) > 0)
                elif o["act"] == L.ACT_BN_SILU:
                    z = xh * ga + be + (act(o["res"]).reshape(-1, C) if pre_res else 0)
                    sg = torch.sigmoid(z)
                    dz = gout * (sg * (1 + z * (1 - sg)))
                else:
                    dz = gout
                want_g, want_b = (dz * xh).sum(0) / scale, dz.sum(0) / scale
                got_g, got_b = G[cs["gamma_off"]:cs["gamma_off"] + C], G[cs["beta_off"]:cs["beta_off"] + C]
                if float(want_g.norm()) > 0:
                    assert rel(got_g, want_g) < 2e-6 and rel(got_b, want_b) < 2e-6, (o["name"], rel(got_g, want_g), rel(got_b, want_b))
                    core = dz - dz.mean(0) - xh * (dz * xh).mean(0)
                    gi = (dy.reshape(-1, C) * core).sum(0) / (core * core).sum(0).clamp_min(1e-300)
                    rms = float(dy.pow(2).mean().sqrt())
                    e_dy = rel(dy.reshape(-1, C), core * gi)
                    assert e_dy < 1e-3 + 6e-8 / max(rms, 1e-30), (o["name"], e_dy, rms)
                    n_bn += 1
                if "res" in o:                 # the residual branch receives dz (residual inside the activation) or the incoming gradient
                    add(o["res"], (dz if pre_res else gout).reshape(B, o["oh"], o["ow"], C))
            xr = nchw(xin).requires_grad_(o.get("needs_dgrad", 1) == 1)
            wr = wq.clone().requires_grad_(True)
            y = F.conv2d(xr, wr, None, o["stride"], o["pad"], o["dil"])
            y.backward(nchw(dy))
            got_w = G[cs["w_off"]:cs["w_off"] + C * k * k * cin].reshape(C, k, k, cin).permute(0, 3, 1, 2)
            want_w = wr.grad / scale
            if float(want_w.norm()) > 0:
                assert rel(got_w, want_w) < 5e-4, (o["name"], rel(got_w, want_w))
                n_w += 1
            if xr.requires_grad:
                g_in = nhwc(xr.grad)
                if g_in.shape[-1] < o["in"][2]:                  # (views padded beyond the stored input channels)
                    g_in = F.pad(g_in, (0, o["in"][2] - g_in.shape[-1]))
                add(o["in"], g_in)
            continue
        xin = nchw(act(o["in"])).requires_grad_(True)
        gout = nchw(act(o["out"], grad=True))
        if typ == L.OP_MAXPOOL3S2:
            y = F.max_pool2d(xin, 3, 2, 1)
        elif typ == L.OP_MAXPOOL2:
            y = F.max_pool2d(xin, 2, 2, ceil_mode=(o["oh"] * 2 != o["ih"]))
        elif typ == L.OP_MAXPOOL3S1:
            y = F.max_pool2d(xin, 3, 1, 1)
        elif typ == L.OP_L2NORM:                             # x / (||x|| + 1e-10) * weight (ssd_model.py:113-128)
            Cn = o["in"][2]
            wn = P[o["gamma_off"]:o["gamma_off"] + Cn].clone().requires_grad_(True)
            y = wn.view(1, -1, 1, 1) * (xin / (xin.pow(2).sum(1, keepdim=True).sqrt() + 1e-10))
            y.backward(gout)
            got_wn = G[o["gamma_off"]:o["gamma_off"] + Cn]
            assert rel(got_wn, wn.grad / scale) < 1e-5, (o["name"], rel(got_wn, wn.grad / scale))
            add(o["in"], nhwc(xin.grad))
            continue
        elif typ == L.OP_MAXPOOL5:
            y = F.max_pool2d(xin, 5, 1, 2)
        elif typ == L.OP_UPSAMPLE2:
            y = F.interpolate(xin, scale_factor=2, mode="nearest")
        elif typ == L.OP_AVGPOOL:
            y = F.adaptive_avg_pool2d(xin, 1)
        elif typ == L.OP_COPY:
            y = xin * 1.0
        elif typ == L.OP_DWCONVT:                            # depthwise ConvTranspose2d(k = 2f, s = f, p = f/2), fp32 master weights [C][2f][2f]
            f, Cd = o["stride"], o["in"][2]
            wd = P[o["w_off"]:o["w_off"] + Cd * 4 * f * f].reshape(Cd, 1, 2 * f, 2 * f).clone().requires_grad_(True)
            y = F.conv_transpose2d(xin, wd, None, stride=f, padding=f // 2, groups=Cd)
            y.backward(gout)
            got_w = G[o["w_off"]:o["w_off"] + Cd * 4 * f * f].reshape(Cd, 1, 2 * f, 2 * f)
            assert rel(got_w, wd.grad / scale) < 1e-5, (o["name"], rel(got_w, wd.grad / scale))
            add(o["in"], nhwc(xin.grad))
            continue
        elif typ == L.OP_RESIZE:
            y = F.interpolate(xin, size=(o["oh"], o["ow"]), mode="bilinear", align_corners=False)
        elif typ == L.OP_DROPOUT:
            out = nchw(act(o["out"]))
            keep = (out != 0) | (xin.detach() == 0)
            nz = xin.detach() != 0
            frac = float((out[nz] == 0).double().mean())
            assert 0.08 < frac < 0.12, frac                                                  # p = 0.1 of the non-zero elements dropped
            assert rel(out, xin.detach() * keep / 0.9) < 1e-3                                # inverted scaling, one fp16 rounding
            y = xin * keep / 0.9
            unknown[o["in"][0]] = (o["in"][1], o["in"][2], nhwc(~nz))                         # the mask is not observable where the input is 0
        else:
            raise AssertionError(f"op type {typ} without a backward check")
        y.backward(gout)
        add(o["in"], nhwc(xin.grad))
    assert n_w >= min_layers and n_bn >= min_layers, (n_w, n_bn)
    checked = 0
    for b, want in expect.items():
        if b == gr.image_buf:
            continue
        got = act((b, 0, gr.bufs[b][2]), grad=True)
        if b in unknown:
            off, c, mask = unknown[b]
            want[..., off:off + c] = torch.where(mask, got[..., off:off + c], want[..., off:off + c])
        got, want = got[..., covered[b]], want[..., covered[b]]
        rms = float(want.pow(2).mean().sqrt())
        e = rel(got, want)
        assert e < 2e-3 + 6e-8 / max(rms, 1e-30), (b, e, rms)
        checked += 1
    assert checked >= min_buffers, checked
    return n_w, n_bn, checked


def test_deeplab_per_layer_backward_on_the_engines_own_operands(dev, gold):
    """_check_backward_per_layer on one DeepLabv3+ training step with dropout active: all 113 convolutions incl. the 7x7 stem on the
    NHWC image copy, the dilated 3x3 and the 1x1 stride-2 downsample; max pool / average pool / resize / dropout backward; the
    Bottleneck identity branches taking dz.  Together with the unit tests this pins the backward pass op by op, which the
    end-to-end comparison (ReLU mask flips) cannot."""
    from computervision.pytorch_amd.deeplab import SegLoss
    g = gold("deeplab_train_97x129.npz")
    m = _deeplab_train_model(dev, g, dropout_p=0.1)
    m.seed = 5
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["target"].astype(np.int64))
    loss = SegLoss("focal")(m(x.to(dev)), t.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    _check_backward_per_layer(m, x.shape[0], m.last_dpred, 100, 100)


def test_deeplab_step_with_gradient_exchange_and_loss_scaling(dev, gold):
    """SegTrainStep: (a) with an RCCL group of size 1 the gradient exchange after the backward pass (4 slices of the flat arena on a
    side stream, 1/world folded into Adam) gives exactly the plain step's losses and parameters; (b) GradScaler semantics: an
    absurd initial scale overflows the fp16 gradients, the step is skipped on the device and the scale backs off until steps
    go through; (c) the fused step repeats bit for bit from the same seed (dropout masks included)."""
    import torch.distributed as dist
    from computervision.pytorch_amd.deeplab import SegLoss, SegTrainStep
    from computervision.pytorch_amd.train import DynamicLossScale, FlatAdam
    g = gold("deeplab_train_97x129.npz")
    x, t = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["target"].astype(np.int64)).to(dev)
    created = False
    if not dist.is_initialized():
        try:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29613", rank=0, world_size=1, device_id=dev)
            created = True
        except Exception as exc:
            pytest.skip(f"RCCL process group unavailable: {exc}")
    try:
        runs = []
        for distributed in (False, True, True):
            m = _deeplab_train_model(dev, g, dropout_p=0.1)
            m.seed = 3
            step = SegTrainStep(m, SegLoss("focal"), FlatAdam(m, lr=1e-3), scaler=DynamicLossScale(dev, init_scale=4096.0))
            step.distributed = distributed
            losses = [step(x, t).clone() for _ in range(3)]
            torch.cuda.synchronize()
            runs.append((torch.cat(losses).cpu(), m.flat_params.clone().cpu()))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
        assert torch.equal(runs[1][0], runs[2][0]) and torch.equal(runs[1][1], runs[2][1])
        assert float(runs[0][0][-1]) < float(runs[0][0][0])
    finally:
        if created:
            dist.destroy_process_group()
    m = _deeplab_train_model(dev, g)
    sc = DynamicLossScale(dev, init_scale=2.0 ** 40, growth_interval=1000)
    opt = FlatAdam(m, lr=1e-3)
    step = SegTrainStep(m, SegLoss("focal"), opt, scaler=sc)
    p0 = m.flat_params.clone()
    step(x, t)
    torch.cuda.synchronize()
    assert torch.equal(m.flat_params, p0)
    sc.poll()
    assert sc.scale == 2.0 ** 39 and sc.skipped == 1
    for _ in range(60):
        step(x, t)
        torch.cuda.synchronize()
    sc.poll()
    assert sc.scale < 2.0 ** 34 and not torch.equal(m.flat_params, p0) and bool(torch.isfinite(m.flat_params).all())


# ---- input side: letterbox pre-processing ------------------------------------------------------------------
@pytest.mark.parametrize("h,w,H,W", [(480, 640, 640, 640), (375, 500, 640, 640), (1080, 1920, 640, 640), (333, 77, 300, 300), (97, 129, 513, 513),
                                     (640, 640, 640, 640), (3, 5, 64, 96), (2000, 31, 512, 384)])
def test_letterbox_kernel_is_bit_exact(dev, h, w, H, W):
    """cvx_letterbox_u8_to_nchw against oracle/letterbox_ref.py (letter_box + TF.to_tensor, image_process.py:41-66): byte and index
    work -> bit-exact, up- and down-scaling, odd sizes, BGR -> RGB swap, the batch-slot form (images_to_batch) and the plain nearest
    resize (letterbox = 0)."""
    from computervision.pytorch_amd.engine import letterbox_u8
    from core.utils.image_process import images_to_batch, letter_box
    from oracle import letterbox_ref as LB
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref_img, ref_scale, ref_pads = LB.letter_box(img, (H, W))
    want = torch.from_numpy(LB.to_tensor(ref_img))
    x, scale, pads = letter_box(img, (H, W), device=dev)
    assert scale == ref_scale and pads == ref_pads and tuple(x.shape) == (1, 3, H, W)
    assert torch.equal(x[0].cpu(), want)
    xs, _, _ = letter_box(img[:, :, ::-1], (H, W), device=dev, swap_rb=True)
    assert torch.equal(xs[0].cpu(), want)
    from core.utils.image_process import read_image_and_convert_to_tensor
    xr, rh, rw = read_image_and_convert_to_tensor(img, (H, W), letterbox=True, device=dev)     # what every predict() feeds its model
    assert (rh, rw) == (h, w) and torch.equal(xr[0].cpu(), want)
    other = rng.integers(0, 256, (w + 1, h + 2, 3), dtype=np.uint8)
    batch = images_to_batch([img, other, img], (H, W), dev)
    assert torch.equal(batch[0].cpu(), want) and torch.equal(batch[2].cpu(), want)
    assert torch.equal(batch[1].cpu(), torch.from_numpy(LB.to_tensor(LB.letter_box(other, (H, W))[0])))
    plain = torch.empty(3, H, W, device=dev)
    letterbox_u8(torch.from_numpy(img).to(dev), plain, letterbox=False)
    assert torch.equal(plain.cpu(), torch.from_numpy(LB.to_tensor(LB.resize_nearest(img, H, W))))



# ---- YOLOv7-l network forward + backward in training mode --------------------------------------------------
def _yolov7_train_step(dev, g):
    from computervision.pytorch_amd.yolov7 import Yolo7L
    from oracle import yolov7_ref as Y7
    torch.manual_seed(0)
    m = Yolo7L(20)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    x = torch.from_numpy(g["x"])
    outs = m(x.to(dev))
    weights = Y7.projection_weights([o.shape for o in outs], int(g["proj_seed"]))
    loss = Y7.projection_loss(outs, [w.to(dev) for w in weights])
    loss.backward()
    torch.cuda.synchronize()
    return m, sd0, x, outs, weights


def test_yolov7_training_forward_backward_matches_the_reference_fixture(dev, gold):
    """model.train(); outs = model(x); loss(outs).backward() on the engine against the REAL reference's autograd (make_golden.py section
    11b: a fixed linear functional of the three outputs): train-mode outputs, running statistics, all 282 parameter gradients.  A
    random-init YOLOv7 (N(0, 0.02) weights, 105 layers) amplifies fp16 rounding: the reference's own arithmetic with fp16-rounded
    operands (the oracle's emulation, run here) is off by 3-8 % on the outputs and ~26 % on the deepest gradients at this size.  That
    is the yardstick; the engine must stay within 1.25x of it overall and 1.6x per tensor.  The exactness of the backward pass
    itself: test_yolov7_per_layer_backward_on_the_engines_own_operands."""
    from oracle import yolov7_ref as Y7
    g = gold("yolov7_train_160x224.npz")
    m, sd0, x, outs, weights = _yolov7_train_step(dev, g)
    _, ref_grads, ref_outs = Y7.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, weights)
    Y7.FP16_STORAGE[0] = True
    try:
        _, emu_grads, emu_outs = Y7.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, weights)
    finally:
        Y7.FP16_STORAGE[0] = False
    assert rel(torch.cat([o.flatten()[::7] for o in ref_outs]), torch.from_numpy(g["out_sub"])) < 1e-4
    for o, r, e in zip(outs, ref_outs, emu_outs):
        assert tuple(o.shape) == tuple(r.shape)
        assert rel(o.detach().cpu(), r) < max(1.25 * rel(e, r), 1e-2), (rel(o.detach().cpu(), r), rel(e, r))
    eg = {k: p.grad.cpu() for k, p in m.named_parameters()}
    assert list(eg.keys()) == [str(k) for k in g["grad_keys"]]

    def total(a, b):
        return (sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in b) / sum(float(b[k].double().pow(2).sum()) for k in b)) ** 0.5

    e_tot, y_tot = total(eg, ref_grads), total(emu_grads, ref_grads)
    print(f"yolov7 train: gradients engine vs reference {e_tot:.3e}, yardstick {y_tot:.3e}")
    assert e_tot < 1.25 * y_tot, (e_tot, y_tot)
    for k in eg:
        e, y = rel(eg[k], ref_grads[k]), rel(emu_grads[k], ref_grads[k])
        assert e < max(1.6 * y, 2e-2), (k, e, y)
    sdm = m.state_dict()
    for k in [str(k) for k in g["stat_keys"]]:
        assert rel(sdm[k].cpu(), torch.from_numpy(g["s:" + k])) < 5e-2, k
    assert int(sdm["backbone.stem.0.bn.num_batches_tracked"]) == 1


def test_yolov7_per_layer_backward_on_the_engines_own_operands(dev, gold):
    """_check_backward_per_layer on the YOLOv7 step: all 95 convolutions (92 Conv + BN + SiLU blocks incl. the RepConv pair (3x3 + BN into a buffer, 1x1 + BN
    with that buffer inside the SiLU: the residual branch takes dz), the three biased heads on their slices of the prediction rows,
    2x2 and 5x5 max pools, nearest upsampling, the concat buffers' accumulated gradients."""
    g = gold("yolov7_train_160x224.npz")
    m, _, x, _, _ = _yolov7_train_step(dev, g)
    n_w, n_bn, checked = _check_backward_per_layer(m, x.shape[0], m.last_dpred, 90, 50)
    assert n_w >= 95 and n_bn >= 92


def test_yolov7_training_through_the_plugin_api(dev):
    """export_from_registry("yolo7") at 640 x 640, batch 4: three optimisation steps with a torch-side loss on the model's outputs and
    torch.optim.Adam over the parameters (their .grad are views of the engine's gradient arena): finite, the loss falls."""
    import builder
    cfg, algo_cls, _ = builder.export_from_registry("yolo7")
    cfg.train.pretrained = False
    torch.manual_seed(0)
    model, _ = algo_cls(cfg, dev).build_model()
    model = model.to(dev).train()
    x = synth.images(4, 640, 640, seed=2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=False)
        outs = model(x)
        assert [tuple(o.shape) for o in outs] == [(4, 75, 20, 20), (4, 75, 40, 40), (4, 75, 80, 80)]
        loss = sum(o.pow(2).mean() for o in outs)
        loss.backward()
        for p in model.parameters():
            p.grad.div_(model.loss_scale)                    # the arena holds loss_scale x gradient (GradScaler.unscale_)
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0], losses



# ---- CenterNet DLA-34 network forward + backward in training mode -------------------------------------------
def _centernet_train_step(dev, g):
    from computervision.pytorch_amd.dla import CenterNetDLA34
    from oracle import centernet_ref as C
    nc = int(g["nc"])
    torch.manual_seed(0)
    m = CenterNetDLA34(nc)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    x = torch.from_numpy(g["x"])
    out = m(x.to(dev))
    weights = C.projection_weights(out.shape, int(g["proj_seed"]))
    loss = C.projection_loss(out, weights.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    return m, sd0, x, out, weights, nc


def test_centernet_training_forward_backward_matches_the_reference_fixture(dev, gold):
    """model.train(); out = model(x); loss(out).backward() on the engine against the REAL reference's autograd (make_golden.py section
    9b): the train-mode output tensor, running statistics, all 165 parameter gradients incl. the depthwise transposed convolutions'
    weights.  Yardstick as for DeepLab (ReLU network: mask flips under fp16 rounding): the oracle's fp16-operand emulation, run here."""
    from oracle import centernet_ref as C
    g = gold("centernet_train_128x160.npz")
    m, sd0, x, out, weights, nc = _centernet_train_step(dev, g)
    _, ref_grads, ref_out = C.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, nc, weights)
    C.FP16_STORAGE[0] = True
    try:
        _, emu_grads, emu_out = C.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, nc, weights)
    finally:
        C.FP16_STORAGE[0] = False
    assert rel(ref_out.flatten()[::7], torch.from_numpy(g["out_sub"])) < 1e-4
    e_out, y_out = rel(out.detach().cpu(), ref_out), rel(emu_out, ref_out)
    assert tuple(out.shape) == tuple(ref_out.shape) and e_out < max(1.25 * y_out, 1e-2), (e_out, y_out)
    eg = {k: p.grad.cpu() for k, p in m.named_parameters() if k in ref_grads}
    assert set(eg.keys()) == set(str(k) for k in g["grad_keys"])

    def total(a, b):
        return (sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in b) / sum(float(b[k].double().pow(2).sum()) for k in b)) ** 0.5

    e_tot, y_tot = total(eg, ref_grads), total(emu_grads, ref_grads)
    print(f"centernet train: output {e_out:.3e} (yardstick {y_out:.3e}); gradients {e_tot:.3e} (yardstick {y_tot:.3e})")
    assert e_tot < max(1.25 * y_tot, 1e-2), (e_tot, y_tot)
    for k in eg:
        e, y = rel(eg[k], ref_grads[k]), rel(emu_grads[k], ref_grads[k])
        assert e < max(1.6 * y, 2e-2), (k, e, y)
    sdm = m.state_dict()
    for k in [str(k) for k in g["stat_keys"]]:
        assert rel(sdm[k].cpu(), torch.from_numpy(g["s:" + k])) < 2e-2, k


def test_centernet_per_layer_backward_on_the_engines_own_operands(dev, gold):
    """_check_backward_per_layer on the CenterNet step: every Conv + BN + ReLU block (BasicBlock residual inside the ReLU), the 7x7 stem,
    the Tree projections (BN without activation), 2x2 max pools, the concat copies, the depthwise transposed convolutions (data
    gradient, and their fp32 weight gradient 1e-5), the fused biased 3x3 head (dy = g * [out > 0] exactly) and the three 1x1 heads."""
    g = gold("centernet_train_128x160.npz")
    m, _, x, _, _, _ = _centernet_train_step(dev, g)
    n_w, n_bn, checked = _check_backward_per_layer(m, x.shape[0], m.last_dpred, 49, 40)
    assert n_w >= 53


def test_centernet_training_through_the_plugin_api(dev):
    """export_from_registry("centernet") at 512 x 512, batch 4: three optimisation steps with a torch-side loss on the model's output
    and torch.optim.Adam over the parameters (their .grad are views of the engine's gradient arena): finite, the loss falls."""
    import builder
    cfg, algo_cls, _ = builder.export_from_registry("centernet")
    torch.manual_seed(0)
    model, _ = algo_cls(cfg, dev).build_model()
    model = model.to(dev).train()
    x = synth.images(4, 512, 512, seed=2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=False)
        out = model(x)
        assert tuple(out.shape) == (4, 128, 128, cfg.dataset.num_classes + 4)
        loss = out.pow(2).mean()
        loss.backward()
        for p in model.parameters():
            if p.grad is not None:
                p.grad.div_(model.loss_scale)                # the arena holds loss_scale x gradient (GradScaler.unscale_)
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0], losses


# ---- SSD300 VGG16-BN network forward + backward in training mode -------------------------------------------
def _ssd_train_step(dev, g):
    from computervision.pytorch_amd.ssd import SSD300VGG
    from oracle import ssd_ref as S
    nc = int(g["nc"])
    torch.manual_seed(0)
    m = SSD300VGG(nc)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev).train()
    x = torch.from_numpy(g["x"].astype(np.float32) / 255.0)
    outs = m(x.to(dev))
    weights = S.projection_weights([o.shape for o in outs], int(g["proj_seed"]))
    loss = S.projection_loss(outs, [w.to(dev) for w in weights])
    loss.backward()
    torch.cuda.synchronize()
    return m, sd0, x, outs, weights, nc


def test_ssd_training_forward_backward_matches_the_reference_fixture(dev, gold):
    """model.train(); loc, conf = model(x); loss(loc, conf).backward() on the engine against the REAL reference's autograd
    (make_golden.py section 12b): train-mode outputs (NCHW-order flattening included), running statistics (the convolution's bias enters
    the running mean), all 97 parameter gradients incl. L2Normalize's weight; conv biases in front of a BatchNorm have zero gradient
    (exactly here, round-off in torch).  Yardstick: the oracle's fp16-operand emulation, run here."""
    from oracle import ssd_ref as S
    g = gold("ssd_train_300.npz")
    m, sd0, x, outs, weights, nc = _ssd_train_step(dev, g)
    _, ref_grads, ref_outs = S.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, nc, weights)
    S.FP16_STORAGE[0] = True
    try:
        _, emu_grads, emu_outs = S.loss_and_grads({k: v.clone() for k, v in sd0.items()}, x, nc, weights)
    finally:
        S.FP16_STORAGE[0] = False
    assert rel(torch.cat([o.flatten()[::7] for o in ref_outs]), torch.from_numpy(g["out_sub"])) < 1e-4
    for o, r, e in zip(outs, ref_outs, emu_outs):
        assert tuple(o.shape) == tuple(r.shape)
        assert rel(o.detach().cpu(), r) < max(1.25 * rel(e, r), 1e-2), (rel(o.detach().cpu(), r), rel(e, r))
    eg = {k: p.grad.cpu() for k, p in m.named_parameters() if k in ref_grads}
    gmax = max(float(v.norm()) for v in ref_grads.values())
    live = [k for k in ref_grads if float(ref_grads[k].norm()) >= 1e-6 * gmax]
    dead = [k for k in ref_grads if k not in live]
    assert len(dead) == 13 and all(float(eg[k].abs().max()) == 0.0 for k in dead)       # the 13 VGG convolution biases behind a BatchNorm

    def total(a, b):
        return (sum(float((a[k].double() - b[k].double()).pow(2).sum()) for k in live) / sum(float(b[k].double().pow(2).sum()) for k in live)) ** 0.5

    e_tot, y_tot = total(eg, ref_grads), total(emu_grads, ref_grads)
    print(f"ssd train: gradients engine vs reference {e_tot:.3e}, yardstick {y_tot:.3e}")
    assert e_tot < max(1.25 * y_tot, 1e-2), (e_tot, y_tot)
    for k in live:
        e, y = rel(eg[k], ref_grads[k]), rel(emu_grads[k], ref_grads[k])
        assert e < max(1.6 * y, 2e-2), (k, e, y)
    sdm = m.state_dict()
    for k in [str(k) for k in g["stat_keys"]]:
        assert rel(sdm[k].cpu(), torch.from_numpy(g["s:" + k])) < 2e-2, k


def test_ssd_per_layer_backward_on_the_engines_own_operands(dev, gold):
    """_check_backward_per_layer on the SSD step: the 13 Conv(bias) + BN + ReLU blocks of VGG16, conv6 / conv7 (bias + ReLU), the eight
    activation-free extra layers, the twelve 3x3 heads on their column ranges of the prediction rows, 2x2 (one ceil-mode) and 3x3
    stride-1 max pools, L2Normalize (data gradient and its weight gradient)."""
    g = gold("ssd_train_300.npz")
    m, _, x, _, _, _ = _ssd_train_step(dev, g)
    n_w, n_bn, checked = _check_backward_per_layer(m, x.shape[0], m.last_dpred, 13, 25)
    assert n_w >= 35


@pytest.mark.parametrize("tag", ["a", "b"])
def test_centernet_loss_kernel_matches_the_reference_fixture(dev, gold, tag):
    """cvx_centernet_loss against the REAL reference's CombinedLoss + torch autograd (make_golden.py section 9c): random head outputs,
    (a) objects present, two of them on one centre; (b) no object at all (num_pos == 0 branch, empty masks).  Loss value 1e-5, the
    gradient on the head rows to one fp16 rounding, the row padding zero."""
    from computervision.pytorch_amd.dla import CenterNetLoss
    g = gold("centernet_loss.npz")
    nc = int(g["nc"])
    hm_w, wh_w, off_w = (float(v) for v in g["weights"])
    pred = torch.from_numpy(g[tag + "_pred"])
    B, h, w, _ = pred.shape
    nc_pad = (nc + 7) & ~7
    ld = nc_pad + 16
    rows = torch.zeros(B, h * w, ld)
    flat = pred.reshape(B, h * w, nc + 4)
    rows[..., :nc], rows[..., nc_pad:nc_pad + 2], rows[..., nc_pad + 8:nc_pad + 10] = flat[..., :nc], flat[..., nc:nc + 2], flat[..., nc + 2:]
    targets = [torch.from_numpy(g[tag + "_" + k]) for k in ("heat", "reg", "wh", "mask", "idx")]
    crit = CenterNetLoss(nc, hm_w, wh_w, off_w)
    scale = 64.0
    items, dpred = crit.op(rows.to(dev), targets, (h, w), scale)
    assert abs(float(items[0]) - float(g[tag + "_loss"])) < 1e-5 * abs(float(g[tag + "_loss"])), (float(items[0]), float(g[tag + "_loss"]))
    got = dpred.float().cpu() / scale
    want = torch.from_numpy(g[tag + "_grad"]).reshape(B, h * w, nc + 4)
    back = torch.cat((got[..., :nc], got[..., nc_pad:nc_pad + 2], got[..., nc_pad + 8:nc_pad + 10]), -1)
    assert rel(back, want) < 1e-3, rel(back, want)
    pad = torch.ones(ld, dtype=torch.bool)
    pad[:nc] = pad[nc_pad:nc_pad + 2] = pad[nc_pad + 8:nc_pad + 10] = False
    assert float(got[..., pad].abs().max()) == 0.0
    bad = [t.clone() for t in targets]
    bad[4][0, 0], bad[3][0, 0] = h * w + 5, 1.0
    with pytest.raises(L.CvxError):
        crit.op(rows.to(dev), bad, (h, w), scale)


def test_centernet_trainer_fused_step(dev):
    """export_from_registry("centernet") -> CenterNetTrainer at the config's 384 x 384, batch 4: six fused steps of the reference's
    train_loop (centernet_train.py:104-121) on a repeated batch: losses finite and falling, parameters and BatchNorm statistics move,
    no overflow skip; the fused path and loss.backward() on the model's output give the same gradients; evaluate_loop's metric."""
    import builder
    from core.trainer.centernet_train import SyntheticCenterNetLoader
    cfg, algo_cls, trainer_cls = builder.export_from_registry("centernet")
    cfg.train.batch_size = 4
    torch.manual_seed(0)
    loader = SyntheticCenterNetLoader(4, (384, 384), cfg.dataset.num_classes, length=2, seed=3)
    tr = trainer_cls(cfg, dev, dataloader=loader)
    assert type(tr.criterion).__name__ == "CenterNetLoss"
    batch = next(iter(loader))
    # (a) autograd path == fused path: loss.backward() on model(x) fills the same gradient arena
    tr.model.train()
    x, targets = batch[0].to(dev), [t.to(dev) for t in batch[1]]
    loss = tr.criterion(tr.model(x), targets)
    loss.backward()
    g_auto = tr.model.flat_grads.clone()
    tr.model.flat_grads.zero_()
    tr.model.flat_stats.copy_(torch.zeros_like(tr.model.flat_stats))
    with torch.no_grad():
        for k, v in tr.model.state_dict().items():
            if k.endswith("running_var"):
                v.fill_(1.0)
    p0 = tr.model.flat_params.clone()
    losses = [float(tr.train_loop(batch, None)[0]) for _ in range(6)]
    torch.cuda.synchronize()
    assert abs(losses[0] - float(loss.detach())) < 2e-3 * abs(losses[0])          # (BatchNorm saw the same batch: same forward)
    assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    assert float(g_auto.abs().max()) > 0 and not torch.equal(tr.model.flat_params, p0)
    tr._step.scaler.poll()
    assert tr._step.scaler.skipped == 0 and tr.optimizer.device_step() == 6 and not tr.criterion.bad_targets()
    ev = tr.evaluate_loop()
    assert np.isfinite(ev["val_loss"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_multibox_loss_kernel_matches_the_reference_fixture(dev, gold, tag):
    """cvx_multibox_loss against the REAL reference's MultiBoxLossV2 + torch autograd (make_golden.py section 12c): random (loc, conf),
    (a) positives present -- batch-wide hard-negative mining, k = 3 x positives; (b) no positive anywhere -- the 100-negatives branch.
    The three loss values 1e-5; the gradients w.r.t. loc and conf 1e-5 (fp32, the radix select takes the same anchors as torch.topk)."""
    from computervision.pytorch_amd.ssd import MultiBoxLoss
    g = gold("ssd_loss.npz")
    nc = int(g["nc"])
    crit = MultiBoxLoss(float(g["neg_pos"]), nc)
    loc, conf, y = (torch.from_numpy(g[tag + "_" + k]).to(dev) for k in ("loc", "conf", "y"))
    items, dloc, dconf = crit.op(loc, conf, y)
    np.testing.assert_allclose(items.cpu().numpy(), g[tag + "_items"], rtol=1e-5, atol=1e-7)
    assert rel(dloc.cpu(), torch.from_numpy(g[tag + "_dloc"])) < 1e-5 or float(np.abs(g[tag + "_dloc"]).max()) == 0.0
    assert float((dloc.cpu() - torch.from_numpy(g[tag + "_dloc"])).abs().max()) < 1e-7
    assert rel(dconf.cpu(), torch.from_numpy(g[tag + "_dconf"])) < 1e-5
    # through autograd on leaf tensors: the criterion's three return values and .backward()
    l2, c2 = loc.clone().requires_grad_(True), conf.clone().requires_grad_(True)
    total, l_loss, c_loss = crit(y_true=y, y_pred=(l2, c2))
    total.backward()
    assert abs(float(total.detach()) - float(g[tag + "_items"][0])) < 1e-5 * abs(float(g[tag + "_items"][0]))
    assert rel(c2.grad.cpu(), torch.from_numpy(g[tag + "_dconf"])) < 1e-5


@pytest.mark.parametrize("B,A,nc", [(3, 1000, 20), (2, 777, 80), (1, 300, 4)])
def test_multibox_loss_kernel_against_the_oracle(dev, B, A, nc):
    """cvx_multibox_loss against oracle/ssd_ref.multibox_loss + torch autograd on seeded inputs: anchor counts that do not fill the last
    256-anchor workgroup, 80 classes (rows too large for the LDS-staged kernels: the direct form runs), and a run of anchors with
    identical logits spanning several workgroups whose hard-negative key ties at the selection threshold -- the kernel takes ties in
    flat-index order, torch.topk may take others of the same value, so the loss values are compared there, and the gradient where
    no tie exists."""
    from computervision.pytorch_amd.ssd import MultiBoxLoss
    from oracle import ssd_ref as SS
    g = torch.Generator().manual_seed(B * 31 + A + nc)
    y = SS.synth_y_true(B, A, nc, n_pos=10, seed=A)
    loc = torch.randn(B, A, 4, generator=g)
    conf = torch.randn(B, A, nc + 1, generator=g) * 2
    crit = MultiBoxLoss(3.0, nc)
    for ties in (False, True):
        c = conf.clone()
        if ties:                                   # anchors 100..699 of image 0: one logit row -> one key value, far more of them than k needs
            c[0, 100:min(700, A)] = c[0, 100].clone()
            c[0, 100:min(700, A), 1:] += 10.0      # ... and the largest foreground mass, so the threshold falls inside the run
        l2, c2 = loc.clone().requires_grad_(True), c.clone().requires_grad_(True)
        want = SS.multibox_loss(y, l2, c2, 3.0)
        want[0].backward()
        items, dloc, dconf = crit.op(loc.to(dev), c.to(dev), y.to(dev))
        np.testing.assert_allclose(items.cpu().numpy(), np.array([float(v) for v in want]), rtol=2e-5, atol=1e-7)
        assert rel(dloc.cpu(), l2.grad) < 1e-5
        if not ties:
            assert rel(dconf.cpu(), c2.grad) < 1e-5
        else:                                      # same number of anchors taken, all of them from the front of the tied run
            got_taken = (dconf.cpu()[0, 100:min(700, A)].abs().sum(-1) > 0)
            ref_taken = (c2.grad[0, 100:min(700, A)].abs().sum(-1) > 0)
            npos_in_run = int(y[0, 100:min(700, A), -1].sum())
            assert int(got_taken.sum()) == int(ref_taken.sum())
            if npos_in_run == 0:
                n = int(got_taken.sum())
                assert bool(got_taken[:n].all()) and not bool(got_taken[n:].any())


def test_ssd_trainer_fused_step(dev):
    """export_from_registry("ssd") -> SsdTrainer at 300 x 300, batch 4: six fused steps of the reference's train_loop on a repeated
    batch: losses finite and falling, parameters move, no overflow skip; the fused path and (criterion(...)[0]).backward() on the model's
    outputs fill the same gradients; evaluate_loop's metric."""
    import builder
    from core.trainer.ssd_train import SyntheticSsdLoader
    cfg, algo_cls, trainer_cls = builder.export_from_registry("ssd")
    cfg.train.batch_size = 4
    cfg.train.pretrained = False
    torch.manual_seed(0)
    loader = SyntheticSsdLoader(4, (300, 300), cfg.dataset.num_classes, length=2, seed=3)
    tr = trainer_cls(cfg, dev, dataloader=loader)
    assert type(tr.criterion).__name__ == "MultiBoxLoss"
    batch = next(iter(loader))
    tr.model.train()
    x, y = batch[0].to(dev), batch[1].to(dev)
    total, _, _ = tr.criterion(y_true=y, y_pred=tr.model(x))
    total.backward()
    g_auto = tr.model.flat_grads.clone() / tr.model.loss_scale
    tr.model.flat_grads.zero_()
    p0 = tr.model.flat_params.clone()
    losses = [float(tr.train_loop(batch, None)[0]) for _ in range(6)]
    torch.cuda.synchronize()
    assert abs(losses[0] - float(total.detach())) < 5e-3 * abs(losses[0])
    assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    assert float(g_auto.abs().max()) > 0 and not torch.equal(tr.model.flat_params, p0)
    tr._step.scaler.poll()
    assert tr._step.scaler.skipped == 0 and tr.optimizer.device_step() == 6
    ev = tr.evaluate_loop()
    assert np.isfinite(ev["val_loss"])


def test_ssd_target_encoding_matches_the_reference_fixture(dev, gold):
    """cvx_ssd_encode_targets against the REAL reference's Ssd.generate_targets (make_golden.py section 12d) on four label sets in one
    batch: ordinary boxes, a sliver no prior overlaps above the threshold (its best prior is forced), two identical boxes of different
    class (the first wins), an image without boxes.  Positive flags, classes and which ground truth a prior takes: exact; the encoded
    boxes to 2 ulp of float32 (the device's float64 log / divide against numpy's)."""
    import builder
    g = gold("ssd_targets.npz")
    cfg, algo_cls, _ = builder.export_from_registry("ssd")
    algo = algo_cls(cfg, dev)
    assert np.array_equal(algo.anchors, g["anchors"]) and int(g["nc"]) == cfg.dataset.num_classes
    labels = [np.concatenate((np.zeros((int(n), 1), np.float32), g["labels"][i, :int(n)]), 1) for i, n in enumerate(g["counts"])]
    y = algo.encode_targets(labels).cpu().numpy()
    ref = g["y_true"]
    assert y.shape == ref.shape == (4, 8732, 26)
    assert np.array_equal(y[..., 4:], ref[..., 4:])                            # one-hot classes, background column, positive flag
    assert [int(v) for v in ref[..., -1].sum(1)] == [56, 1, 104, 0]
    np.testing.assert_allclose(y[..., :4], ref[..., :4], rtol=3e-7, atol=1e-7)
    one = algo.generate_targets(labels[1]).cpu().numpy()
    assert np.array_equal(one[:, 4:], ref[1][:, 4:])


def test_centernet_target_drawing_matches_the_reference_fixture(dev, gold):
    """cvx_centernet_draw_targets against the REAL reference's CenterNet.generate_targets (make_golden.py section 9d) on four label sets in
    one batch: ordinary boxes, two overlapping boxes of one class (maximum merge), a box at the corner (clipped Gaussian) with a sub-pixel
    box (radius 0), no box.  reg / wh / mask / indices exact; the heat map to 1 ulp of float32 (float64 exp on the device vs numpy)."""
    import builder
    g = gold("centernet_targets.npz")
    cfg, algo_cls, _ = builder.export_from_registry("centernet")
    algo = algo_cls(cfg, dev)
    assert int(g["nc"]) == cfg.dataset.num_classes and int(g["fh"]) == cfg.arch.input_size[1] // cfg.arch.downsampling_ratio
    labels = [np.concatenate((np.zeros((int(n), 1), np.float32), g["labels"][i, :int(n)]), 1) for i, n in enumerate(g["counts"])]
    heat, reg, wh, mask, ind = (t.cpu().numpy() for t in algo.draw_targets(labels))
    assert np.array_equal(reg, g["reg"]) and np.array_equal(wh, g["wh"]) and np.array_equal(mask, g["mask"]) and np.array_equal(ind, g["ind"])
    assert np.array_equal(heat == 1.0, g["heat"] == 1.0) and np.array_equal(heat == 0.0, g["heat"] == 0.0)      # centres and support
    np.testing.assert_allclose(heat, g["heat"], rtol=2e-7, atol=0)
    assert int((g["heat"] == 1.0).sum()) >= 10
    # the drawn targets feed the loss kernel directly
    from computervision.pytorch_amd.dla import CenterNetLoss
    crit = CenterNetLoss(cfg.dataset.num_classes)
    B, h, w, nc = heat.shape
    rows = torch.randn(B, h * w, ((nc + 7) & ~7) + 16, device=dev)
    items, _ = crit.op(rows, algo.draw_targets(labels), (h, w), 1.0)
    assert bool(torch.isfinite(items).all())


def test_yolov7_loss_kernel_matches_the_reference_fixture(dev, gold):
    """cvx_yolo7_loss against the REAL reference's Yolo7Loss + torch autograd (make_golden.py section 11c): random head outputs at 256 x 256,
    an image with six objects (two on top of each other: shared candidate cells), an image with one, an image with none.  Candidate
    generation, SimOTA assignment and the three terms on the device: the four loss values 2e-5, the gradient w.r.t. the head outputs
    1e-3 (fp16 rounding of the scaled gradient; box / class gradients are accumulated with float atomics)."""
    from computervision.pytorch_amd.yolov7 import Yolo7Loss
    g = gold("yolov7_loss.npz")
    nc, (H, W) = int(g["nc"]), (int(v) for v in g["hw"])
    outs = [torch.from_numpy(g[f"out{i}"]) for i in range(3)]
    grads = [torch.from_numpy(g[f"grad{i}"]) for i in range(3)]
    B = outs[0].shape[0]
    level_hw = [tuple(o.shape[2:]) for o in outs]
    A, ld = sum(h * w for h, w in level_hw), 80
    rows = torch.zeros(B, A, ld)
    a_off = 0
    for o, (h, w) in zip(outs, level_hw):
        rows[:, a_off:a_off + h * w, :o.shape[1]] = o.permute(0, 2, 3, 1).reshape(B, h * w, -1)
        a_off += h * w
    crit = Yolo7Loss(None, nc, (H, W))
    scale = 256.0
    items, dpred = crit.op(rows.to(dev), level_hw, torch.from_numpy(g["targets"]), float(H), scale)
    assert crit.overflowed() == 0
    np.testing.assert_allclose(items.cpu().numpy(), g["items"], rtol=2e-5, atol=1e-7)
    got = dpred.float().cpu() / scale
    a_off = 0
    for gr, (h, w) in zip(grads, level_hw):
        want = gr.permute(0, 2, 3, 1).reshape(B, h * w, -1)
        have = got[:, a_off:a_off + h * w, :want.shape[2]]
        assert rel(have, want) < 1e-3, rel(have, want)
        a_off += h * w
    assert float(got[..., 75:].abs().max()) == 0.0
    # no targets at all: only the objectness term
    items0, _ = crit.op(rows.to(dev), level_hw, torch.zeros(0, 6), float(H), scale)
    assert float(items0[1]) == 0.0 and float(items0[3]) == 0.0 and float(items0[2]) > 0.0



def test_yolov7_trainer_fused_step(dev):
    """export_from_registry("yolo7") -> Yolo7Trainer at 640 x 640, batch 4: the reference's train_loop (yolo7_train.py:79-97) as the
    fused step, six times on a repeated batch: the four loss items finite, the total falling, parameters move, no overflow skip, no
    capacity overflow in the assignment; criterion(model(x), targets, x)[0].backward() fills the same kind of gradients; evaluate_loop."""
    import builder
    from core.trainer.yolo7_train import SyntheticYolo7Loader
    cfg, algo_cls, trainer_cls = builder.export_from_registry("yolo7")
    cfg.train.pretrained = False
    cfg.train.batch_size = 4
    torch.manual_seed(0)
    loader = SyntheticYolo7Loader(4, (640, 640), cfg.dataset.num_classes, length=2, seed=3)
    tr = trainer_cls(cfg, dev, dataloader=loader)
    batch = next(iter(loader))
    tr.model.train()
    x, t = batch[0].to(dev), batch[1].to(dev)
    total, box_l, obj_l, cls_l = tr.criterion(tr.model(x), t, x)
    total.backward()
    assert abs(float(total.detach()) - float(box_l + obj_l + cls_l)) < 1e-5 * abs(float(total.detach()))
    g_auto = tr.model.flat_grads.clone()
    assert float(g_auto.abs().max()) > 0 and bool(torch.isfinite(g_auto).all())
    tr.model.flat_grads.zero_()
    p0 = tr.model.flat_params.clone()
    losses = [[float(v) for v in tr.train_loop(batch, None)] for _ in range(6)]
    torch.cuda.synchronize()
    assert abs(losses[0][0] - float(total.detach())) < 2e-2 * abs(losses[0][0])         # (same batch, BatchNorm in batch-statistics mode)
    assert all(np.isfinite(v) for row in losses for v in row) and losses[-1][0] < losses[0][0], losses
    assert not torch.equal(tr.model.flat_params, p0)
    tr._step.scaler.poll()
    assert tr._step.scaler.skipped == 0 and tr.optimizer.device_step() == 6 and tr.criterion.overflowed() == 0
    ev = tr.evaluate_loop()
    assert np.isfinite(ev["val_loss"])


def test_overlapped_exchange_of_the_other_train_steps_is_bit_identical(dev):
    """CenterNet / SSD / YOLOv7 fused steps with an RCCL group of size 1: the backward pass cut into op ranges by
    graph.generic_grad_buckets (from the ops' own parameter offsets), each range's slice of the gradient arena folded and all-reduced on
    a side stream while the next range runs -- the same losses and parameters as the plain step, bit for bit."""
    import torch.distributed as dist
    from computervision.pytorch_amd.dla import CenterNetDLA34, CenterNetLoss, CenterNetTrainStep
    from computervision.pytorch_amd.ssd import MultiBoxLoss, SSD300VGG, SsdTrainStep
    from computervision.pytorch_amd.train import FlatAdam
    from computervision.pytorch_amd.yolov7 import Yolo7L, Yolo7Loss, Yolo7TrainStep
    from core.trainer.centernet_train import SyntheticCenterNetLoader
    from core.trainer.ssd_train import SyntheticSsdLoader
    from core.trainer.yolo7_train import SyntheticYolo7Loader
    created = False
    if not dist.is_initialized():
        try:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29615", rank=0, world_size=1, device_id=dev)
            created = True
        except Exception as exc:
            pytest.skip(f"RCCL process group unavailable: {exc}")
    cases = {
        "centernet": (lambda: CenterNetDLA34(20), lambda m: CenterNetLoss(20), CenterNetTrainStep, SyntheticCenterNetLoader(2, (128, 160), 20, length=1, seed=2)),
        "ssd": (lambda: SSD300VGG(20), lambda m: MultiBoxLoss(3, 20), SsdTrainStep, SyntheticSsdLoader(2, (300, 300), 20, length=1, seed=2)),
        "yolov7": (lambda: Yolo7L(20), lambda m: Yolo7Loss(None, 20, (160, 224)), Yolo7TrainStep, SyntheticYolo7Loader(2, (160, 224), 20, length=1, seed=2)),
    }
    try:
        for name, (make, crit, step_cls, loader) in cases.items():
            images, targets = next(iter(loader))
            images = images.to(dev)
            targets = [t.to(dev) for t in targets] if isinstance(targets, list) else targets.to(dev)
            runs = []
            for distributed in (False, True):
                torch.manual_seed(0)
                m = make().to(dev).train()
                step = step_cls(m, crit(m), FlatAdam(m, lr=1e-3), n_buckets=4)
                step.distributed = distributed
                losses = [step(images, targets).clone() for _ in range(3)]
                torch.cuda.synchronize()
                runs.append((torch.stack(losses).cpu(), m.flat_params.clone().cpu()))
            assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), name
            assert bool(torch.isfinite(runs[0][0]).all()), name
    finally:
        if created:
            dist.destroy_process_group()


def test_loss_kernels_edge_cases(dev):
    """Degenerate inputs of the added loss kernels behave like the reference's torch code: (a) segmentation loss with every pixel
    ignored -- focal: loss 0 and zero gradient (the mean still divides by all pixels), cross-entropy: nan loss (0 / 0, as
    nn.CrossEntropyLoss) and zero gradient; (b) YOLOv7 loss with more ground truths in one image than the assignment kernel holds (64):
    flagged through overflowed(), never silent, result finite; (c) MultiBoxLoss where every anchor of an image is positive (no negatives
    to mine there) against the oracle."""
    from computervision.pytorch_amd.deeplab import SegLoss
    from computervision.pytorch_amd.ssd import MultiBoxLoss
    from computervision.pytorch_amd.yolov7 import Yolo7Loss
    from oracle import ssd_ref as S
    g = torch.Generator().manual_seed(3)
    rows = torch.randn(1, 9 * 9, 24, generator=g).to(dev)
    t = torch.full((1, 33, 33), -100, dtype=torch.long, device=dev)
    for kind in ("focal", "ce"):
        crit = SegLoss(kind)
        crit.nc = 21
        loss, dpred = crit.op(rows, t, (9, 9), 1024.0)
        assert float(dpred.float().abs().max()) == 0.0
        ref = F.cross_entropy(torch.zeros(1, 21, 33, 33), t.cpu(), reduction="mean") if kind == "ce" else torch.zeros(())
        assert (np.isnan(float(loss)) and bool(torch.isnan(ref))) if kind == "ce" else float(loss) == 0.0
    # (b)
    level_hw = [(4, 4), (8, 8), (16, 16)]
    A = sum(h * w for h, w in level_hw)
    rows7 = torch.randn(2, A, 80, generator=g).to(dev)
    tg = torch.zeros(70, 6)
    tg[:, 1] = torch.randint(0, 20, (70,), generator=g).float()
    tg[:, 2:4] = torch.rand(70, 2, generator=g) * 0.8 + 0.1
    tg[:, 4:6] = torch.rand(70, 2, generator=g) * 0.3 + 0.05
    crit7 = Yolo7Loss(None, 20, (128, 128))
    items, dp = crit7.op(rows7, level_hw, tg, 128.0, 64.0)
    assert crit7.overflowed() & 1 and bool(torch.isfinite(items).all()) and bool(torch.isfinite(dp.float()).all())
    items_ok, _ = crit7.op(rows7, level_hw, tg[:20], 128.0, 64.0)
    assert crit7.overflowed() == 0 and bool(torch.isfinite(items_ok).all())
    # (c)
    B, Aa, nc = 2, 300, 20
    y = S.synth_y_true(B, Aa, nc, 10, seed=4)
    y[0, :, 4], y[0, :, 5], y[0, :, -1] = 0.0, 1.0, 1.0            # every anchor of image 0 is a positive of class 1
    y[0, :, 6:-1] = 0.0
    loc, conf = torch.randn(B, Aa, 4, generator=g), torch.randn(B, Aa, nc + 1, generator=g) * 2
    want = S.multibox_loss(y, loc.clone().requires_grad_(True), conf.clone().requires_grad_(True), 3.0)
    items, _, _ = MultiBoxLoss(3.0, nc).op(loc.to(dev), conf.to(dev), y.to(dev))
    np.testing.assert_allclose(items.cpu().numpy(), [float(v) for v in want], rtol=2e-5)


# ---- tile-resident chains (csrc/conv_chain.hip): eval-mode cross-layer fusion ------------------------------------------------
# Since round 4 the chain kernel is in the TUNING build only (tools/build_tuning.sh; it measured slower than the per-layer launches): these
# tests run under CVX_LIB=build/libcvx_tuning.so and skip on the release library, which test_release_library_refuses_fusion covers.
def _needs_chain():
    if not L.has_chain():
        pytest.skip("chain kernel: tuning build only (CVX_LIB=build/libcvx_tuning.so)")


def _chain_conv_ref(x_nhwc, w_okkc, k, stride=1):
    return F.conv2d(x_nhwc.float().permute(0, 3, 1, 2), w_okkc.float().permute(0, 3, 1, 2), stride=stride, padding=k // 2).permute(0, 2, 3, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,C,th,tw,shortcut", [(2, 40, 40, 64, 10, 20, True), (1, 24, 40, 32, 8, 8, False), (3, 20, 20, 128, 5, 10, True),
                                                    (2, 30, 50, 64, 12, 16, True), (2, 16, 48, 16, 8, 16, True)])
def test_chain_bottleneck_pair_unit(dev, B, H, W, C, th, tw, shortcut):
    """Bottleneck (modules.py:124-135, eval) as ONE launch vs torch fp32 on the same fp16 operands (the intermediate rounded to fp16 like the
    per-layer path); ragged tilings (tiles overhanging the image, 24 = 3 x 8) included."""
    _needs_chain()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, C, generator=g).half()
    w1, w2 = [(torch.randn(C, 3, 3, C, generator=g) * (9 * C) ** -0.5).half() for _ in range(2)]
    sc1, sc2 = torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5
    sh1, sh2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    mid = F.silu(_chain_conv_ref(x, w1, 3) * sc1 + sh1).half()
    ref = F.silu(_chain_conv_ref(mid, w2, 3) * sc2 + sh2) + (x.float() if shortcut else 0)
    d = [t.to(dev) for t in (x, w1, sc1, sh1, w2, sc2, sh2)]
    out = torch.zeros(B, H, W, C, dtype=torch.float16, device=dev)
    L.check(L.load().cvx_chain_pair_unit(L.ptr(d[0]), B, H, W, C, L.ptr(d[1]), L.ptr(d[2]), L.ptr(d[3]), L.ptr(d[4]), L.ptr(d[5]), L.ptr(d[6]),
                                         int(shortcut), L.ptr(out), th, tw, 1, None, L.stream_ptr(dev)), "chain pair")
    torch.cuda.synchronize()
    assert float((out.float().cpu() - ref).abs().max() / ref.abs().max()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Ci,Co,k,s,up,th,tw", [(2, 40, 40, 64, 64, 3, 1, 0, 10, 20), (2, 40, 40, 64, 128, 3, 2, 0, 10, 10),
                                                      (2, 20, 20, 64, 32, 1, 1, 0, 10, 20), (2, 40, 40, 48, 64, 1, 1, 1, 10, 20),
                                                      (1, 36, 28, 32, 80, 3, 1, 0, 12, 14)])
def test_chain_single_conv_unit(dev, B, H, W, Ci, Co, k, s, up, th, tw):
    """one conv + folded BN + SiLU as a one-stage chain: 1x1 / 3x3, stride 1 / 2, and the nearest-2x upsample folded into the load
    (yolo_v8.py:39-41) -- x is then the HALF-resolution tensor"""
    _needs_chain()
    g = torch.Generator().manual_seed(1)
    xs = torch.randn(B, H // 2 if up else H, W // 2 if up else W, Ci, generator=g).half()
    x = xs.repeat_interleave(2, 1).repeat_interleave(2, 2) if up else xs
    w = (torch.randn(Co, k, k, Ci, generator=g) * (k * k * Ci) ** -0.5).half()
    sc, sh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    ref = F.silu(_chain_conv_ref(x, w, k, s) * sc + sh)
    d = [t.to(dev) for t in (xs, w, sc, sh)]
    out = torch.zeros(B, H // s, W // s, Co, dtype=torch.float16, device=dev)
    L.check(L.load().cvx_chain_conv_unit(L.ptr(d[0]), B, H, W, Ci, L.ptr(d[1]), Co, k, s, up, L.ptr(d[2]), L.ptr(d[3]), 0, L.ptr(out), th, tw, 1, None,
                                         L.stream_ptr(dev)), "chain conv")
    torch.cuda.synchronize()
    assert float((out.float().cpu() - ref).abs().max() / ref.abs().max()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,th,tw", [(2, 16, 24, 64, 8, 16), (2, 40, 40, 128, 8, 10), (2, 20, 20, 256, 5, 10)])
def test_chain_detect_level_unit(dev, B, H, W, Cin, th, tw):
    """one Detect level (modules.py:428-433, train-mode rows) as ONE launch: rows outside the level stay untouched"""
    _needs_chain()
    cb, cc, ncp = 64, 80, 80
    g = torch.Generator().manual_seed(2)
    mk = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).half()  # noqa: E731
    x = mk(B, H, W, Cin)
    wa = mk(cb + cc, 3, 3, Cin, scale=(9 * Cin) ** -0.5)
    wb1, wb2 = mk(cb, 3, 3, cb, scale=(9 * cb) ** -0.5), mk(cc, 3, 3, cc, scale=(9 * cc) ** -0.5)
    wo1, wo2 = mk(64, 1, 1, cb, scale=cb ** -0.5), mk(ncp, 1, 1, cc, scale=cc ** -0.5)
    sca, scb = torch.rand(cb + cc, generator=g) + 0.5, torch.rand(cb + cc, generator=g) + 0.5
    sha, shb = torch.randn(cb + cc, generator=g) * 0.1, torch.randn(cb + cc, generator=g) * 0.1
    bias = torch.randn(64 + ncp, generator=g)
    a = F.silu(_chain_conv_ref(x, wa, 3) * sca + sha).half()
    hb = F.silu(_chain_conv_ref(a[..., :cb], wb1, 3) * scb[:cb] + shb[:cb]).half()
    hc = F.silu(_chain_conv_ref(a[..., cb:], wb2, 3) * scb[cb:] + shb[cb:]).half()
    ref = torch.cat([_chain_conv_ref(hb, wo1, 1) + bias[:64], _chain_conv_ref(hc, wo2, 1) + bias[64:]], -1).reshape(B, H * W, 64 + ncp)
    d = [t.to(dev) for t in (x, wa, sca, sha, wb1, wb2, scb, shb, wo1, wo2, bias)]
    A = H * W + 64
    pred = torch.full((B, A, 64 + ncp), 7.0, device=dev)
    L.check(L.load().cvx_chain_detect_unit(L.ptr(d[0]), B, H, W, Cin, cb, cc, ncp, L.ptr(d[1]), L.ptr(d[2]), L.ptr(d[3]), L.ptr(d[4]), L.ptr(d[5]), L.ptr(d[6]),
                                           L.ptr(d[7]), L.ptr(d[8]), L.ptr(d[9]), L.ptr(d[10]), L.ptr(pred), A, 32, th, tw, 1, None, L.stream_ptr(dev)), "chain detect")
    torch.cuda.synchronize()
    assert float((pred[:, 32:32 + H * W].cpu() - ref).abs().max() / ref.abs().max()) < 1e-3
    assert bool((pred[:, :32] == 7.0).all()) and bool((pred[:, 32 + H * W:] == 7.0).all())


@pytest.mark.gpu
def test_inference_keeps_its_weight_shadows_only_while_the_weights_stand(dev):
    """Engine.forward lets the engine skip the fp32 -> fp16 weight conversion (cvx_engine_keep_shadows) when neither torch's version counter of
    the parameter arena nor the package's counter of raw-pointer writers moved since the engine's previous forward: same output bit for bit
    with the shadows kept; an in-place torch update and an Adam kernel step must both be seen by the very next forward."""
    m = new_model(dev).eval()
    x = synth.images(2, 128, 128, seed=4).to(dev)
    with torch.no_grad():
        y0, _ = m(x)
        y1, _ = m(x)                                  # second forward: shadows kept
        assert torch.equal(y0, y1)
        m.flat_params.mul_(2.0)                       # torch sees this write (version counter of the arena)
        y2, _ = m(x)
        assert not torch.equal(y2, y1)
        m.flat_params.mul_(0.5)                       # exact inverse
        y3, _ = m(x)
        assert torch.equal(y3, y0)
        y3b, _ = m(x)                                 # kept again
        assert torch.equal(y3b, y0)
        g = torch.full_like(m.flat_params, 1e-2)      # torch does not see this one: the Adam kernel writes through a raw pointer
        mm, vv = torch.zeros_like(g), torch.zeros_like(g)
        E.adam_step(m.flat_params, g, mm, vv, 1e-2, (0.9, 0.999), 1e-8, 1)
        y4, _ = m(x)
        assert not torch.equal(y4, y0)
        y5, _ = m(x)
        assert torch.equal(y5, y4)


@pytest.mark.gpu
def test_release_library_refuses_fusion(dev):
    """the release library has no chain kernel: enabling the fusion groups is an error that says where the kernel went, never a silent no-op"""
    if L.has_chain():
        pytest.skip("tuning build")
    m = new_model(dev).eval()
    with torch.no_grad():
        m(synth.images(1, 64, 64, seed=0).to(dev))
    eng = m._last_engine
    eng.set_fusion(False)
    assert eng.fused_groups() == 0
    with pytest.raises(L.CvxError, match="tuning"):
        eng.set_fusion(True)


@pytest.mark.gpu
def test_eval_forward_with_fused_groups_matches_the_per_layer_path(dev, gold):
    """the engine's eval forward with the fusion groups (Bottleneck pairs, Detect levels) against the same forward layer by layer, and
    both against the reference fixture: 128x128 (the 4x4 level stays unfused: no feasible tile) and 320x320 (all group kinds)"""
    _needs_chain()
    g = gold("yolov8n_fwd_128.npz")
    m = new_model(dev).eval()
    for x in (torch.from_numpy(g["x"]), synth.images(2, 320, 320, seed=3)):
        with torch.no_grad():
            y0, _ = m(x.to(dev))  # the default: layer by layer (the groups measured 1-6 % slower at batch 32)
            eng = m._last_engine
            assert eng.fused_groups() == 0
            eng.set_fusion(True)
            y1, _ = m(x.to(dev))
            n = eng.fused_groups()
            eng.set_fusion(False)
            assert eng.fused_groups() == 0
            eng.set_fusion(True)
            y2, _ = m(x.to(dev))
            eng.set_fusion(False)
        assert n >= 6, n
        assert torch.equal(y1, y2)
        # same operands, same fp16 rounding points, another summation order inside a K loop: fp32 round-off through the decode
        np.testing.assert_allclose(y1.cpu().numpy(), y0.cpu().numpy(), rtol=2e-3, atol=2e-3)
        if x.shape[-1] == 128:
            np.testing.assert_allclose(y1.cpu().numpy(), g["eval_y"], rtol=5e-3, atol=5e-3)


# ---- BASELINE configs 4 and 5 at their stated batch ------------------------------------------------------------------------------
@pytest.mark.gpu
def test_centernet_config4_at_batch_64(dev):
    """BASELINE.json configs[3]: CenterNet 512x512 inference, batch 64, forward + heat-map decode through the plugin API: finite outputs,
    count bounds, and per-image independence at the full batch (image 37 of the batch == the same image alone, bit for bit)."""
    import builder
    m = _centernet(dev)
    x = synth.images(64, 512, 512, seed=5).to(dev)
    with torch.no_grad():
        raw = m.forward_raw(x)
        one = m.forward_raw(x[37:38])
    assert raw.shape[0] == 64 and bool(torch.isfinite(raw).all()) and torch.equal(raw[37:38], one)
    cfg, algo_cls, _ = builder.export_from_registry("centernet")
    cfg.dataset.num_classes = 80
    cfg.arch.input_size = (3, 512, 512)
    algo = algo_cls(cfg, dev)
    d_all, d_one = algo.decode_raw(raw, 128, 128), algo.decode_raw(one, 128, 128)
    counts = d_all["counts"].cpu()
    assert counts.shape[0] == 64 and bool((counts >= 0).all()) and bool((counts <= 100).all())
    n = int(d_one["counts"][0])
    assert n == int(counts[37]) and torch.equal(d_all["topk_index"][37], d_one["topk_index"][0]) and torch.equal(d_all["keep"][37, :n], d_one["keep"][0, :n])
    for k in ("boxes", "scores"):
        assert bool(torch.isfinite(d_all[k]).all())


@pytest.mark.gpu
def test_deeplab_config5_at_batch_16(dev):
    """BASELINE.json configs[4]: DeepLabv3+ 513x513 train step, batch 16: fused steps of the registered trainer on one repeated batch --
    finite losses that fall, no overflow skip of the dynamic loss scale, every BatchNorm updated."""
    import builder
    from core.trainer.segmentation_trainer import SyntheticSegmentationLoader
    cfg, _, trainer_cls = builder.export_from_registry("deeplabv3plus")
    assert cfg.train.batch_size == 16
    torch.manual_seed(0)
    loader = SyntheticSegmentationLoader(16, (513, 513), 21, length=1, seed=4)
    tr = trainer_cls(cfg, dev, dataloader=loader)
    batch = next(iter(loader))
    tr.model.train()
    rv0 = tr.model.flat_stats.clone()
    losses = [float(tr.train_loop(batch, None)[0]) for _ in range(3)]
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    tr._step.scaler.poll()
    assert tr._step.scaler.skipped == 0 and tr.optimizer.device_step() == 3
    assert not torch.equal(tr.model.flat_stats, rv0) and bool(torch.isfinite(tr.model.flat_params).all())


# ---- gradient exchange behind the C ABI (csrc/comm.hip) ---------------------------------------------------------------------------
@pytest.mark.gpu
def test_c_abi_exchange_with_one_rank_is_bit_identical_to_plain_backward(dev):
    """cvx_engine_backward_exchange (ranges, slab folds and RCCL all-reduces enqueued from C, communicator created through
    cvx_comm_unique_id / cvx_comm_create) with a communicator of ONE rank: the step must give exactly the plain step's losses and
    parameters; cvx_allreduce_grads (whole arena) leaves a gradient arena unchanged at world 1."""
    from computervision.pytorch_amd.train import CvxComm, FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    x, batch = synth.images(2, 128, 128, seed=1).to(dev), {k: v.to(dev) for k, v in synth.targets(2, seed=2).items()}
    try:
        comm = CvxComm(dev, rank=0, world=1)
    except L.CvxError as exc:                                        # no RCCL library on this box
        pytest.skip(f"RCCL unavailable: {exc}")
    runs = []
    for c in (None, comm):
        m = new_model(dev).train()
        step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), FlatAdam(m, lr=1e-3), n_buckets=4, comm=c)
        losses = [step(x, batch).clone() for _ in range(3)]
        torch.cuda.synchronize()
        runs.append((torch.stack(losses).cpu(), m.flat_params.clone().cpu()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    g = torch.randn(100003, device=dev)
    g0 = g.clone()
    comm.all_reduce_(g)
    torch.cuda.synchronize()
    assert torch.equal(g, g0)
    m = new_model(dev).train()
    crit = V8DetectionLoss(Yolo8DetConfig(), m)
    pred = m._run_forward(x, training=True)
    from computervision.pytorch_amd.train import flatten_targets
    _, dpred = crit.op(pred, flatten_targets(batch, dev), m.level_shapes(128, 128), (8, 16, 32), crit.loss_scale)
    eng = m.engine_for(128, 128)
    eng.backward(dpred, crit.loss_scale)
    torch.cuda.synchronize()
    before = m.flat_grads.clone()
    L.check(L.load().cvx_allreduce_grads(eng.handle, comm.handle, L.stream_ptr(dev)), "cvx_allreduce_grads")
    torch.cuda.synchronize()
    assert torch.equal(m.flat_grads, before)


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one process per GPU over RCCL)")
@pytest.mark.parametrize("exchange", ["c", "torch"])
def test_two_rank_rccl_step_against_the_dp_fixture(gold, tmp_path, exchange):
    """Two FRESH child processes (tools/dp_rccl_child.py; never a re-exec of a process that touched the GPU), one per GPU, each with its
    half of the fixture batch: the gradients after the overlapped RCCL exchange (C ABI path and torch.distributed path) are the mean of
    the per-shard gradients -- equal on both ranks, and the reference's shard-by-shard mean (tests/golden/dp_sim_96.npz) within the
    end-to-end bound of the single-GPU simulation test."""
    import subprocess
    import sys as _sys
    port = 29650 + (0 if exchange == "c" else 1)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([_sys.executable, os.path.join(ROOT, "tools", "dp_rccl_child.py"), str(r), "2", str(port), str(tmp_path), exchange], env=env)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["grads"], r1["grads"])
    f = gold("dp_sim_96.npz")
    from computervision.pytorch_amd.graph import ParamLayout
    mean = ParamLayout("n", 80).views(torch.from_numpy(r0["grads"]))
    flat = torch.cat([mean[str(k)].flatten() for k in f["keys"]])
    ref = torch.from_numpy(f["w2_sub"])
    assert float((flat[::211] - ref).norm() / ref.norm()) < 1.6e-1
    assert abs(float(flat.norm()) / float(f["w2_norm"]) - 1) < 5e-2
