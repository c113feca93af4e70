"""Per-kernel parity: HIP kernels (through the C ABI) vs the fp32 torch ops the oracle is built
from.  Inputs are rounded to bf16 first so the comparison measures kernel arithmetic
(bf16 operands, fp32 accumulate, one rounding on output) -- tolerance 2^-7 relative to the
output scale."""
import math

import pytest
import torch
from _report import report
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REL = 2.0 ** -7


def bfr(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rel_err(got, want):
    return float((got.cpu() - want).abs().max() / (want.abs().max() + 1e-12))


@pytest.fixture(scope="module")
def ops():
    from agenda_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("B,Cin,H,Cout,k,stride,up", [
    (2, 64, 16, 64, 3, 1, False),
    (2, 128, 16, 320, 3, 1, False),      # N tail (320 = 2.5 tiles of 128)
    (1, 64, 8, 128, 3, 2, False),        # downsample
    (1, 64, 8, 64, 3, 1, True),          # nearest-2x upsample fused in the gather
    (2, 4, 16, 64, 3, 1, False),         # conv_in: 4 channels zero-padded to 64
    (1, 64, 16, 4, 3, 1, False),         # conv_out: N=4
    (1, 192, 12, 64, 1, 1, False),       # 1x1
    (8, 320, 32, 320, 3, 1, False),      # 128x128-tile path (many tiles)
    (3, 64, 9, 64, 3, 1, False),         # ragged M (243 rows)
    (8, 1280, 8, 1280, 3, 1, False),     # split-K path (M=512, K=11520)
    (2, 640, 16, 1280, 3, 1, False),     # split-K, M=512, ragged K split
    (8, 320, 32, 320, 3, 2, False),      # stride-2 at 128-tile size
])
def test_conv2d(ops, B, Cin, H, Cout, k, stride, up):
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + k)
    x = bfr(torch.randn(B, Cin, H, H, generator=g))
    w = bfr(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g) * 0.1
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    want = F.conv2d(xi, w, b, stride=stride, padding=1 if k == 3 else 0)
    got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), stride=stride, upsample=up)
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-4      # fp32 output, bf16-exact inputs: only accumulation order differs


@pytest.mark.parametrize("M,K,N,geglu,res", [
    (256, 320, 320, False, True),
    (616, 768, 640, False, False),       # M = 8*77 ragged
    (512, 320, 2560, True, False),       # GEGLU
    (100, 64, 128, True, True),
    (2, 1280, 1280, False, False),       # tiny M
    (4096, 1280, 320, False, True),
    (8192, 320, 320, False, True),       # BN=160 tile path
    (8192, 320, 960, False, False),      # BN=160, fused qkv width
    (512, 5120, 1280, False, True),      # split-K linear: 64 tiles of 64x160 x 4 K slices
    (2048, 1280, 1280, False, True),     # 256 tiles of 64x160 on the deep ring (one workgroup per CU)
    (2048, 5120, 1280, False, True),     # the same tile, 80 K steps, unsplit
    (1984, 1280, 1280, False, False),    # M % 64 == 0 but not % 128: 248 tiles of 64x160
    (2048, 1280, 1120, False, True),     # N = 7 x 160
    (2048, 1280, 3840, False, False),    # weights outweigh activations: W-major tile walk
    (2048, 1280, 10240, True, False),    # W-major + GEGLU ([8 values | 8 gates] row groups)
    (8192, 2560, 640, False, True),      # long-K 1x1 -> 256 tiles of 128x160
    (300, 128, 200, False, True),        # ragged M and N tail (N % 16 != 0: element-wise epilogue on the last lanes)
    (130, 64, 72, False, False),         # N < one tile, 8-column remainder
])
def test_linear(ops, M, K, N, geglu, res):
    g = torch.Generator().manual_seed(M + K + N)
    x = bfr(torch.randn(M, K, generator=g))
    w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    y = F.linear(x, w, b)
    if geglu:
        val, gate = y.chunk(2, dim=-1)
        y = val * F.gelu(gate)
    r = bfr(torch.randn(M, y.shape[1], generator=g)) if res else None
    if res:
        y = y + r
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None, geglu=geglu)
    assert rel_err(got, y) < 1e-4


@pytest.mark.parametrize("B,Cin,Cout", [(8, 1280, 1280), (8, 128, 64), (3, 320, 128), (16, 192, 192), (1, 64, 64)])
def test_conv3x3_whole_images_resident_kernel_8x8(ops, B, Cin, Cout):
    """igemm_smap.h: 3x3 / stride 1 / pad 1 on 8 x 8 maps -- all rows x 64 output channels x a slice of the input channels per workgroup,
    the eight images with their zero border in one LDS image per 64-channel chunk, every weight tile streamed once (split-K slabs + the
    ordered slab pass behind it).  vs F.conv2d fp32 and vs the default kernels: full tile (UNet batch 8), one chunk (no split), fewer
    than eight images, two row tiles, a single image."""
    g = torch.Generator().manual_seed(B * 7 + Cin + Cout)
    x = bfr(torch.randn(B, Cin, 8, 8, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    b = torch.randn(Cout, generator=g) * 0.1
    want = F.conv2d(x, w, b, padding=1)
    got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), smap=True)
    ref = ops.conv2d(x.cuda(), w.cuda(), b.cuda())
    assert rel_err(got, want) < 1e-4, rel_err(got, want)
    assert rel_err(got, ref.cpu()) < 1e-4
    assert torch.equal(got, ops.conv2d(x.cuda(), w.cuda(), b.cuda(), smap=True))        # ordered slab sum: run-to-run identical


@pytest.mark.parametrize("M,K,N,geglu,res", [
    (2048, 1280, 1280, False, True),     # 16 x 16 maps' C -> C projection + residual
    (8192, 640, 1920, False, False),     # 32 x 32 qkv
    (2048, 1280, 2560, True, False),     # GEGLU (256-wide tile, [8 values | 8 gates] row groups)
    (1000, 192, 384, False, True),       # ragged M (1000 = 15 x 64 + 40), odd stage count (3 stages of 64)
    (512, 5120, 1280, False, True),      # long K, few tiles
])
def test_linear_weight_streaming_kernel(ops, M, K, N, geglu, res):
    """igemm_wreg.h: the weight matrix in MFMA fragment order straight to registers, activations register-staged through LDS; behind it the
    same register epilogue as the other igemm kernels.  vs fp32 torch (bf16 output: 2^-7) and vs the launcher's default kernel."""
    g = torch.Generator().manual_seed(M + K + N)
    x = bfr(torch.randn(M, K, generator=g))
    w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    y = F.linear(x, w, b)
    if geglu:
        val, gate = y.chunk(2, dim=-1)
        y = val * F.gelu(gate)
    r = bfr(torch.randn(M, y.shape[1], generator=g)) if res else None
    if res:
        y = y + r
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None, geglu=geglu, wreg=True)
    ref = ops.linear(x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None, geglu=geglu)
    assert rel_err(got, y) < REL
    assert rel_err(got, ref.cpu()) < REL
    assert torch.equal(got, ops.linear(x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None, geglu=geglu, wreg=True))


@pytest.mark.parametrize("M,K,N,res", [
    (2048, 1280, 1280, True),      # 16 x 16 maps C -> C: 64 x 160 tiles, two 2-slot rings
    (2048, 1280, 1280, False),
    (512, 1280, 1280, True),       # 8 x 8 maps: 64 x 64 tiles, two 4-slot rings
    (512, 2560, 1280, False),
    (512, 1344, 1280, False),      # odd number of K steps (21): group 1 runs one step on a zero-filled slot
    (500, 1280, 640, True),        # ragged M
    (64, 512, 64, False),          # one tile, K shorter than the rings
])
def test_linear_two_k_groups_per_workgroup(ops, M, K, N, res):
    """igemm_kernel KG = 2 (the one-workgroup-per-CU 1x1 launches of the small maps): two groups of four waves walk alternate K steps through their own
    LDS rings and accumulators, group 1 hands its sums to group 0 through LDS.  vs fp32 torch, vs the single-group kernel (same products, another
    summation order: fp32 noise), and run-to-run identical."""
    g = torch.Generator().manual_seed(M + K + N)
    x = bfr(torch.randn(M, K, generator=g))
    w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    r = bfr(torch.randn(M, N, generator=g)) if res else None
    y = F.linear(x, w, b) + (r if res else 0)
    args = (x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None)
    got = ops.linear(*args, kgroups=True)
    ref = ops.linear(*args)
    assert rel_err(got, y) < 1e-4, rel_err(got, y)            # fp32 output of bf16-exact operands
    assert rel_err(got, ref.cpu()) < 1e-4
    assert torch.equal(got, ops.linear(*args, kgroups=True))
    if N % 128 == 0 and (K // 64) % 2 == 0 and K >= 512:
        # the same decomposition on the weight-streaming kernel (igemm_wreg.h KG = 2; bf16 output)
        g2 = ops.linear(*args, wreg=True, kgroups=True)
        assert rel_err(g2, y) < REL and rel_err(g2, ops.linear(*args, wreg=True).cpu()) < REL
        assert torch.equal(g2, ops.linear(*args, wreg=True, kgroups=True))


@pytest.mark.parametrize("M,K,N,res", [(2048, 1280, 1280, True), (2048, 1280, 1280, False), (2048, 6400, 1280, True), (2048, 640, 1280, False), (1536, 1024, 1280, True),
                                        (2000, 1280, 1280, True), (8192, 3200, 640, True), (8192, 1280, 640, False), (8100, 2560, 640, True)])
def test_linear_producer_consumer_kernel(ops, M, K, N, res):
    """igemm_pc.h (round 5): the 1x1 launches of the 16 x 16 maps (M = 2048, N = 1280: 256 tiles of 64 x 160, one workgroup per CU) with the roles split over the waves of a
    workgroup -- four loader waves issue the ring's LDS-DMA pieces, four consumer waves read fragments and run the MFMAs.  Same tiles, fragment layout and epilogue as the
    4-wave kernel: vs fp32 torch, and bit-identical to the 4-wave kernel's result (the same products summed in the same order); short K (fewer steps than ring stages + 1),
    a K that is not a multiple of the ring depth, and a ragged M tail.  M = 8192 / 8100, N = 640: the 128 x 160-tile form (igemm_pc bit 5: 256 tiles, one per CU)."""
    g = torch.Generator().manual_seed(M + K + N)
    x = bfr(torch.randn(M, K, generator=g))
    w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    r = bfr(torch.randn(M, N, generator=g)) if res else None
    y = F.linear(x, w, b) + (r if res else 0)
    args = (x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None)
    got = ops.linear(*args, pc=33)
    ref = ops.linear(*args)
    assert rel_err(got, y) < 1e-4, rel_err(got, y)            # fp32 output of bf16-exact operands
    assert torch.equal(got, ref)
    assert torch.equal(got, ops.linear(*args, pc=33))


@pytest.mark.parametrize("kind,args", [("lin", (2048, 1280, 1280)), ("lin", (2048, 6400, 1280)), ("lin", (8192, 640, 1920)), ("lin", (512, 1280, 1280)), ("lin", (2048, 1280, 3840)),
                                       ("conv", (8, 16, 1280, 1280)), ("conv", (8, 32, 640, 640)), ("conv", (4, 64, 320, 320)), ("conv", (8, 8, 1280, 1280)), ("conv", (3, 16, 192, 320))])
def test_xcd_tile_blocks_do_not_change_results(ops, kind, args):
    """`xcd_block` (igemm.hip pick_xcd_block): the tile grid cut into one block of tiles per XCD so that the XCD's L2 working set is smallest -- the same tiles computed by
    other workgroups, so every output must be BIT-identical to the default walk's (4-wave, row-halo, 8-phase, whole-images and producer / consumer kernels, split-K included),
    and every tile must still be computed exactly once (vs fp32 torch)."""
    g = torch.Generator().manual_seed(sum(args))
    if kind == "lin":
        M, K, N = args
        x = bfr(torch.randn(M, K, generator=g)); w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K)); b = torch.randn(N, generator=g) * 0.1
        want = F.linear(x, w, b)
        for kw in ({}, {"pc": 1}, {"p8": 1}):
            a0 = ops.linear(x.cuda(), w.cuda(), b.cuda(), **kw)
            a1 = ops.linear(x.cuda(), w.cuda(), b.cuda(), xcd_block=True, **kw)
            assert rel_err(a1, want) < 1e-4 and torch.equal(a0, a1), kw
    else:
        B, H, Cin, Cout = args
        x = bfr(torch.randn(B, Cin, H, H, generator=g)); w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)); b = torch.randn(Cout, generator=g) * 0.1
        want = F.conv2d(x, w, b, padding=1)
        for kw in ({}, {"halo": True}, {"halo": True, "smap": True}, {"p8": 1}):
            a0 = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), **kw)
            a1 = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), xcd_block=True, **kw)
            assert rel_err(a1, want) < 1e-4 and torch.equal(a0, a1), kw


@pytest.mark.parametrize("B,H,Cin,Cout,pc", [(8, 16, 1280, 1280, 2), (8, 16, 1280, 1280, 4), (8, 16, 640, 1280, 2), (2, 32, 128, 1280, 2), (8, 8, 1280, 1280, 8), (8, 8, 2560, 1280, 8),
                                              (6, 16, 192, 1280, 2)])
def test_conv3x3_producer_consumer_kernel(ops, B, H, Cin, Cout, pc):
    """igemm_pc.h on 3x3 stride-1 convs of the small maps: the loader waves compute the im2col offsets (borders through the buffer range check) of every (tap, chunk) step and
    issue the LDS-DMA pieces, the consumer waves run the MFMAs.  pc = 2: 64 x 160 tiles unsplit (256 workgroups at 16 x 16, UNet batch 8); 4: 128 x 160 tiles x 2 K slices;
    8: the 8 x 8 maps on 64 x 160 tiles x 4 K slices.  vs F.conv2d in fp32 on bf16-exact operands, and run-to-run identical (ordered slab sum)."""
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x = bfr(torch.randn(B, Cin, H, H, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    b = torch.randn(Cout, generator=g) * 0.1
    want = F.conv2d(x, w, b, padding=1)
    got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), pc=pc)
    assert rel_err(got, want) < 1e-4, rel_err(got, want)
    assert torch.equal(got, ops.conv2d(x.cuda(), w.cuda(), b.cuda(), pc=pc))


@pytest.mark.parametrize("B,H,Cin,Cout", [(8, 32, 640, 640), (8, 32, 1280, 640), (8, 16, 1280, 1280), (2, 32, 320, 640), (1, 64, 128, 200), (3, 16, 192, 320), (1, 128, 64, 160)])
def test_conv3x3_row_halo_producer_consumer_kernel(ops, B, H, Cin, Cout):
    """igemm_pch.h (igemm_pc bit 4): the row-halo 3x3 kernel's tiles and operand scheme with loader / consumer waves, one workgroup per CU -- unsplit (256 tiles at 32 x 32,
    UNet batch 8) and split-K (16 x 16); fewer tiles, a ragged N tile (200 = 160 + 40), a ragged M tile (3 x 256 pixels = 6 tiles), whole 128-pixel rows.
    vs F.conv2d in fp32 on bf16-exact operands, BIT-IDENTICAL to igemm_halo_kernel (same tiles, fragments and summation order) and run-to-run identical."""
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x = bfr(torch.randn(B, Cin, H, H, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    b = torch.randn(Cout, generator=g) * 0.1
    want = F.conv2d(x, w, b, padding=1)
    got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), halo=True, pc=16)
    assert rel_err(got, want) < 1e-4, rel_err(got, want)
    assert torch.equal(got, ops.conv2d(x.cuda(), w.cuda(), b.cuda(), halo=True))
    assert torch.equal(got, ops.conv2d(x.cuda(), w.cuda(), b.cuda(), halo=True, pc=16))


@pytest.mark.parametrize("M", [128, 1000, 4096 * 2])      # one tile; ragged tail (1000 = 7 x 128 + 104); many tiles
def test_ff_fused_matches_torch_and_the_unfused_kernels(ops, M):
    """tblock.hip ff_fused_kernel (norm3 -> GEGLU -> ff.net.2 + residual in one launch, the hidden activation never in HBM) vs fp32 torch
    on bf16-exact inputs, and vs the chain of kernels it replaces (layer_norm -> linear GEGLU -> linear + residual).  The fused path
    rounds the hidden activation to bf16 once (as the unfused path does when it stores it) and the output once."""
    C = 320
    g = torch.Generator().manual_seed(M)
    x = bfr(torch.randn(M, C, generator=g) * 1.5 + 0.3)
    ga, be = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.2
    w1 = bfr(torch.randn(8 * C, C, generator=g) / math.sqrt(C)); b1 = torch.randn(8 * C, generator=g) * 0.1
    w2 = bfr(torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)); b2 = torch.randn(C, generator=g) * 0.1
    val, gate = F.linear(F.layer_norm(x, (C,), ga, be, 1e-5), w1, b1).chunk(2, dim=-1)
    want = x + F.linear(val * F.gelu(gate), w2, b2)
    cu = lambda t: t.cuda()
    got = ops.ff_fused(cu(x), cu(ga), cu(be), cu(w1), cu(b1), cu(w2), cu(b2))
    hid = ops.linear(ops.layer_norm(cu(x), cu(ga), cu(be)), cu(w1), cu(b1), geglu=True)
    unf = ops.linear(hid, cu(w2), cu(b2), cu(x))
    e_t, e_u = rel_err(got, want), rel_err(got, unf.cpu())
    print(f"ff_fused M={M}: vs torch {e_t:.5f}, vs unfused kernels {e_u:.5f}")
    report(f"op_ff_fused[M={M}]", max_rel_vs_torch=e_t, max_rel_vs_unfused=e_u)
    assert e_t < REL, e_t                # bf16 hidden activation + bf16 output against fp32 torch
    assert e_u < REL, e_u
    assert torch.equal(got, ops.ff_fused(cu(x), cu(ga), cu(be), cu(w1), cu(b1), cu(w2), cu(b2)))      # run-to-run identical


@pytest.mark.parametrize("B,HW,T,C,r32", [(1, 128, 77, 320, 0), (2, 256, 77, 320, 0), (2, 4096, 77, 320, 0), (1, 1024, 96, 320, 0), (1, 384, 33, 320, 0),
                                          (1, 64, 77, 640, 0), (2, 1024, 77, 640, 0), (1, 192, 50, 640, 0),
                                          (1, 64, 77, 640, 1), (2, 1024, 77, 640, 1), (1, 96, 50, 640, 1), (8, 1024, 77, 640, 1)])      # r32: the 32-row panel form (round 5)
def test_attn_chain_fused_matches_torch(ops, B, HW, T, C, r32):
    """tblock.hip attn_chain_kernel (norm2 -> to_q -> cross-attention -> to_out + residual, with the head-summed probability side output
    of the DAAM recorder) vs fp32 torch on bf16-exact inputs: the op sequence of data_generation/hook.py:91-120 (explicit softmax) behind
    a LayerNorm, SD-1.5's 64 x 64 (C = 320, 8 heads of 40: 128-row panels) and 32 x 32 (C = 640, 8 heads of 80: 64-row panels) block shapes."""
    H = 8
    D = C // H
    g = torch.Generator().manual_seed(B * 1000 + HW + T + C)
    x = bfr(torch.randn(B, HW, C, generator=g) * 1.2 + 0.2)
    ga, be = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.2
    wq = bfr(torch.randn(C, C, generator=g) / math.sqrt(C)); wo = bfr(torch.randn(C, C, generator=g) / math.sqrt(C))
    bo = torch.randn(C, generator=g) * 0.1
    kv = bfr(torch.randn(B, T, 2 * C, generator=g))
    qh = F.linear(F.layer_norm(x, (C,), ga, be, 1e-5), wq).reshape(B, HW, H, D).permute(0, 2, 1, 3)
    kh = kv[..., :C].reshape(B, T, H, D).permute(0, 2, 1, 3); vh = kv[..., C:].reshape(B, T, H, D).permute(0, 2, 1, 3)
    P = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(D), dim=-1)                  # [B, H, HW, T]
    want = x + F.linear((P @ vh).permute(0, 2, 1, 3).reshape(B, HW, C), wo, bo)
    want_p = P.sum(1).transpose(1, 2)                                                    # [B, T, HW]
    cu = lambda t: t.cuda()
    got, pr = ops.attn_chain(cu(x), cu(ga), cu(be), cu(wq), cu(kv), cu(wo), cu(bo), heads=H, return_probs=True, rows32=bool(r32))
    e_y, e_p = rel_err(got, want), float((pr.cpu() - want_p).abs().max())
    print(f"attn_chain C={C} B={B} HW={HW} T={T} rows32={r32}: out {e_y:.5f}, head-summed probabilities max abs {e_p:.5f}")
    report(f"op_attn_chain[C={C},B={B},HW={HW},T={T},rows32={r32}]", out_max_rel=e_y, head_summed_probs_max_abs=e_p)
    assert e_y < REL, e_y
    assert e_p < 8 * 2e-3, e_p                             # sum of 8 heads' probabilities (2e-3 each: bf16 Q / K operands)
    got2, pr2 = ops.attn_chain(cu(x), cu(ga), cu(be), cu(wq), cu(kv), cu(wo), cu(bo), heads=H, return_probs=True, rows32=bool(r32))
    assert torch.equal(got, got2) and torch.equal(pr, pr2)                               # run-to-run identical (no atomics)


@pytest.mark.parametrize("B,HW,T,C", [(2, 256, 77, 1280), (1, 64, 77, 1280), (2, 256, 50, 1280), (1, 128, 80, 1280), (3, 64, 1, 1280), (2, 1024, 77, 640), (1, 256, 33, 640)])
def test_xattn_premul_matches_torch(ops, B, HW, T, C):
    """xattn_pre.hip (attn2 of the C = 1280 blocks against per-image pre-multiplied context matrices: K'' = gamma scale (k Wq), V'' = Wo v built first, then
    S = folded-LayerNorm(x) K''^T -> softmax (+ per-head recorder rows) -> P V''^T + bo + x as two GEMMs) vs fp32 torch on bf16-exact inputs: the op sequence of
    data_generation/hook.py:91-120 (explicit softmax) behind a LayerNorm, at SD-1.5's 16 x 16 / 8 x 8 block shape (C = 1280, 8 heads of 160), with fewer
    tokens than 77 (padded columns masked), the full 80, and a single token (softmax of one column = 1)."""
    H = 8                                                  # (C = 640: head dim 80 -- a ragged last 32-deep step in the context products; option attn2_premul bit 1)
    D = C // H
    g = torch.Generator().manual_seed(B * 1000 + HW + T + C)
    x = bfr(torch.randn(B, HW, C, generator=g) * 1.2 + 0.2)
    ga, be = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.2
    wq = bfr(torch.randn(C, C, generator=g) / math.sqrt(C)); wo = bfr(torch.randn(C, C, generator=g) / math.sqrt(C))
    bo = torch.randn(C, generator=g) * 0.1
    kv = bfr(torch.randn(B, T, 2 * C, generator=g))
    qh = F.linear(F.layer_norm(x, (C,), ga, be, 1e-5), wq).reshape(B, HW, H, D).permute(0, 2, 1, 3)
    kh = kv[..., :C].reshape(B, T, H, D).permute(0, 2, 1, 3); vh = kv[..., C:].reshape(B, T, H, D).permute(0, 2, 1, 3)
    P = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(D), dim=-1)                  # [B, H, HW, T]
    want = x + F.linear((P @ vh).permute(0, 2, 1, 3).reshape(B, HW, C), wo, bo)
    want_p = P.transpose(2, 3)                                                           # [B, H, T, HW]
    cu = lambda t: t.cuda()
    got, pr = ops.xattn_premul(cu(x), cu(ga), cu(be), cu(wq), cu(kv), cu(wo), cu(bo), heads=H, return_probs=True)
    e_y, e_p = rel_err(got, want), float((pr.cpu() - want_p).abs().max())
    print(f"xattn_premul C={C} B={B} HW={HW} T={T}: out {e_y:.5f}, probabilities max abs {e_p:.5f}")
    report(f"op_xattn_premul[C={C},B={B},HW={HW},T={T}]", out_max_rel=e_y, probs_max_abs=e_p)
    assert e_y < REL, e_y
    assert e_p < 2e-3, e_p                                 # the bound of the attention kernel's recorder (bf16 operands of the score GEMM)
    assert float((pr.sum(2) - 1).abs().max()) < 1e-5       # every row of every head sums to one: probability mass is conserved in the recorder
    got2, pr2 = ops.xattn_premul(cu(x), cu(ga), cu(be), cu(wq), cu(kv), cu(wo), cu(bo), heads=H, return_probs=True)
    assert torch.equal(got, got2) and torch.equal(pr, pr2)                               # run-to-run identical (no atomics)


@pytest.mark.parametrize("B,C,H,groups,silu,eps", [
    (2, 320, 16, 32, True, 1e-5), (1, 1920, 8, 32, True, 1e-5), (2, 2560, 4, 32, False, 1e-6),
    (1, 128, 64, 32, True, 1e-6), (2, 64, 8, 32, False, 1e-5), (1, 960, 16, 32, True, 1e-5),
])
def test_groupnorm(ops, B, C, H, groups, silu, eps):
    g = torch.Generator().manual_seed(C + H)
    x = bfr(torch.randn(B, C, H, H, generator=g) * 2 + 0.5)
    ga = torch.randn(C, generator=g) * 0.2 + 1
    be = torch.randn(C, generator=g) * 0.2
    want = F.group_norm(x, groups, ga, be, eps)
    if silu:
        want = F.silu(want)
    got = ops.group_norm(x.cuda(), groups, ga.cuda(), be.cuda(), eps, silu)
    assert rel_err(got, want) < REL


@pytest.mark.parametrize("B,Cin,H,Cout", [(2, 64, 16, 320), (8, 320, 32, 320), (2, 128, 16, 640), (1, 64, 8, 1280), (3, 64, 16, 128)])
def test_groupnorm_statistics_from_the_conv_epilogue(ops, B, Cin, H, Cout):
    """The production GroupNorm path: the conv launch leaves per-(M tile, channel) partial sums of its bf16 outputs, the
    GroupNorm kernel reduces them instead of re-reading the activation.  vs torch (conv -> bf16 -> group_norm -> silu) and vs
    the two-kernel GroupNorm on the same conv output: the statistics are the same sums in another order, so the two HIP
    paths agree except for isolated 1-ulp bf16 flips."""
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = bfr(torch.randn(B, Cin, H, H, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    b = torch.randn(Cout, generator=g) * 0.1
    ga = torch.randn(Cout, generator=g) * 0.2 + 1
    be = torch.randn(Cout, generator=g) * 0.2
    h = bfr(F.conv2d(x, w, b, padding=1))
    want = F.silu(F.group_norm(h, 32, ga, be, 1e-5))
    fused = ops.conv_groupnorm(x.cuda(), w.cuda(), b.cuda(), ga.cuda(), be.cuda(), fused=True)
    sep = ops.conv_groupnorm(x.cuda(), w.cuda(), b.cuda(), ga.cuda(), be.cuda(), fused=False)
    assert rel_err(fused, want) < REL and rel_err(sep, want) < REL
    d = (fused - sep).abs().cpu()
    assert float(d.max()) <= float(want.abs().max()) * 2.0 ** -7                      # at most a bf16 ulp of the largest value
    assert float((d > 0).float().mean()) < 0.02                                       # ... and only on isolated elements
    if B * H * H >= 2048 and Cout % 20 == 0:          # the same through the 8-phase kernel: partial sums per 256-row tile
        f8 = ops.conv_groupnorm(x.cuda(), w.cuda(), b.cuda(), ga.cuda(), be.cuda(), fused=True, p8=3)
        assert rel_err(f8, want) < REL
        assert float((f8 - sep).abs().max()) <= float(want.abs().max()) * 2.0 ** -7


@pytest.mark.parametrize("rows,C", [(64, 320), (10, 1280), (7, 64), (33, 640)])
def test_layernorm(ops, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = bfr(torch.randn(rows, C, generator=g) * 3 + 1)
    ga = torch.randn(C, generator=g) * 0.2 + 1
    be = torch.randn(C, generator=g) * 0.2
    want = F.layer_norm(x, (C,), ga, be, 1e-5)
    got = ops.layer_norm(x.cuda(), ga.cuda(), be.cuda())
    assert rel_err(got, want) < REL


def _attn_ref(q, k, v, heads):
    from oracle import sd_oracle as O
    d = q.shape[-1] // heads
    qh, kh, vh = (O.head_to_batch_dim(t, heads) for t in (q, k, v))
    p = O.attention_scores(qh, kh, d ** -0.5)
    return O.batch_to_head_dim(torch.bmm(p, vh), heads), p


@pytest.mark.parametrize("B,H,D,Nq,Nk", [
    (2, 8, 40, 256, 256), (1, 8, 80, 64, 64), (1, 8, 160, 64, 64), (2, 2, 64, 128, 128),
    (1, 2, 32, 256, 256), (1, 4, 40, 144, 144),   # ragged query/key tails (768-px mid block)
    (1, 8, 40, 1024, 1024), (1, 2, 128, 64, 64),
    (1, 8, 80, 256, 256), (1, 4, 40, 96, 192), (1, 2, 64, 200, 320),   # several whole 64-key tiles with ragged query tiles
])
def test_self_attention(ops, B, H, D, Nq, Nk):
    g = torch.Generator().manual_seed(D + Nq)
    q, k, v = (bfr(torch.randn(B, n, H * D, generator=g)) for n in (Nq, Nk, Nk))
    want, _ = _attn_ref(q, k, v, H)
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    assert rel_err(got, want) < REL


def test_self_attention_online_softmax_rescale(ops):
    """Force the running max to jump at a later KV tile (a spike key) so the rescale branch runs."""
    g = torch.Generator().manual_seed(5)
    B, H, D, N = 1, 2, 64, 256
    q, k, v = (bfr(torch.randn(B, N, H * D, generator=g)) for _ in range(3))
    k[:, 200] = q[:, 7] * 4.0          # key 200 (4th tile) dominates query 7
    k[:, 100] = q[:, 9] * 3.0
    want, _ = _attn_ref(q, k, v, H)
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    assert rel_err(got, want) < REL


@pytest.mark.parametrize("B,H,D,Nq,Nk", [(2, 8, 40, 256, 77), (2, 8, 80, 64, 77), (2, 8, 160, 16, 77), (2, 2, 64, 64, 77),
                                          (2, 4, 40, 144, 77), (1, 2, 32, 64, 20)])
def test_cross_attention_with_probs(ops, B, H, D, Nq, Nk):
    g = torch.Generator().manual_seed(D + Nq + Nk)
    q = bfr(torch.randn(B, Nq, H * D, generator=g))
    k, v = (bfr(torch.randn(B, Nk, H * D, generator=g)) for _ in range(2))
    want, p = _attn_ref(q, k, v, H)                       # p [B*H, Nq, Nk]
    got, probs = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, return_probs=True)
    assert rel_err(got, want) < REL
    want_p = p.reshape(B, H, Nq, Nk).permute(0, 1, 3, 2)  # token-major [B,H,Nk,Nq]
    assert float((probs.cpu() - want_p).abs().max()) < 2e-3
    assert float((probs.cpu().sum(2) - 1).abs().max()) < 1e-3   # rows of P sum to 1


@pytest.mark.parametrize("B,H,D,Nq,Nk", [(2, 8, 40, 256, 77), (2, 8, 80, 200, 77), (2, 8, 160, 16, 77), (4, 5, 64, 130, 77), (1, 2, 32, 64, 20)])
def test_cross_attention_head_summed_probs(ops, B, H, D, Nq, Nk):
    """The recording form of the daam layers at latent resolution: one workgroup walks all heads of its query tile and adds the
    head SUM of the probabilities once; the per-head outputs must be those of the per-head kernel."""
    g = torch.Generator().manual_seed(3 * D + Nq)
    q = bfr(torch.randn(B, Nq, H * D, generator=g))
    k, v = (bfr(torch.randn(B, Nk, H * D, generator=g)) for _ in range(2))
    want, p = _attn_ref(q, k, v, H)                       # p [B*H, Nq, Nk]
    got, psum = ops.attention_headsum(q.cuda(), k.cuda(), v.cuda(), H)
    assert rel_err(got, want) < REL
    want_p = p.reshape(B, H, Nq, Nk).sum(1).permute(0, 2, 1)      # [B, Nk, Nq]
    assert float((psum.cpu() - want_p).abs().max()) < 2e-3 * H
    assert float((psum.cpu().sum(1) - H).abs().max()) < 1e-3 * H  # every head's row of P sums to 1
    got1, probs = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, return_probs=True)
    assert torch.equal(got, got1)                                  # same arithmetic per head
    assert float((psum - probs.sum(1)).abs().max()) < 1e-5


def test_bicubic_clamp_mean_matches_torch(ops):
    g = torch.Generator().manual_seed(3)
    for side in (8, 16, 32, 64):
        m = torch.rand(5, 7, side, side, generator=g)
        m[1] = (m[1] > 0.9).float() * 4      # undershoot -> clamp matters
        want = torch.stack([F.interpolate(x[None], size=(64, 64), mode="bicubic")[0].clamp_(min=0) for x in m]).mean(0)
        got = ops.bicubic_clamp_mean(m.cuda(), 64)
        assert float((got.cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("B,C,H,Co", [(8, 128, 64, 160), (8, 192, 32, 320), (8, 128, 16, 1280), (1, 128, 512, 128), (2, 128, 128, 256),
                                      (8, 320, 64, 320), (3, 192, 64, 200), (16, 256, 16, 640), (8, 512, 16, 1280)])
def test_conv3x3_row_halo_kernel(ops, B, C, H, Co):
    """igemm_halo.h: one LDS image of the tile's pixel rows (+ one halo pixel each side) per (ky, channel chunk) serves the three kx
    taps.  Shapes that dispatch to it with two workgroups per CU, with the deep weight ring (<= 256 tiles), with split-K (last case),
    with row segments (W > 128), whole rows (W = 128) and several rows per tile, an M tail and an N tail."""
    g = torch.Generator().manual_seed(B * 1000 + C + H + Co)
    x = bfr(torch.randn(B, C, H, H, generator=g))
    w = bfr(torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5))
    b = torch.randn(Co, generator=g)
    want = F.conv2d(x, w, b, padding=1)
    got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), halo=True).cpu()
    ref = ops.conv2d(x.cuda(), w.cuda(), b.cuda()).cpu()                 # the general tap-by-tap kernel
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) / scale < 2e-6                # fp32 accumulation of exact bf16 products
    assert float((got - ref).abs().max()) / scale < 1e-5 and not torch.equal(got, ref)   # a different K order: not the same kernel


@pytest.mark.parametrize("B,C,H,Co", [(8, 128, 32, 160), (8, 192, 16, 320), (2, 128, 128, 256), (8, 64, 8, 1280)])
def test_conv3x3_row_halo_kernel_upsampled(ops, B, C, H, Co):
    """The same kernel on the nearest-2x upsampled input (Upsample2D.conv, folded into the gather): halo columns / rows map to
    source pixel (row >> 1, col >> 1)."""
    g = torch.Generator().manual_seed(B * 77 + C + H + Co)
    x = bfr(torch.randn(B, C, H, H, generator=g))
    w = bfr(torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5))
    b = torch.randn(Co, generator=g)
    want = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), upsample=True, halo=True).cpu()
    ref = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), upsample=True).cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) / scale < 2e-6
    assert float((got - ref).abs().max()) / scale < 1e-5


@pytest.mark.parametrize("B,C,H,Co,p8", [(8, 640, 32, 640, 0), (8, 640, 32, 640, 1), (8, 1280, 16, 1280, 1), (4, 256, 64, 256, 1), (2, 128, 32, 160, 0), (3, 192, 24, 256, 0), (1, 64, 16, 128, 1), (8, 1280, 8, 1280, 1), (8, 320, 8, 160, 0)])      # the last two: few source rows -> 64 x 160 tiles
def test_upsampling_conv_as_four_phase_convs(ops, B, C, H, Co, p8):
    """Upsample2D.conv -- conv3x3(nearest2x(x)) -- as four 2x2 convs on the un-upsampled map, one per output phase (2 i + a, 2 j + b), with the taps that coincide
    pre-summed (IgemmP::ups4: one launch of the general igemm over N = 4 Cout, the epilogue writes the pixel-shuffled rows): 4/9 of the MACs.  vs F.conv2d on the
    upsampled input (bf16 output and bf16-rounded merged weights: 2^-7) and vs the production upsampling kernel; an M tail (B = 3, 24 x 24) and the borders of every phase."""
    g = torch.Generator().manual_seed(B * 13 + C + H + Co)
    x = bfr(torch.randn(B, C, H, H, generator=g))
    w = bfr(torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5))
    b = torch.randn(Co, generator=g)
    want = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), upsample=True, phases=True, p8=p8).cpu()      # p8 = 1: the 8-phase kernel where the launcher takes it (256-row tiles inside one phase, >= 256 of them)
    ref = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), upsample=True, halo=True).cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) / scale < REL, float((got - want).abs().max()) / scale
    assert float((got - ref).abs().max()) / scale < REL
    assert torch.equal(got, ops.conv2d(x.cuda(), w.cuda(), b.cuda(), upsample=True, phases=True, p8=p8).cpu())


def test_igemm_random_shape_sweep(ops):
    """Randomised conv / linear shapes: tails in M and N, every (ksize, stride, upsample) combination the path uses,
    channel counts that are / are not multiples of 64, and sizes on both sides of the tile and split-K thresholds."""
    import random
    rnd = random.Random(1234)
    g = torch.Generator().manual_seed(99)
    n_conv = n_lin = 0
    for _ in range(28):
        k = rnd.choice([1, 3, 3])
        stride, up = rnd.choice([(1, False), (1, False), (2, False), (1, True)]) if k == 3 else (1, False)
        B = rnd.choice([1, 2, 3, 8])
        H = rnd.choice([5, 8, 12, 16, 24, 32]) if not up else rnd.choice([4, 8, 16])
        if stride == 2 and H % 2:
            H += 1
        Cin = rnd.choice([4, 64, 128, 192, 320, 640, 1280])
        Cout = rnd.choice([3, 4, 64, 100, 128, 160, 320, 640])
        if B * H * H * Cin * 9 > 6e7:
            Cin = 64
        x = bfr(torch.randn(B, Cin, H, H, generator=g))
        w = bfr(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
        b = torch.randn(Cout, generator=g) * 0.1
        xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
        want = F.conv2d(xi, w, b, stride=stride, padding=1 if k == 3 else 0)
        got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), stride=stride, upsample=up)
        assert got.shape == want.shape, (B, Cin, H, Cout, k, stride, up)
        assert rel_err(got, want) < 1e-4, (B, Cin, H, Cout, k, stride, up, rel_err(got, want))
        n_conv += 1
    for _ in range(16):
        M = rnd.choice([1, 7, 77, 128, 200, 616, 1000, 2048, 4100])
        K = rnd.choice([64, 320, 768, 1280, 4096, 5120])
        N = rnd.choice([8, 64, 160, 320, 960, 1280])
        geglu = rnd.random() < 0.25
        if geglu:
            N = rnd.choice([128, 256, 2560])
        res = rnd.random() < 0.5
        x = bfr(torch.randn(M, K, generator=g))
        w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K))
        b = torch.randn(N, generator=g) * 0.1
        y = F.linear(x, w, b)
        if geglu:
            val, gate = y.chunk(2, dim=-1)
            y = val * F.gelu(gate)
        r = bfr(torch.randn(M, y.shape[1], generator=g)) if res else None
        if res:
            y = y + r
        got = ops.linear(x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None, geglu=geglu)
        assert rel_err(got, y) < 1e-4, (M, K, N, geglu, res, rel_err(got, y))
        n_lin += 1
    assert n_conv == 28 and n_lin == 16


@pytest.mark.parametrize("B,Cin,H,Cout,k,stride,up,p8", [
    (8, 128, 32, 256, 3, 1, False, 2),     # 256 x 256 tiles, 32 M tiles x 1
    (8, 128, 32, 320, 3, 1, False, 3),     # 256 x 160 tiles (4 x 2 waves, 12 + 8 MFMA phases)
    (3, 192, 20, 160, 3, 1, False, 3),     # M tail (1200 rows), odd number of K tiles (27)
    (3, 64, 20, 512, 3, 1, False, 2),      # nk = 9 (odd), M tail
    (2, 64, 32, 256, 3, 2, False, 2),      # stride 2
    (2, 128, 16, 320, 3, 1, True, 3),      # nearest-2x upsample folded into the gather
    (2, 64, 32, 96, 3, 1, False, 2),       # N tail inside the only N tile (96 of 256 columns)
    (4, 320, 32, 640, 1, 1, False, 3),     # 1x1: im2col row = pixel
    (2, 64, 64, 1280, 1, 1, False, 2),     # one K tile only (nk = 1)
    (2, 128, 64, 400, 1, 1, False, 3),     # N tail across tiles (400 = 2.5 x 160), nk = 2
])
def test_igemm8p_conv(ops, B, Cin, H, Cout, k, stride, up, p8):
    """igemm8p.h (256-row tiles, 8 waves, 8-phase schedule) forced through the op entry point, against F.conv2d and against the
    4-wave kernels: 3x3 / 1x1, stride, upsample, M and N tails, odd / tiny K-tile counts, both tile widths."""
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + k + p8)
    x = bfr(torch.randn(B, Cin, H, H, generator=g))
    w = bfr(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g) * 0.1
    xi = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    want = F.conv2d(xi, w, b, stride=stride, padding=1 if k == 3 else 0)
    got = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), stride=stride, upsample=up, p8=p8).cpu()
    ref = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), stride=stride, upsample=up).cpu()
    assert got.shape == want.shape
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) / scale < 2e-6                # fp32 accumulation of exact bf16 products
    assert float((got - ref).abs().max()) / scale < 1e-5


@pytest.mark.parametrize("M,K,N,geglu,res,p8", [
    (4096, 1280, 320, False, True, 3),
    (2048, 320, 2560, True, False, 2),       # GEGLU ([8 values | 8 gates] row groups in a lane's 16 columns)
    (1000, 640, 5120, True, False, 2),       # GEGLU, M tail
    (8192, 320, 960, False, False, 3),
    (777, 2560, 640, False, True, 3),        # ragged M, residual, 40 K tiles
    (4096, 256, 512, False, True, 2),
    (300, 64, 1280, False, False, 2),        # nk = 1
])
def test_igemm8p_linear(ops, M, K, N, geglu, res, p8):
    g = torch.Generator().manual_seed(M + K + N + p8)
    x = bfr(torch.randn(M, K, generator=g))
    w = bfr(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    y = F.linear(x, w, b)
    if geglu:
        val, gate = y.chunk(2, dim=-1)
        y = val * F.gelu(gate)
    r = bfr(torch.randn(M, y.shape[1], generator=g)) if res else None
    if res:
        y = y + r
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None, geglu=geglu, p8=p8)
    ref = ops.linear(x.cuda(), w.cuda(), b.cuda(), r.cuda() if res else None, geglu=geglu)
    assert rel_err(got, y) < 1e-4
    assert rel_err(got, ref.cpu()) < 1e-4
