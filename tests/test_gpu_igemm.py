"""GPU parity of the implicit-GEMM engine (through the C ABI) against torch fp32 references on the same inputs, for the bf16
MFMA path and the exact-f32 MFMA path. Tolerance: accumulation is fp32 in both, so against an fp32 reference computed from the
same (bf16-rounded) inputs the error is summation order only: <= 2e-3 of max|ref| for fp32 outputs; bf16 outputs add one
rounding (2^-8 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16, F32 = 0, 1


@pytest.fixture(autouse=True, params=[0, 1, 2, 3, 4], ids=["auto", "w128", "w256x128", "w256", "narrow"])
def tile_policy(request):
    """Every case runs under the automatic tile choice and with each tile family forced (include/clite.h: clite_set_tile_policy), so every
    instantiation of the wide-K 8-wave kernels and of the 4-wave kernels meets the same references. f32 launches ignore the policy."""
    dt = request.node.callspec.params.get("dt", BF16) if hasattr(request.node, "callspec") else BF16
    if dt == F32 and request.param != 0:
        pytest.skip("f32 launches do not depend on the tile policy")
    from clip_lite_amd import hip
    hip.set_tile_policy(request.param)
    yield request.param
    hip.set_tile_policy(0)


def _hip():
    from clip_lite_amd import hip
    return hip


def _rel(got, ref):
    return ((got.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


def _t(shape, g, dt, scale=1.0):
    x = torch.randn(*shape, device="cuda", generator=g) * scale
    return x.bfloat16() if dt == BF16 else x


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 104), (300, 64, 32), (70, 1000, 200), (3840, 768, 768), (128, 2048, 2048)])
def test_gemm_nt_bias_relu(dt, M, N, K):
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A, B = _t((M, K), g, dt), _t((N, K), g, dt)
    bias = torch.randn(N, device="cuda", generator=g)
    out, pre = torch.empty_like(A[:, :1].expand(M, N).contiguous()), torch.empty(M, N, device="cuda", dtype=A.dtype)
    hip.gemm_nt(dt, A, B, M, N, K, hip.epilogue(out, N, bias=bias, act=hip.ACT_RELU, preact=pre))
    ref = A.float() @ B.float().t() + bias
    tol = 6e-3 if dt == BF16 else 1e-4
    assert _rel(pre, ref) < tol and _rel(out, ref.relu()) < tol


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 104), (300, 64, 40), (3840, 768, 3072)])
def test_gemm_nn_and_strided_rows(dt, M, N, K):
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A, B = _t((M, K), g, dt), _t((K, N), g, dt)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    hip.gemm_nn(dt, A, B, M, N, K, hip.epilogue(out, N))
    assert _rel(out, A.float() @ B.float()) < 2e-3
    # every 3rd row of A through lda (BERT pooler reads h[:, 0] this way), output scattered through ldc
    Ms = M // 3
    out2 = torch.zeros(Ms, 2 * N, device="cuda", dtype=torch.float32)
    hip.gemm_nn(dt, A, B, Ms, N, K, hip.epilogue(out2, 2 * N), lda=3 * K)
    assert _rel(out2[:, :N], A[::3][:Ms].float() @ B.float()) < 2e-3 and not out2[:, N:].any()


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (72, 136, 300), (256, 8, 1000), (768, 3072, 3840)])
def test_gemm_tn_atomic(dt, M, N, K):
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A, B = _t((K, M), g, dt), _t((K, N), g, dt)
    out = torch.ones(M, N, device="cuda", dtype=torch.float32)
    hip.gemm_tn(dt, A, B, M, N, K, hip.epilogue(out, N, atomic=True))
    assert _rel(out, 1 + A.float().t() @ B.float()) < 2e-3


CONV_CASES = [
    (2, 8, 8, 32, 64, 3, 3, 1, 1), (2, 9, 7, 64, 32, 3, 3, 2, 1), (3, 6, 6, 64, 128, 1, 1, 1, 0),
    (2, 8, 8, 32, 64, 1, 1, 2, 0), (1, 10, 10, 32, 160, 3, 3, 1, 1),
    (8, 56, 56, 64, 64, 3, 3, 1, 1), (8, 56, 56, 256, 128, 1, 1, 1, 0), (8, 28, 28, 128, 128, 3, 3, 2, 1),
    (4, 14, 14, 1024, 2048, 1, 1, 2, 0), (4, 7, 7, 512, 512, 3, 3, 1, 1),
]


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("N,H,W,Cc,K,R,S,st,pad", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dt, N, H, W, Cc, K, R, S, st, pad):
    hip = _hip()
    cv = hip.conv_desc(dt, N, H, W, Cc, K, R, S, st, pad)
    Ho, Wo = cv.Ho, cv.Wo
    g = torch.Generator(device="cuda").manual_seed(N * H + Cc + K)
    x, w, dy = _t((N, H, W, Cc), g, dt), _t((K, R, S, Cc), g, dt, 0.1), _t((N, Ho, Wo, K), g, dt)
    x32 = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    w32 = w.float().permute(0, 3, 1, 2).requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        yref = F.conv2d(x32, w32, stride=st, padding=pad)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    yref = yref.detach().permute(0, 2, 3, 1)

    y = torch.empty(N, Ho, Wo, K, device="cuda", dtype=torch.float32)
    stats = hip.Stats(torch.zeros(8 * 3 * K, device="cuda"), 8, K)
    hip.conv_fwd(x, w, cv, hip.epilogue(y, K, colsum=stats))
    cs = stats.t.view(8, 3, K).sum(0)
    assert _rel(y, yref) < 2e-3
    assert _rel(cs[0], yref.reshape(-1, K).sum(0)) < 2e-3 and _rel(cs[1], (yref.reshape(-1, K) ** 2).sum(0)) < 2e-3
    dx = torch.empty(N, H, W, Cc, device="cuda", dtype=torch.float32)
    hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cc))
    assert _rel(dx, x32.grad.permute(0, 2, 3, 1)) < 2e-3
    # the same input gradient on the transposed weight [C][R][S][K] (clite_conv_dgrad_wt: forward-form GEMM, what the bf16 step runs), with
    # the transposed copy made by the grouped transpose kernel (clite_transpose_weights) and checked against torch's permute
    wt = torch.zeros(Cc, R, S, K, device="cuda", dtype=w.dtype)
    if dt == BF16:
        it = hip.TransposeItem(0, 0, K, Cc, R * S * Cc, R * S * K, R * S, Cc, K, 0)
        items = torch.frombuffer(bytearray(bytes(it)), dtype=torch.uint8).cuda()
        hip.transpose_weights(w, wt, items, 1, R * S * ((K + 63) // 64) * ((Cc + 63) // 64))
        assert torch.equal(wt, w.permute(3, 1, 2, 0).contiguous())
    else:
        wt.copy_(w.permute(3, 1, 2, 0))
    dx2 = torch.full((N, H, W, Cc), 3.0, device="cuda", dtype=torch.float32)
    hip.conv_dgrad(dy, wt, cv, hip.epilogue(dx2, Cc), wt=True)
    assert _rel(dx2, x32.grad.permute(0, 2, 3, 1)) < 2e-3
    if hip.s2_classes_ok(cv):
        dx3 = torch.full((N * H * W, Cc), 5.0, device="cuda", dtype=torch.float32)
        hip.conv_dgrad_s2(dy, wt, cv, lambda: hip.epilogue(dx3, Cc), wt=True)
        assert _rel(dx3.view(N, H, W, Cc), x32.grad.permute(0, 2, 3, 1)) < 2e-3
    dw = torch.zeros(K, R, S, Cc, device="cuda", dtype=torch.float32)
    hip.conv_wgrad(dy, x, cv, dw)
    assert _rel(dw, w32.grad.permute(0, 2, 3, 1)) < 2e-3


@pytest.mark.parametrize("dt", [BF16, F32])
def test_stem_7x7(dt):
    hip = _hip()
    N, H, W = 4, 64, 48
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.randn(N, 3, H, W, device="cuda", generator=g)
    w = torch.randn(64, 7, 7, 3, device="cuda", generator=g) * 0.1
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = H + 6, W + 8
    td = torch.bfloat16 if dt == BF16 else torch.float32
    xpad = torch.empty(N, Hp, Wp, 4, device="cuda", dtype=td)
    hip.image_to_nhwc4(dt, img, xpad, N, H, W, 3, Hp, Wp)
    wv = torch.empty(64, 7, 8, 4, device="cuda", dtype=td)
    hip.stem_pack(dt, w, wv)
    y = torch.empty(N * Ho * Wo, 64, device="cuda", dtype=torch.float32)
    hip.stem_fwd(dt, xpad, wv, N, Hp, Wp, Ho, Wo, hip.epilogue(y, 64))
    imgr = img.to(td).float().requires_grad_(True)
    wr = w.to(td).float().permute(0, 3, 1, 2).requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        ref = F.conv2d(imgr, wr, stride=2, padding=3)
    assert _rel(y.view(N, Ho, Wo, 64), ref.detach().permute(0, 2, 3, 1)) < 2e-3
    dy = torch.randn(N, Ho, Wo, 64, device="cuda", generator=g).to(td)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    dwv = torch.zeros(64, 7, 8, 4, device="cuda")
    hip.stem_wgrad(dt, dy, xpad, N, Hp, Wp, Ho, Wo, dwv)
    dw = torch.zeros(64, 7, 7, 3, device="cuda")
    hip.stem_unpack_grad(dwv, dw)
    assert _rel(dw, wr.grad.permute(0, 2, 3, 1)) < 2e-3


@pytest.mark.policy_independent
@pytest.mark.parametrize("N,H,W", [(16, 224, 224), (3, 70, 96)])
def test_stem_weight_gradient_patch_resident(N, H, W):
    """clite_stem_wgrad_patch (ABI v11) at the benchmark's image size and on an image with an odd number of output rows, against torch's f32 weight
    gradient and clite_stem_wgrad; it accumulates into a non-zero gradient."""
    hip = _hip()
    hip.set_tile_policy(0)          # (a forced tile policy makes the entry point decline)
    g = torch.Generator(device="cuda").manual_seed(H + W)
    img = torch.randn(N, 3, H, W, device="cuda", generator=g)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = H + 6, W + 8
    xpad = torch.empty(N, Hp, Wp, 4, device="cuda", dtype=torch.bfloat16)
    hip.image_to_nhwc4(BF16, img, xpad, N, H, W, 3, Hp, Wp)
    dy = (torch.randn(N, Ho, Wo, 64, device="cuda", generator=g) * 0.1).bfloat16()
    hip.patch_workspace(torch.device("cuda", 0))
    dw = torch.full((64, 7, 7, 3), 0.25, device="cuda")
    assert hip.stem_wgrad_patch(BF16, dy, xpad, N, Hp, Wp, Ho, Wo, dw)
    wr = torch.zeros(64, 3, 7, 7, device="cuda", requires_grad=True)
    with torch.backends.cudnn.flags(enabled=False):
        F.conv2d(img.bfloat16().float(), wr, stride=2, padding=3).backward(dy.float().permute(0, 3, 1, 2))
    assert _rel(dw - 0.25, wr.grad.permute(0, 2, 3, 1)) < 2e-3
    dwv = torch.zeros(64, 7, 8, 4, device="cuda")
    hip.stem_wgrad(BF16, dy, xpad, N, Hp, Wp, Ho, Wo, dwv)
    dw2 = torch.zeros(64, 7, 7, 3, device="cuda")
    hip.stem_unpack_grad(dwv, dw2)
    assert _rel(dw - 0.25, dw2) < 2e-3



@pytest.mark.parametrize("M,N,K", [(200, 136, 104), (3840, 2304, 768), (1000, 64, 72)])
def test_plain_epilogue_bf16_store_bias_and_column_statistics(M, N, K):
    """The branch-free plain epilogue instantiation (bf16 store of alpha*acc + bias, column sums of what was stored): every conv forward
    and the QKV projection take it. Statistics are of the ROUNDED stored values (what the following BatchNorm normalises)."""
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A, B = _t((M, K), g, BF16), _t((N, K), g, BF16)
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    stats = hip.Stats(torch.zeros(8 * 3 * N, device="cuda"), 8, N)
    hip.gemm_nt(BF16, A, B, M, N, K, hip.epilogue(out, N, bias=bias, alpha=0.5, colsum=stats))
    ref = 0.5 * (A.float() @ B.float().t()) + bias
    assert _rel(out, ref) < 6e-3
    cs = stats.t.view(8, 3, N).sum(0)
    of = out.float()
    assert _rel(cs[0], of.sum(0)) < 1e-3 and _rel(cs[1], (of ** 2).sum(0)) < 1e-3


FULL_SIZE_CONVS = [   # ResNet-50 layers at the benchmark's per-GPU batch of 128
    (128, 56, 56, 64, 256, 1, 1, 1, 0), (128, 14, 14, 256, 256, 3, 3, 1, 1), (128, 56, 56, 128, 128, 3, 3, 2, 1), (128, 28, 28, 256, 512, 1, 1, 2, 0),
]


@pytest.mark.parametrize("N,H,W,Cc,K,R,S,st,pad", FULL_SIZE_CONVS)
def test_full_size_conv_adjoint_identities(N, H, W, Cc, K, R, S, st, pad):
    """Size-independent property at BASELINE.json's sizes (no reference convolution is run): forward, dgrad and wgrad are the three
    faces of one trilinear form, <conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(dy, x)>. bf16 operands, f32 outputs: the three values
    agree to summation order."""
    hip = _hip()
    cv = hip.conv_desc(BF16, N, H, W, Cc, K, R, S, st, pad)
    g = torch.Generator(device="cuda").manual_seed(H + Cc + K)
    x, w, dy = _t((N, H, W, Cc), g, BF16), _t((K, R, S, Cc), g, BF16, 0.05), _t((N, cv.Ho, cv.Wo, K), g, BF16)
    y = torch.empty(N, cv.Ho, cv.Wo, K, device="cuda", dtype=torch.float32)
    hip.conv_fwd(x, w, cv, hip.epilogue(y, K))
    dx = torch.empty(N, H, W, Cc, device="cuda", dtype=torch.float32)
    hip.conv_dgrad(dy, w, cv, hip.epilogue(dx, Cc))
    dw = torch.zeros(K, R, S, Cc, device="cuda", dtype=torch.float32)
    hip.conv_wgrad(dy, x, cv, dw)
    a = (y.double() * dy.double()).sum().item()
    b = (dx.double() * x.double()).sum().item()
    c = (dw.double() * w.double()).sum().item()
    scale = (y.double().abs() * dy.double().abs()).sum().item()
    assert abs(a - b) < 1e-5 * scale and abs(a - c) < 1e-5 * scale, (a, b, c, scale)


@pytest.mark.parametrize("M,N,K", [(3840, 3072, 768), (3840, 768, 3072)])
def test_full_size_gemm_adjoint_identities(M, N, K):
    """<A B^T, dC> = <A, dC B> = <B, dC^T A> at the BERT FFN sizes of the benchmark (nt forward, nn input gradient, tn weight gradient)."""
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A, B, dC = _t((M, K), g, BF16), _t((N, K), g, BF16, 0.05), _t((M, N), g, BF16)
    Cm = torch.empty(M, N, device="cuda", dtype=torch.float32)
    hip.gemm_nt(BF16, A, B, M, N, K, hip.epilogue(Cm, N))
    dA = torch.empty(M, K, device="cuda", dtype=torch.float32)
    hip.gemm_nn(BF16, dC, B, M, K, N, hip.epilogue(dA, K))
    dB = torch.zeros(N, K, device="cuda", dtype=torch.float32)
    hip.gemm_tn(BF16, dC, A, N, K, M, hip.epilogue(dB, K, atomic=True))
    a = (Cm.double() * dC.double()).sum().item()
    b = (dA.double() * A.double()).sum().item()
    c = (dB.double() * B.double()).sum().item()
    scale = (Cm.double().abs() * dC.double().abs()).sum().item()
    assert abs(a - b) < 1e-5 * scale and abs(a - c) < 1e-5 * scale, (a, b, c, scale)


@pytest.mark.parametrize("kind,M,N,K", [("nt", 128, 2048, 2048), ("nn", 128, 768, 2048), ("nt", 256, 200, 1000), ("nn", 256, 1000, 200), ("nt", 128, 768, 768)])
def test_splitk_workspace_form_matches_fused_path(kind, M, N, K):
    """The heads' tiny-M GEMMs through clite_epilogue.splitk_ws (split-K into a zeroed f32 workspace + finishing kernel) against the
    fused single-pass path on the same operands and against torch: bias + ReLU + pre-activation / residual / column statistics."""
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = _t((M, K), g, BF16)
    B = _t((N, K) if kind == "nt" else (K, N), g, BF16, 0.05)
    res, bias = _t((M, N), g, BF16), torch.randn(N, device="cuda", generator=g)
    f = hip.gemm_nt if kind == "nt" else hip.gemm_nn
    outs = []
    for use_ws in (False, True):
        out, pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        stats = hip.Stats(torch.zeros(8 * 3 * N, device="cuda"), 8, N)
        ws = torch.zeros(M, N, device="cuda") if use_ws else None
        f(BF16, A, B, M, N, K, hip.epilogue(out, N, bias=bias, act=hip.ACT_RELU, preact=pre, residual=res, colsum=stats, ws=ws))
        if use_ws:
            assert (ws.abs().max() > 0) == (K >= 256)      # the split-K form ran (K < 8 tiles does not qualify: the fused path is taken)
        outs.append((out.float(), pre.float(), stats.t.view(8, 3, N).sum(0)))
    z = A.float() @ (B.float().t() if kind == "nt" else B.float()) + bias
    ref = z.relu() + res.float()
    for out, pre, cs in outs:
        assert _rel(out, ref) < 6e-3 and _rel(pre, z) < 6e-3
        assert _rel(cs[0], out.sum(0)) < 1e-3 and _rel(cs[1], (out ** 2).sum(0)) < 1e-3
    assert _rel(outs[1][0], outs[0][0]) < 8e-3


def test_grouped_weight_gradients_match_members(tile_policy):
    """clite_wgrad_group on the GPU: the weight gradients of a ResNet-50 stage slice (all three conv tile families, one member with
    > 128 K tiles per workgroup chunk) and two BERT linear gradients as ONE grouped launch set accumulate the same values (+= onto a
    non-zero arena) as torch fp32 on the same bf16 inputs; and the member-by-member path of the same entry point (deterministic mode)
    agrees. Bound 2e-3 of max (fp32 accumulation, summation order only)."""
    if tile_policy != 0:
        pytest.skip("independent of the tile policy")
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(17)
    cases = [(8, 56, 56, 64, 64, 3, 1, 1), (8, 56, 56, 64, 256, 1, 1, 0), (8, 28, 28, 512, 128, 1, 1, 0), (16, 14, 14, 256, 256, 3, 1, 1), (32, 7, 7, 2048, 512, 1, 1, 0)]
    for det in (False, True):
        hip.set_deterministic(det)
        try:
            grp = hip.WgradGroup(BF16)
            checks = []
            for (N, H, W, Cc, K, R, st, pad) in cases:
                cv = hip.conv_desc(BF16, N, H, W, Cc, K, R, R, st, pad)
                x, dy = _t((N, H, W, Cc), g, BF16), _t((N, cv.Ho, cv.Wo, K), g, BF16)
                dw = torch.ones(K, R, R, Cc, device="cuda")
                grp.conv(dy, x, cv, dw)
                w32 = torch.zeros(K, Cc, R, R, device="cuda", requires_grad=True)
                with torch.backends.cudnn.flags(enabled=False):
                    F.conv2d(x.float().permute(0, 3, 1, 2), w32, stride=st, padding=pad).backward(dy.float().permute(0, 3, 1, 2))
                checks.append((dw, 1 + w32.grad.permute(0, 2, 3, 1)))
            for (M, N, K) in [(3072, 768, 3840), (768, 768, 3840)]:
                A, B = _t((K, M), g, BF16), _t((K, N), g, BF16)
                out = torch.ones(M, N, device="cuda")
                grp.linear(A, B, M, N, K, out)
                checks.append((out, 1 + A.float().t() @ B.float()))
            grp.launch()
            torch.cuda.synchronize()
            for got, ref in checks:
                assert _rel(got, ref) < 2e-3
        finally:
            hip.set_deterministic(False)


@pytest.mark.parametrize("N,H,W", [(8, 56, 56), (5, 28, 28), (3, 30, 44), (300, 14, 14)])
def test_patch_resident_conv3x3_64ch(N, H, W, tile_policy):
    """conv_patch.hip (taken under the automatic policy; every forced policy runs the implicit-GEMM kernels on the same launch, so this case also
    checks those against the same references): 3 x 3 / stride 1 / pad 1, 64 -> 64 channels, bf16 — forward with the per-channel statistics of the
    stored values, the input gradient on the transposed weights with a plain store and in the BatchNorm-backward form (packed relu' bits, sum v,
    sum v (bn_y - mean)). Against torch fp32 convolutions of the same bf16 inputs: 6e-3 of max|ref| (one bf16 rounding of the output), statistics
    1e-3 of their own magnitude. Shapes: the step's 56 x 56 (7 strips per workgroup would need 1792 strips: N = 8 gives 112 strips on 112
    workgroups; N = 300 at 14 x 14 gives 300 whole-image strips on 256 workgroups, i.e. two strips for some), 28 x 28 (9-row strips, ragged last
    strip), 30 x 44 (5-row strips)."""
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(N + H + W)
    C = 64
    x, dy = _t((N, H, W, C), g, BF16), _t((N, H, W, C), g, BF16)
    w = _t((C, 3, 3, C), g, BF16, 0.1)
    cv = hip.conv_desc(BF16, N, H, W, C, C, 3, 3, 1, 1)
    M = N * H * W
    y = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    st = hip.Stats(torch.zeros(4 * 3 * C, device="cuda"), 4, C)
    hip.conv_fwd(x, w, cv, hip.epilogue(y, C, colsum=st))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(M, C)
    assert _rel(y, ref) < 6e-3
    s = st.t.view(4, 3, C).sum(0)
    yf = y.float()
    assert _rel(s[0], yf.sum(0)) < 1e-3 and _rel(s[1], (yf * yf).sum(0)) < 1e-3 and not s[2].any()
    wt = w.permute(3, 1, 2, 0).contiguous()
    xin = torch.zeros(N, C, H, W, device="cuda", requires_grad=True)
    gref = torch.autograd.grad(F.conv2d(xin, w.float().permute(0, 3, 1, 2), padding=1), xin, dy.float().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1).reshape(M, C)
    dx = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    hip.conv_dgrad(dy, wt, cv, hip.epilogue(dx, C), wt=True)
    assert _rel(dx, gref) < 6e-3
    aux = torch.randn(M, C, device="cuda", generator=g)
    bits = torch.from_numpy(__import__("numpy").packbits((aux > 0).cpu().numpy(), axis=-1, bitorder="little")).cuda()
    by = _t((M, C), g, BF16) + 3
    fst = hip.Stats(torch.zeros(2 * 3 * C, device="cuda"), 2, C)
    fst.t.view(2, 3, C)[:, 0] = by.float().sum(0) / 2
    dst = hip.Stats(torch.zeros(2 * 3 * C, device="cuda"), 2, C)
    out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    hip.conv_dgrad(dy, wt, cv, hip.epilogue(out, C, relu_bits=bits, colsum=dst, bn=(by, fst, M)), wt=True)
    want = gref * (aux > 0)
    assert _rel(out, want) < 6e-3
    of = out.float()
    d = dst.t.view(2, 3, C).sum(0)
    mean = by.float().sum(0) / M
    assert _rel(d[0], of.sum(0)) < 1e-3 and _rel(d[1], (of * (by.float() - mean)).sum(0)) < 2e-3


@pytest.mark.parametrize("N,H,W", [(8, 56, 56), (5, 28, 28), (300, 14, 14)])
def test_patch_resident_wgrad3x3_64ch(N, H, W, tile_policy):
    """clite_conv_wgrad_patch (conv_patch.hip; through hip.conv_wgrad_patch, which falls back to clite_conv_wgrad when the library declines — under
    every forced tile policy, so that path meets the same reference): dW [64][3][3][64] f32 += dy^T im2col(x) against torch's fp32 weight
    gradient of the same bf16 operands, 2e-3 of max|ref|; accumulation into a non-zero dW."""
    hip = _hip()
    g = torch.Generator(device="cuda").manual_seed(N * H)
    C = 64
    x, dy = _t((N, H, W, C), g, BF16), _t((N, H, W, C), g, BF16)
    cv = hip.conv_desc(BF16, N, H, W, C, C, 3, 3, 1, 1)
    wz = torch.zeros(C, C, 3, 3, device="cuda", requires_grad=True)
    ref = torch.autograd.grad(F.conv2d(x.float().permute(0, 3, 1, 2), wz, padding=1), wz, dy.float().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)      # [K][R][S][C]
    dw = torch.full((C, 3, 3, C), 0.25, device="cuda")
    hip.conv_wgrad_patch(dy, x, cv, dw)
    assert _rel(dw - 0.25, ref) < 2e-3
    # as a member of a grouped launch (what the backward executors do)
    dw2 = torch.zeros(C, 3, 3, C, device="cuda")
    grp = hip.WgradGroup(BF16)
    grp.conv(dy, x, cv, dw2)
    grp.launch()
    assert _rel(dw2, ref) < 2e-3
