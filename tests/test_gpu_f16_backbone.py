"""The backbones' fp16-storage kernels (BASELINE config 5's precision: CudnnConvOp<float16> with
fp32 math, conv_op_cudnn.cc:631-636): pointwise GEMM (gemm_f16.hip), pointwise filter gradient
(conv3x3_wgrad_f16_kernel<true>), ResNeXt's grouped 3x3 (grouped_f16.hip), the elementwise passes
and the stem pool -- each against a float64 evaluation of the SAME fp16-rounded operands, so that
what is left is fp32 accumulation order and the one output rounding (<= 1 fp16 ulp of the output
scale); then the networks built from them against tests/torch_ref.py."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ssad_amd  # noqa: F401
from ssad_amd import kernels as K

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def _blk(x, scale=1.0):
    return K.f16_pack_activations(x.contiguous(), scale)


def _unblk(xb, Cc):
    return K.f16_unpack_activations(xb, Cc)


def _h(x):
    """float32 tensor rounded to fp16 values, as float64."""
    return x.half().double()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pw(x, w, M, Ho, Wo, bias=None, res=None, mask=None, relu=False, res_up=False, stride=1, dgrad=False):
    """x blocked; w float32 [M][C] (or for dgrad the forward filter [C_out_of_fwd = C][..]).  Returns blocked y."""
    L = K.lib()
    Cc = w.shape[1]
    n = L.ssad_pw_f16_filter_halves(M, Cc)
    wf = torch.empty(n, dtype=torch.float16, device=DEV)
    rc = L.ssad_pw_f16_pack_filter(w.data_ptr(), M, Cc, wf.data_ptr(), None, _stream())
    assert rc == 0
    y = torch.empty((x.shape[0], M // 8, Ho, Wo, 8), dtype=torch.float16, device=DEV)
    d = K.PwF16()
    d.x, d.w, d.y = x.data_ptr(), wf.data_ptr(), y.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual = res.data_ptr() if res is not None else None
    d.mask = mask.data_ptr() if mask is not None else None
    d.N, d.C, d.M, d.Ho, d.Wo, d.Hi, d.Wi, d.stride = x.shape[0], Cc, M, Ho, Wo, x.shape[2], x.shape[3], stride
    d.flags = (K.CONV_RELU if relu else 0) | (K.PW_F16_RES_UPSAMPLE2 if res_up else 0)
    rc = L.ssad_conv1x1_f16(C.byref(d), _stream())
    assert rc == 0, rc
    return y


def _close(got, want, what, tol=1.5e-3):
    """Elementwise: |got - want| <= tol * max|want| (an fp16 result: half an ulp is 4.9e-4 relative
    to the element, i.e. less than that of the tensor's scale)."""
    got, want = got.double(), want.double()
    err = float((got - want).abs().max())
    top = float(want.abs().max())
    assert torch.isfinite(got).all(), what
    assert err <= tol * top + 1e-30, "%s: max abs err %.3e of max %.3e" % (what, err, top)


@pytest.mark.parametrize("N,Cc,M,H,W", [(2, 256, 64, 24, 40),       # c1 of res2: narrow output (half a 128-row tile)
                                        (1, 64, 256, 17, 23),       # c3 of res2: odd map, pixel tail
                                        (3, 72, 136, 9, 11),        # channel tails: 9 blocks in, 17 out
                                        (16, 1024, 256, 8, 12),     # res4 c1 at config 5's own map (512 x 768 / 16)
                                        (2, 512, 2048, 16, 24)])    # res5 c3: 16 output blocks, 256-pixel tiles
def test_pointwise_f16_forward_bias_residual_relu(N, Cc, M, H, W):
    g = _gen(1)
    x = torch.randn((N, Cc, H, W), device=DEV, generator=g)
    w = torch.randn((M, Cc), device=DEV, generator=g) / np.sqrt(Cc)
    b = torch.randn((M,), device=DEV, generator=g) * 0.3
    r = torch.randn((N, M, H, W), device=DEV, generator=g)
    want = torch.einsum("mc,nchw->nmhw", _h(w), _h(x)) + b.double().view(1, -1, 1, 1)
    y = _unblk(_pw(_blk(x), w, M, H, W, bias=b), M)
    _close(y, want, "bias")
    y = _unblk(_pw(_blk(x), w, M, H, W, bias=b, res=_blk(r), relu=True), M)
    _close(y, F.relu(want + _h(r)), "bias + residual + relu")
    # run-to-run bit equality
    y2 = _unblk(_pw(_blk(x), w, M, H, W, bias=b, res=_blk(r), relu=True), M)
    assert torch.equal(y, y2)


def test_pointwise_f16_stride2_and_upsampled_residual():
    g = _gen(2)
    N, Cc, M, H, W = 2, 256, 512, 12, 20
    x = torch.randn((N, Cc, 2 * H, 2 * W), device=DEV, generator=g)
    w = torch.randn((M, Cc), device=DEV, generator=g) / np.sqrt(Cc)
    want = torch.einsum("mc,nchw->nmhw", _h(w), _h(x[:, :, ::2, ::2]))
    _close(_unblk(_pw(_blk(x), w, M, H, W, stride=2), M), want, "stride 2 in the loader")
    # FPN top-down: lateral + 2x nearest upsampling of the coarser level (FPN.py:283-306)
    xs = torch.randn((N, Cc, 2 * H, 2 * W), device=DEV, generator=g)
    coarse = torch.randn((N, M, H, W), device=DEV, generator=g)
    want = torch.einsum("mc,nchw->nmhw", _h(w), _h(xs)) + F.interpolate(_h(coarse), scale_factor=2, mode="nearest")
    got = _unblk(_pw(_blk(xs), w, M, 2 * H, 2 * W, res=_blk(coarse), res_up=True), M)
    _close(got, want, "lateral + upsampled residual")


def test_pointwise_f16_data_gradient_with_mask_and_sum():
    """dX = W^T dY (conv_op_impl.h:524-560) through the transposed pack, ReluGradient mask of the
    layer below and the identity shortcut's gradient in the epilogue."""
    g = _gen(3)
    N, Cc, M, H, W = 2, 256, 1024, 10, 14            # forward layer C -> M
    w = torch.randn((M, Cc), device=DEV, generator=g) / np.sqrt(M)
    dy = torch.randn((N, M, H, W), device=DEV, generator=g)
    act = torch.randn((N, Cc, H, W), device=DEV, generator=g)           # the layer's input (post-ReLU elsewhere)
    other = torch.randn((N, Cc, H, W), device=DEV, generator=g)
    L = K.lib()
    n = L.ssad_pw_f16_filter_halves(M, Cc)
    wd = torch.empty(n, dtype=torch.float16, device=DEV)
    assert L.ssad_pw_f16_pack_filter(w.data_ptr(), M, Cc, None, wd.data_ptr(), _stream()) == 0
    dyb, actb, otherb = _blk(dy), _blk(act), _blk(other)
    dx = torch.empty_like(actb)
    d = K.PwF16()
    d.x, d.w, d.y, d.residual, d.mask = dyb.data_ptr(), wd.data_ptr(), dx.data_ptr(), otherb.data_ptr(), actb.data_ptr()
    d.N, d.C, d.M, d.Ho, d.Wo, d.Hi, d.Wi, d.stride, d.flags = N, M, Cc, H, W, H, W, 1, 0
    assert L.ssad_conv1x1_f16(C.byref(d), _stream()) == 0
    want = torch.einsum("mc,nmhw->nchw", _h(w), _h(dy)) + _h(other)
    want = torch.where(_h(act) > 0, want, torch.zeros_like(want))
    _close(_unblk(dx, Cc), want, "masked data gradient + sum")


@pytest.mark.parametrize("N,Cc,M,H,W", [(2, 256, 64, 24, 40), (1, 72, 136, 9, 11), (16, 1024, 256, 8, 12)])
def test_pointwise_f16_filter_gradient(N, Cc, M, H, W):
    g = _gen(4)
    x = torch.randn((N, Cc, H, W), device=DEV, generator=g)
    dy = torch.randn((N, M, H, W), device=DEV, generator=g)
    L = K.lib()
    nb = L.ssad_conv1x1_wgrad_f16_workspace_bytes(N, Cc, H, W, M)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    dw = torch.empty((M, Cc), device=DEV)
    db = torch.empty((M,), device=DEV)
    inv = torch.tensor([0.5], device=DEV)
    xb, dyb = _blk(x), _blk(dy)              # (kept alive: the launcher only sees addresses)
    rc = L.ssad_conv1x1_wgrad_f16(xb.data_ptr(), dyb.data_ptr(), N, Cc, H, W, M, 0, 2.0, inv.data_ptr(),
                                  dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, _stream())
    assert rc == 0, rc
    want = torch.einsum("nmhw,nchw->mc", _h(dy), _h(x))          # scale 2.0 * 0.5 = 1
    _close(dw, want, "dW", tol=2e-5)                             # fp32 results: accumulation order only
    _close(db, _h(dy).sum(dim=(0, 2, 3)), "db", tol=2e-5)
    dw2 = torch.empty_like(dw)
    L.ssad_conv1x1_wgrad_f16(xb.data_ptr(), dyb.data_ptr(), N, Cc, H, W, M, 0, 2.0, inv.data_ptr(),
                             dw2.data_ptr(), None, ws.data_ptr(), nb, _stream())
    assert torch.equal(dw, dw2)                                  # deterministic


@pytest.mark.parametrize("Cc,cg,H,W", [(256, 4, 19, 33), (512, 8, 16, 24), (1024, 16, 9, 12), (2048, 32, 8, 12)])
def test_grouped_conv3x3_f16_matches_float64(Cc, cg, H, W):
    """ResNeXt's cardinality-64 3x3 (ResNet.py:247-258) at the four group widths of X-101-64x4d."""
    g = _gen(5)
    N, group = 2, Cc // cg
    x = torch.randn((N, Cc, H, W), device=DEV, generator=g)
    w = torch.randn((Cc, cg, 3, 3), device=DEV, generator=g) / np.sqrt(9 * cg)
    b = torch.randn((Cc,), device=DEV, generator=g) * 0.2
    L = K.lib()
    n = L.ssad_grouped_conv3x3_f16_filter_halves(Cc, group)
    assert n > 0
    pf = torch.empty(n, dtype=torch.float16, device=DEV)
    assert L.ssad_grouped_conv3x3_f16_pack_filter(w.data_ptr(), Cc, group, pf.data_ptr(), _stream()) == 0
    xb = _blk(x)
    y = torch.empty_like(xb)
    assert L.ssad_grouped_conv3x3_f16(xb.data_ptr(), pf.data_ptr(), b.data_ptr(), N, Cc, H, W, group, 1,
                                      y.data_ptr(), _stream()) == 0
    want = F.relu(F.conv2d(_h(x), _h(w), b.double(), 1, 1, 1, group))
    _close(_unblk(y, Cc), want, "grouped 3x3 cg=%d" % cg)


def test_f16_elementwise_passes_and_stem_pool():
    g = _gen(6)
    L = K.lib()
    N, Cc, H, W = 2, 72, 6, 10
    a = torch.randn((N, Cc, 2 * H, 2 * W), device=DEV, generator=g)
    small = torch.randn((N, Cc, H, W), device=DEV, generator=g)
    ab, sb = _blk(a), _blk(small)

    def run(mode, A, B, shape, stride=1, acc=0, init=None):
        y = init.clone() if init is not None else torch.empty((N, (Cc + 7) // 8) + shape + (8,), dtype=torch.float16,
                                                              device=DEV)
        rc = L.ssad_f16_elementwise(mode, A.data_ptr(), B.data_ptr() if B is not None else None, y.data_ptr(), N, Cc,
                                    shape[0], shape[1], stride, acc, _stream())
        assert rc == 0
        return _unblk(y, Cc).double()

    assert torch.equal(run(K.EW_SUBSAMPLE, ab, None, (H, W), 2), _h(a[:, :, ::2, ::2]))
    z = torch.zeros_like(_h(a))
    z[:, :, ::2, ::2] = _h(small)
    assert torch.equal(run(K.EW_SUBSAMPLE_GRAD, sb, None, (2 * H, 2 * W), 2), z)
    got = run(K.EW_SUBSAMPLE_GRAD, sb, None, (2 * H, 2 * W), 2, acc=1, init=ab)
    _close(got, z + _h(a), "scatter + accumulate", tol=1e-3)
    up = _h(a).view(N, Cc, H, 2, W, 2).sum(dim=(3, 5))
    _close(run(K.EW_UPSAMPLE_GRAD, ab, None, (H, W)), up, "upsample gradient", tol=1e-3)
    _close(run(K.EW_UPSAMPLE_GRAD, ab, sb, (H, W)), up + _h(small), "upsample gradient + own", tol=1e-3)
    _close(run(K.EW_SUM2, sb, sb, (H, W)), 2 * _h(small), "sum", tol=1e-3)
    assert torch.equal(run(K.EW_RELU, sb, None, (H, W)), F.relu(_h(small)))
    b2 = _blk(torch.randn((N, Cc, H, W), device=DEV, generator=g))
    assert torch.equal(run(K.EW_RELU_GRAD, sb, b2, (H, W)),
                       torch.where(_h(small) > 0, _unblk(b2, Cc).double(), torch.zeros_like(_h(small))))
    # stem tail: bias + ReLU + 3x3/2 max pool, fp32 NCHW in, blocked fp16 out
    zz = torch.randn((N, 64, 14, 22), device=DEV, generator=g)
    bias = torch.randn((64,), device=DEV, generator=g)
    y = torch.empty((N, 8, 7, 11, 8), dtype=torch.float16, device=DEV)
    assert L.ssad_stem_pool_f16(zz.data_ptr(), bias.data_ptr(), N, 64, 14, 22, y.data_ptr(), _stream()) == 0
    want = F.max_pool2d(F.relu(zz + bias.view(1, -1, 1, 1)), 3, 2, 1)
    assert torch.equal(_unblk(y, 64), want.half().float())


# ---------------------------------------------------------------------------
# the networks built from these kernels
# ---------------------------------------------------------------------------

def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-300))


def test_f16_backbone_forward_backward_vs_float64_reference():
    """NativeResNetFPNF16 (R-50-FPN, train) against tests/torch_ref.py in float64 on the same
    weights, ReLU masks made flip-proof: FPN levels to ~1e-2 of their scale (fp16 activations
    through 50 layers), every trained gradient norm-wise (fp16 activations AND gradients)."""
    from ssad_amd.backbone_f16 import NativeResNetFPNF16
    from torch_ref import RefResNetFPN
    N, hw = 2, (256, 384)
    g = _gen(7)
    images = torch.randn((N, 3) + hw, device=DEV, generator=g)
    ref = RefResNetFPN("r50", seed=11).calibrate(images, margin=0.05)
    nat = NativeResNetFPNF16("r50", N, hw, DEV, train=True, src=ref.state_dict(), lr=0.01, affine_scales=ref.scales)
    nat.pack()
    nat.forward(images)
    got = nat.fpn_f32()
    want = ref(images)
    errs_f = [_rel(a, b.detach()) for a, b in zip(got, want)]
    assert max(errs_f) < 1e-2, errs_f
    d_fpn = [torch.randn(t.shape, device=DEV, generator=g) for t in want]
    torch.autograd.backward(want, [d.double() for d in d_fpn])
    S = 64.0                                            # a loss scale: the program divides it out again
    nat.inv_scale.fill_(1.0 / S)
    nat.backward(d_fpn, scale=S)
    torch.cuda.synchronize()
    errs = {}
    for name, p in ref.p.items():
        if not p.requires_grad:
            continue
        lname, kind = name.rsplit(".", 1)
        layer = nat._layers[lname]
        errs[name] = _rel(layer.gw if kind == "weight" else layer.gb, p.grad)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("fp16 backbone: FPN level errors", errs_f, "worst gradients", worst)
    assert worst[0][1] < 3e-2, worst
    # the update runs on the fp32 master copies
    p0 = nat.params_flat.clone()
    nat.sgd_step()
    assert torch.isfinite(nat.params_flat).all() and not torch.equal(p0, nat.params_flat)


def test_f16_resnext_teacher_forward_vs_float64_reference():
    from ssad_amd.backbone_f16 import NativeResNetFPNF16
    from torch_ref import RefResNetFPN
    N, hw = 1, (128, 256)
    images = torch.randn((N, 3) + hw, device=DEV, generator=_gen(9))
    ref = RefResNetFPN("x101-64x4d", seed=13)
    nat = NativeResNetFPNF16("x101-64x4d", N, hw, DEV, train=False, src=ref.state_dict())
    assert nat._layers["res2.0.c2"].group == 64 and nat._layers["res3.0.c2"].stride == 2
    nat.forward(images)
    with torch.no_grad():
        want = ref(images)
    errs = [_rel(a, b) for a, b in zip(nat.fpn_f32(), want)]
    print("fp16 X-101 teacher: FPN level errors", errs)
    assert max(errs) < 1.5e-2, errs


def test_config5_step_on_native_fp16_kernels_matches_fp32_native_step():
    """BASELINE config 5's step -- R-101-FPN student + ResNeXt-101-64x4d teacher, every
    convolution with fp16 storage / fp32 accumulation on this repo's kernels -- against the fp32
    native step on the same weights and inputs (small image, bs 2): losses, every parameter
    gradient (norm-wise per tensor), and the dynamic loss scale dropping BOTH updates when a
    backbone gradient overflows."""
    from ssad_amd import synth
    from ssad_amd.head_pipeline import DistillHeads, DistillHeadsF16
    from ssad_amd.backbone_pipeline import NativeDistillModel
    from ssad_amd.modeling.retinanet_heads import HeadConfig
    from torch_ref import RefResNetFPN
    N, hw = 2, (256, 384)
    shapes = [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
    rng = np.random.default_rng(21)
    images = torch.randn((N, 3) + hw, device=DEV, generator=_gen(21))
    ref_s = RefResNetFPN("r101", seed=31).calibrate(images, margin=0.05)
    ref_t = RefResNetFPN("x101-64x4d", seed=32)
    cfg = HeadConfig(num_gpus=1)
    # flip-proof ReLU masks in the towers too (fp16 and fp32 activations differ by ~1e-3: a
    # pre-activation that close to zero would take different sides in the two runs)
    from test_gpu_operators import make_mask_safe
    with torch.no_grad():
        fs = [f.float().cpu().numpy() for f in ref_s(images)]
    S, T = make_mask_safe(cfg, synth.head_params(rng), fs), synth.head_params(rng)
    labs = [synth.distill_inputs(rng, N, 9, 80, h, w)[2] for h, w in shapes]
    tg = [synth.bbox_targets(rng, l) for l in labs]
    fg = np.array([max(1, sum(t[0].shape[0] for t in tg))], np.float32)
    to = lambda a: torch.from_numpy(a).to(DEV)
    labels, targets, fg_num = [to(a) for a in labs], [(to(y), to(l)) for y, l in tg], to(fg)
    kw = dict(N=N, shapes=shapes, device=DEV, student_init=S, teacher_init=T, lr=1e-4)
    mk = dict(student_src=ref_s.state_dict(), teacher_src=ref_t.state_dict(), student_scales=ref_s.scales)

    h32 = DistillHeads(cfg, **kw)
    m32 = NativeDistillModel(h32, "r101", "x101-64x4d", N, hw, DEV, **mk)
    m32.step(images, labels, targets, fg_num, update=False)
    h16 = DistillHeadsF16(cfg, blocked_io=True, **kw)
    m16 = NativeDistillModel(h16, "r101", "x101-64x4d", N, hw, DEV, **mk)
    assert m16.backbone_f16 and type(m16.student).__name__ == "NativeResNetFPNF16"
    m16.step(images, labels, targets, fg_num, update=False)
    torch.cuda.synchronize()
    for name in ("losses", "focal_losses", "bbox_losses"):
        a, b = getattr(h16, name).double().cpu(), getattr(h32, name).double().cpu()
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()), (name, a, b)
    errs = {}
    for name in S:
        errs[name] = _rel(h16.grads[name], h32.grads[name])
    for lname, l16 in m16.student._layers.items():
        if l16.train:
            errs[lname] = _rel(l16.gw, m32.student._layers[lname].gw)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("config 5 step, fp16 vs fp32 native: worst gradient differences", worst)
    assert worst[0][1] < 5e-3, worst          # measured 1.6e-3 (fp16 activations and gradients)
    # one real step: both updates applied, everything finite
    p_h, p_b = h16.params.flat.clone(), m16.student.params_flat.clone()
    m16.step(images, labels, targets, fg_num)
    torch.cuda.synchronize()
    assert torch.isfinite(h16.params.flat).all() and torch.isfinite(m16.student.params_flat).all()
    assert not torch.equal(p_h, h16.params.flat) and not torch.equal(p_b, m16.student.params_flat)
    # an overflowing step: scale so large that fp16 gradients become Inf -> BOTH updates are
    # dropped, the scale is halved, momentum untouched
    p_h, p_b = h16.params.flat.clone(), m16.student.params_flat.clone()
    mo_h, mo_b = h16.moms.flat.clone(), m16.student.moms_flat.clone()
    h16.ls_state.copy_(torch.tensor([1.0e9, 1.0e-9], device=DEV))
    m16.step(images, labels, targets, fg_num)
    torch.cuda.synchronize()
    assert torch.equal(p_h, h16.params.flat) and torch.equal(p_b, m16.student.params_flat)
    assert torch.equal(mo_h, h16.moms.flat) and torch.equal(mo_b, m16.student.moms_flat)
    assert float(h16.ls_state[0]) == pytest.approx(min(0.5e9, DistillHeadsF16.LOSS_SCALE_MAX))
    assert int(h16.ls_counters[0]) == 0


def test_config5_fp16_step_against_the_oracle_composition():
    """The fp16 step against the SAME reference the fp32 headline object is held to
    (tests/test_gpu_native_model.py): float64 torch backbones -> oracle/head_step.py (the CPU
    restatement of the subnets and all four losses) -> autograd through the float64 student.
    SURVEY section 7's bar for the fp16 route: activations ~1e-2, losses within 1e-3."""
    from ssad_amd import synth
    from ssad_amd.head_pipeline import DistillHeadsF16
    from ssad_amd.backbone_pipeline import NativeDistillModel
    from ssad_amd.modeling.retinanet_heads import HeadConfig
    from torch_ref import RefResNetFPN
    from test_gpu_operators import make_mask_safe
    import test_gpu_native_model as NM
    N, hw = NM.N, NM.HW
    shapes = NM.SHAPES
    rng = np.random.default_rng(41)
    images = torch.randn((N, 3) + hw, device=DEV, generator=_gen(41))
    ref_s = RefResNetFPN("r101", seed=51).calibrate(images, margin=0.05)
    ref_t = RefResNetFPN("x101-64x4d", seed=52)
    cfg = HeadConfig(num_gpus=1)
    with torch.no_grad():
        fs = [f.float().cpu().numpy() for f in ref_s(images)]
    S, T = make_mask_safe(cfg, synth.head_params(rng), fs), synth.head_params(rng)
    labs = [synth.distill_inputs(rng, N, 9, 80, h, w)[2] for h, w in shapes]
    tg = [synth.bbox_targets(rng, l) for l in labs]
    fg = np.array([max(1, sum(t[0].shape[0] for t in tg))], np.float32)
    ref = NM._reference(cfg, images, ref_s, ref_t, S, T, labs, tg, fg)
    labels, targets, fg_num = NM._inputs(labs, tg, fg)
    h16 = DistillHeadsF16(cfg, blocked_io=True, N=N, shapes=shapes, device=DEV, student_init=S, teacher_init=T, lr=1e-4)
    m16 = NativeDistillModel(h16, "r101", "x101-64x4d", N, hw, DEV, student_src=ref_s.state_dict(),
                             teacher_src=ref_t.state_dict(), student_scales=ref_s.scales)
    m16.step(images, labels, targets, fg_num, update=False)
    torch.cuda.synchronize()
    rel = {}
    for name, want in (("losses", ref["losses"]), ("focal_losses", ref["focal_losses"]), ("bbox_losses", ref["bbox_losses"])):
        got = getattr(h16, name).double().cpu().numpy()
        assert np.isfinite(got).all()
        rel[name] = float(np.abs(got - want).max() / np.abs(want).max())
    gerr = {k: _rel(h16.grads[k], torch.from_numpy(np.asarray(g, np.float64)).to(DEV)) for k, g in ref["grads"].items()}
    for lname, l16 in m16.student._layers.items():
        if l16.train:
            gerr[lname] = _rel(l16.gw, ref["backbone_grads"][lname + ".weight"])
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:5]
    print("config 5 fp16 step vs oracle composition: loss errors", rel, "worst gradients", worst)
    assert max(rel.values()) <= 1e-3, rel          # measured 8e-6
    assert worst[0][1] < 5e-3, worst               # measured 1.9e-3
