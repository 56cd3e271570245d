"""The object bench.py's headline number is measured on: backbone_pipeline.NativeDistillModel
(teacher backbone on a second stream, filter gradients on the executor's auxiliary stream, FPN
gradient Sum, both SGD updates), i.e. build_generic_retinanet_model_dissstillation
(detectron/lib/modeling/model_builder.py:373-411) on one GPU.

Reference = composition of independent pieces: tests/torch_ref.RefResNetFPN in float64 (plain
torch convolutions) for both backbones -> oracle/head_step.py (the CPU restatement of the subnets
and all four losses) -> torch autograd through the float64 student backbone.  ReLU masks are made
flip-proof (torch_ref.calibrate, make_mask_safe), so what is compared is arithmetic and the bar is
north_star's 1e-4.  Then the same steps with every overlap switched off must reproduce the
overlapped run BIT FOR BIT (a stream race shows up as a difference).
"""
import numpy as np
import pytest
import torch

import ssad_amd  # noqa: F401
from ssad_amd import synth
from ssad_amd.modeling.retinanet_heads import HeadConfig
from oracle import head_step

from torch_ref import RefResNetFPN
from test_gpu_operators import make_mask_safe, close_1e4

pytestmark = pytest.mark.gpu

N, HW = 2, (256, 384)
SHAPES = [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
LR, MU, WD = 1e-4, 0.9, 1e-4


def _problem(seed=3):
    dev = "cuda"
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=dev).manual_seed(seed)
    images = torch.randn((N, 3) + HW, device=dev, generator=gen)
    ref_s = RefResNetFPN("r50", seed=11).calibrate(images)
    ref_t = RefResNetFPN("r101", seed=12)
    with torch.no_grad():
        fs = [f.float().cpu().numpy() for f in ref_s(images)]
    cfg = HeadConfig(num_gpus=1)
    S = make_mask_safe(cfg, synth.head_params(rng), fs)
    T = synth.head_params(rng)
    labs = [synth.distill_inputs(rng, N, 9, 80, h, w)[2] for h, w in SHAPES]
    tg = [synth.bbox_targets(rng, l) for l in labs]
    fg = np.array([max(1, sum(t[0].shape[0] for t in tg))], np.float32)
    return cfg, images, ref_s, ref_t, S, T, labs, tg, fg


def _reference(cfg, images, ref_s, ref_t, S, T, labs, tg, fg):
    """Losses, subnet gradients, backbone gradients of one iteration on the current weights."""
    ref_s.zero_grad()
    f_s = ref_s(images)
    with torch.no_grad():
        f_t = ref_t(images)
    fs = [f.detach().float().cpu().numpy() for f in f_s]
    ft = [f.float().cpu().numpy() for f in f_t]
    out = head_step.head_step(S, T, fs, ft, labs, scale=cfg.loss_scale * cfg.temperature ** 2,
                              loss_scale=cfg.loss_scale, bbox_targets=tg, fg_num=fg)
    d_fpn = [out["d_fpn"]["cls"][l].astype(np.float64) + out["d_fpn"]["bbox"][l] for l in range(len(fs))]
    torch.autograd.backward(f_s, [torch.from_numpy(d).to(images.device) for d in d_fpn])
    out["fpn_student"], out["fpn_teacher"], out["d_fpn_sum"] = fs, ft, d_fpn
    out["backbone_grads"] = {k: v.grad.detach().clone() for k, v in ref_s.named_parameters()}
    return out


def _model(cfg, ref_s, ref_t, S, T, overlap):
    from ssad_amd.head_pipeline import DistillHeads
    from ssad_amd.backbone_pipeline import NativeDistillModel
    heads = DistillHeads(cfg, N=N, shapes=SHAPES, device="cuda", student_init=S, teacher_init=T, lr=LR,
                         momentum=MU, weight_decay=WD, overlap_wgrad=overlap)
    return NativeDistillModel(heads, "r50", "r101", N, HW, "cuda", lr=LR, momentum=MU, weight_decay=WD,
                              two_streams=overlap, overlap_wgrad=overlap, student_src=ref_s.state_dict(),
                              teacher_src=ref_t.state_dict(), student_scales=ref_s.scales)


def _inputs(labs, tg, fg):
    to = lambda a: torch.from_numpy(a).cuda()
    return [to(a) for a in labs], [(to(y), to(l)) for y, l in tg], to(fg)


def _check_gradients(model, ref, tag):
    h, st = model.heads, model.student
    for l in range(len(SHAPES)):
        close_1e4(st.fpn[l].cpu().numpy(), ref["fpn_student"][l], "%s student P%d" % (tag, l + 3))
        close_1e4(model.teacher.fpn[l].cpu().numpy(), ref["fpn_teacher"][l], "%s teacher P%d" % (tag, l + 3))
        if l != 3:      # P6's buffer also receives P7's gradient through relu(P6) in place (FPN.py:193-224)
            close_1e4(st.d_fpn[l].cpu().numpy(), ref["d_fpn_sum"][l], "%s d_fpn P%d" % (tag, l + 3))
    np.testing.assert_allclose(h.losses.cpu().numpy(), ref["losses"], rtol=1e-4)
    np.testing.assert_allclose(h.focal_losses.cpu().numpy(), ref["focal_losses"], rtol=1e-4)
    np.testing.assert_allclose(h.bbox_losses.cpu().numpy(), ref["bbox_losses"], rtol=1e-4)
    for name, g in ref["grads"].items():
        close_1e4(h.grads[name].cpu().numpy(), g, "%s %s" % (tag, name))
    seen = 0
    for name, g in ref["backbone_grads"].items():
        lname, kind = name.rsplit(".", 1)
        layer = st._layers[lname]
        mine = layer.gw if kind == "weight" else layer.gb
        assert mine is not None, name
        close_1e4(mine.cpu().numpy(), g.cpu().numpy(), "%s %s" % (tag, name))
        seen += 1
    # every trainable tensor of the native backbone was compared (no bias of the body is trained)
    assert seen == sum(1 + (l.gb is not None) for l in st._layers.values() if l.train)
    assert all(l.gb is None for l in st._layers.values() if l.train and l.affine)


def _expected_update(model, ref, p_heads, p_body, m_heads, m_body):
    """MomentumSGDUpdate on the reference gradients (optimizer.py:115-130, momentum_sgd_op_gpu.cu:22-38):
    weights g + wd w (rows of a folded filter times s^2), biases 2 g."""
    h, st = model.heads, model.student
    want_h = torch.empty_like(p_heads)
    for name, shape, is_bias, _ in h.params.specs:
        off, n = h.params.offsets[name], int(np.prod(shape))
        g = torch.from_numpy(np.asarray(ref["grads"][name], np.float64)).reshape(-1).cuda()
        w = p_heads[off:off + n].double()
        gg = 2.0 * g if is_bias else g + WD * w
        want_h[off:off + n] = (LR * gg + MU * m_heads[off:off + n].double()).float()
    want_b = torch.empty_like(p_body)
    for lname, layer in st._layers.items():
        if not layer.train:
            continue
        off = layer.w.data_ptr() - st.params_flat.data_ptr()
        off //= 4
        n = layer.w.numel()
        g = ref["backbone_grads"][lname + ".weight"].reshape(layer.cout, -1)
        if layer.s2 is not None:
            g = g * layer.s2.double().view(-1, 1)
        w = p_body[off:off + n].double()
        want_b[off:off + n] = (LR * (g.reshape(-1) + WD * w) + MU * m_body[off:off + n].double()).float()
        if layer.gb is not None:
            ob = (layer.b.data_ptr() - st.params_flat.data_ptr()) // 4
            gb = ref["backbone_grads"][lname + ".bias"]
            want_b[ob:ob + layer.cout] = (LR * 2.0 * gb + MU * m_body[ob:ob + layer.cout].double()).float()
    return want_h, want_b


def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-300))


def test_native_distill_model_step_matches_composed_reference_and_is_race_free():
    cfg, images, ref_s, ref_t, S, T, labs, tg, fg = _problem()
    labels, targets, fg_num = _inputs(labs, tg, fg)
    model = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    assert model.side is not None and model.student._wstreams == [1, 2] and model.heads._wstream == 1
    assert any(s2 is not None for (_, _, _, _, s2) in model.student.segments)      # the folded c3 scales
    h, st = model.heads, model.student

    # iteration 1, gradients only
    ref1 = _reference(cfg, images, ref_s, ref_t, S, T, labs, tg, fg)
    model.step(images, labels, targets, fg_num, update=False)
    torch.cuda.synchronize()
    _check_gradients(model, ref1, "step 1")

    # iteration 1 again, as bench.py runs it (with both updates)
    p_h0, p_b0 = h.params.flat.clone(), st.params_flat.clone()
    z_h, z_b = torch.zeros_like(p_h0), torch.zeros_like(p_b0)
    model.step(images, labels, targets, fg_num)
    torch.cuda.synchronize()
    want_h, want_b = _expected_update(model, ref1, p_h0, p_b0, z_h, z_b)
    m_h1, m_b1 = h.moms.flat.clone(), st.moms_flat.clone()
    assert _rel(m_h1, want_h) < 1e-4 and _rel(m_b1, want_b) < 1e-4, (_rel(m_h1, want_h), _rel(m_b1, want_b))
    assert torch.equal(h.params.flat, p_h0 - m_h1) and torch.equal(st.params_flat, p_b0 - m_b1)
    # the frozen values did not move
    assert torch.equal(st._layers["res3.0.c1"].b.cpu(), ref_s.p["res3.0.c1.bias"].float().cpu())
    assert torch.equal(st._layers["res2.0.c1"].w.cpu(), ref_s.p["res2.0.c1.weight"].float().cpu())

    # iteration 2 on the updated weights (repacked filters, momentum in play)
    with torch.no_grad():
        for lname, layer in st._layers.items():
            if layer.train:
                ref_s.p[lname + ".weight"].copy_(layer.w.double())
                if layer.gb is not None:
                    ref_s.p[lname + ".bias"].copy_(layer.b.double())
    S2 = {name: h.params[name].cpu().numpy().copy() for name in S}
    ref2 = _reference(cfg, images, ref_s, ref_t, S2, T, labs, tg, fg)
    p_h1, p_b1 = h.params.flat.clone(), st.params_flat.clone()
    model.step(images, labels, targets, fg_num)
    torch.cuda.synchronize()
    np.testing.assert_allclose(h.losses.cpu().numpy(), ref2["losses"], rtol=1e-4)
    want_h, want_b = _expected_update(model, ref2, p_h1, p_b1, m_h1, m_b1)
    assert _rel(h.moms.flat, want_h) < 1e-4 and _rel(st.moms_flat, want_b) < 1e-4, \
        (_rel(h.moms.flat, want_h), _rel(st.moms_flat, want_b))
    # the part of the second update that is new (m2 - mu m1 = lr g'): the gradient itself
    new_b, want_new_b = st.moms_flat - MU * m_b1, want_b - MU * m_b1
    assert _rel(new_b, want_new_b) < 2e-4, _rel(new_b, want_new_b)
    assert torch.equal(st.params_flat, p_b1 - st.moms_flat)


def test_overlapped_and_serial_execution_agree_bit_for_bit():
    """Teacher on a second stream + filter gradients on the auxiliary stream against the same
    program on one stream: identical bits in every loss, gradient, parameter and momentum after
    three iterations (kernels are deterministic, so any difference is a missing dependency)."""
    cfg, images, ref_s, ref_t, S, T, labs, tg, fg = _problem(seed=4)
    labels, targets, fg_num = _inputs(labs, tg, fg)
    a = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    b = _model(cfg, ref_s, ref_t, S, T, overlap=False)
    assert a.side is not None and b.side is None and b.student._wstreams == [0] and b.heads._wstream == 0
    for it in range(3):
        for m in (a, b):
            if it == 0:
                # poison every scratch buffer: anything read before it is written shows up
                m.student.poison()
                m.teacher.poison()
            m.step(images, labels, targets, fg_num)
        torch.cuda.synchronize()
        for name in ("losses", "focal_losses", "bbox_losses"):
            assert torch.equal(getattr(a.heads, name), getattr(b.heads, name)), (it, name)
        assert torch.isfinite(a.heads.losses).all()
        for x, y, what in ((a.heads.params.flat, b.heads.params.flat, "subnet parameters"),
                           (a.heads.moms.flat, b.heads.moms.flat, "subnet momentum"),
                           (a.heads.grads.flat, b.heads.grads.flat, "subnet update"),
                           (a.student.params_flat, b.student.params_flat, "backbone parameters"),
                           (a.student.moms_flat, b.student.moms_flat, "backbone momentum"),
                           (a.student.grads_flat, b.student.grads_flat, "backbone update")):
            assert torch.isfinite(x).all(), (it, what)
            assert torch.equal(x, y), (it, what, float((x - y).abs().max()))
        for l in range(len(SHAPES)):
            assert torch.equal(a.student.d_fpn[l], b.student.d_fpn[l]), (it, "d_fpn", l)
            assert torch.equal(a.teacher.fpn[l], b.teacher.fpn[l]), (it, "teacher fpn", l)


def test_update_lr_reaches_the_backbone():
    """One schedule for the whole detector (detector.py:594-648): UpdateWorkspaceLr + momentum
    correction on the backbone's flat buffers as on the subnets'."""
    cfg, images, ref_s, ref_t, S, T, labs, tg, fg = _problem(seed=5)
    m = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    assert float(m.student.lr) == pytest.approx(float(m.heads.lr)) == pytest.approx(LR)
    m.student.moms_flat.fill_(1.0)
    m.heads.moms.flat.fill_(1.0)
    m.update_lr(LR * 0.1)
    assert float(m.student.lr) == pytest.approx(LR * 0.1) and float(m.heads.lr) == pytest.approx(LR * 0.1)
    assert torch.allclose(m.student.moms_flat, torch.full_like(m.student.moms_flat, 0.1))
    assert torch.allclose(m.heads.moms.flat, torch.full_like(m.heads.moms.flat, 0.1))
    m.update_lr(LR * 0.1 * 1.05)          # below the threshold: no momentum correction
    assert torch.allclose(m.student.moms_flat, torch.full_like(m.student.moms_flat, 0.1))


def test_teacher_running_ahead_of_the_previous_step_changes_no_bit():
    """The frozen teacher's forward pass is ordered after the previous step's last reader of its output buffers,
    not after the previous step's update (NativeDistillModel.step): with the launching thread ahead of the GPU --
    six iterations enqueued back to back, no synchronisation in between -- it overlaps the previous step's
    backward pass.  Same bits as the model whose teacher waits for the whole previous step; also with a fresh
    image tensor every step (falls back to the full wait) and with the input pipeline's event."""
    cfg, images, ref_s, ref_t, S, T, labs, tg, fg = _problem(seed=6)
    labels, targets, fg_num = _inputs(labs, tg, fg)
    a = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    b = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    c = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    d = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    a._teacher_ahead, b._teacher_ahead, c._teacher_ahead, d._teacher_ahead = True, False, True, True
    copy_stream = torch.cuda.Stream()
    for m in (a, b, c, d):
        m.student.poison()
        m.teacher.poison()
        torch.cuda.synchronize()
        for it in range(6):
            if m is c:
                m.step(images.clone(), labels, targets, fg_num)          # a tensor the model has not seen
            elif m is d:
                with torch.cuda.stream(copy_stream):                      # the input pipeline: copy + event
                    fresh = images.clone()
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                torch.cuda.current_stream().wait_event(ev)               # the student reads it on this stream
                m.step(fresh, labels, targets, fg_num, images_event=ev)
            else:
                m.step(images, labels, targets, fg_num)
        torch.cuda.synchronize()
    assert a._t_fpn_read is not None and a._images_ref is images
    for m, what in ((a, "ahead"), (c, "fresh tensor"), (d, "images_event")):
        assert torch.isfinite(m.heads.losses).all()
        assert torch.equal(m.heads.losses, b.heads.losses), what
        assert torch.equal(m.heads.params.flat, b.heads.params.flat), what
        assert torch.equal(m.student.params_flat, b.student.params_flat), what
        assert torch.equal(m.student.moms_flat, b.student.moms_flat), what
        for l in range(len(SHAPES)):
            assert torch.equal(m.teacher.fpn[l], b.teacher.fpn[l]), (what, l)


def test_fresh_batch_tensor_every_step_is_never_read_before_its_producer():
    """A training loop hands step() a NEW tensor every iteration, produced by work queued on the
    current stream (host-to-device copy, preprocessing).  The caching allocator gives the new batch
    the address of the one just freed and its version counter starts at 0 again: the teacher must not
    take that for "the tensor the previous step already read" and run ahead of the producer on its
    side stream (it would compute this step's distillation targets on the previous batch).  Six
    iterations enqueued back to back with a different batch each, against the model whose teacher
    waits for everything: same bits."""
    cfg, images, ref_s, ref_t, S, T, labs, tg, fg = _problem(seed=7)
    labels, targets, fg_num = _inputs(labs, tg, fg)
    a = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    b = _model(cfg, ref_s, ref_t, S, T, overlap=True)
    a._teacher_ahead, b._teacher_ahead = True, False
    seen = {}
    for m in (a, b):
        m.student.poison()
        m.teacher.poison()
        torch.cuda.synchronize()
        ptrs = []
        for it in range(6):
            batch = images * (1.0 + 0.25 * it)            # produced on the current stream, right now
            ptrs.append(batch.data_ptr())
            m.step(batch, labels, targets, fg_num)
            del batch
        torch.cuda.synchronize()
        seen[m] = ptrs
    # the situation the test is about did occur: a later batch reused an earlier batch's address
    assert len(set(seen[a])) < len(seen[a])
    assert torch.isfinite(a.heads.losses).all()
    assert torch.equal(a.heads.losses, b.heads.losses)
    assert torch.equal(a.heads.params.flat, b.heads.params.flat)
    assert torch.equal(a.student.params_flat, b.student.params_flat)
    for l in range(len(SHAPES)):
        assert torch.equal(a.teacher.fpn[l], b.teacher.fpn[l]), l
