"""The drop-in route on the GPU: the reference's graph created ONCE with workspace.CreateNet and run with
one workspace.RunNet per iteration (detectron/tools/train_net.py:165-189) -- the lowered net (one
synchronisation, fused / grouped launches, cached filter packs) against the same net run operator by
operator with a synchronisation after each (the reference executors' model), against the oracle, and
against the hand-built program (head_pipeline.DistillHeads) including the SGD update."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import ssad_amd  # noqa: E402,F401
from oracle import head_step, oracle  # noqa: E402
from ssad_amd import synth  # noqa: E402
from ssad_amd.caffe2_hip import core, dyndep, workspace  # noqa: E402
from ssad_amd.operator_surface import HeadsNetStep  # noqa: E402
from test_gpu_kernels import CONV_FLOOR, CONV_RTOL, close  # noqa: E402
from test_gpu_operators import (SHAPES, assert_typical, close_chain, count_flips, oracle_tower_acts,  # noqa: E402
                                small_problem)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def fresh_workspace():
    assert torch.cuda.is_available()
    dyndep.InitOpsLibrary()
    workspace.ResetWorkspace()
    yield
    workspace.ResetWorkspace()


def fetch_all(step):
    out = {}
    for l in step.levels:
        for stem in ("teacher/retnet_cls_prob_fpn%d", "retnet_cls_pred_fpn%d", "retnet_bbox_pred_fpn%d",
                     "fl_distill_fpn%d", "fl_fpn%d", "retnet_loss_bbox_fpn%d"):
            out[stem % l] = workspace.FetchBlob(stem % l)
        out["fpn_%d_grad" % l] = workspace.FetchBlob(step.grad_map["fpn_%d" % l])
        for tw in ("cls", "bbox"):
            for d in range(step.cfg.num_convs):
                name = "retnet_%s_conv_n%d_fpn%d" % (tw, d, l)
                out[name] = workspace.FetchBlob(name)
    for p in step.student_init:
        out[p + "_grad"] = workspace.FetchBlob(step.grad_map[p])
    out["distill_normalizer"] = workspace.FetchBlob("distill_normalizer")
    return out


def test_lowered_net_vs_operator_by_operator_vs_oracle():
    cfg, S, T, fs, ft, labs, tg, fg = small_problem()
    step = HeadsNetStep(cfg, N=fs[0].shape[0], shapes=SHAPES, student_init=S, teacher_init=T, update=False)
    step.feed_params()
    step.feed_inputs(fs, ft, labs, tg, fg)
    step.create()
    low = step.lowered()
    assert [o.type for o in low["teacher"]].count("ConvGroup") == 6 and not any(o.type == "Sigmoid" for o in low["teacher"])
    assert [o.type for o in low["student"]].count("ConvGradientGroup") == 5
    assert len(low["student"]) < (step.forward_ops + step.backward_ops) // 2

    calls0 = workspace.Counter("conv_launch_calls")
    step.step()                                  # lowered, one synchronisation per net
    calls_lowered = workspace.Counter("conv_launch_calls") - calls0
    got = fetch_all(step)
    step.step(sync_every_op=True)                # the list as written, a synchronisation after every operator
    calls_plain = workspace.Counter("conv_launch_calls") - calls0 - calls_lowered
    plain = fetch_all(step)
    # teacher: 4 tower depths + 2 prediction classes; student forward the same; backward: per group 2 filter
    # gradients + the data-gradient classes -- against one launcher call per (operator, kind) as written
    assert calls_lowered <= 6 + 6 + 5 * 4, calls_lowered
    assert calls_plain >= 50 + 50 + 50 + 45, calls_plain

    ref = head_step.head_step(S, T, fs, ft, labs, scale=cfg.loss_scale, bbox_targets=tg, fg_num=fg,
                              focal_gamma=cfg.focal_gamma, focal_alpha=cfg.focal_alpha,
                              bbox_beta=cfg.bbox_reg_beta)
    acts = oracle_tower_acts(S, fs)
    for res, what in ((got, "lowered"), (plain, "as written")):
        # (activations on the other side of zero than the oracle's: none for this seed -> the tight bound)
        flips = count_flips(lambda tw, d, l: res["retnet_%s_conv_n%d_fpn%d" % (tw, d, step.levels[l])], acts)
        close(res["distill_normalizer"], ref["normalizer"], 1e-5, 0, what + " normalizer")
        for i, l in enumerate(step.levels):
            close(res["teacher/retnet_cls_prob_fpn%d" % l], ref["t_prob"][i], CONV_RTOL, CONV_FLOOR, what + " t prob")
            close(res["retnet_cls_pred_fpn%d" % l], ref["cls_logits"][i], CONV_RTOL, CONV_FLOOR, what + " logits")
            close(res["fl_distill_fpn%d" % l], ref["losses"][i], 2e-4, 0, what + " distill loss")
            close(res["fl_fpn%d" % l], ref["focal_losses"][i], 2e-4, 0, what + " focal loss")
            close(res["retnet_loss_bbox_fpn%d" % l], ref["bbox_losses"][i], 2e-4, 1e-9, what + " bbox loss")
            want = ref["d_fpn"]["cls"][i] + ref["d_fpn"]["bbox"][i]
            close_chain(res["fpn_%d_grad" % l], want, what + " d fpn", None, flips)
        errs = []
        for name, g in ref["grads"].items():
            close_chain(res[name + "_grad"], g, what + " grad " + name, errs, flips)
        assert_typical(errs, what + " grads")
        assert flips == 0, "seed with an activation on the other side of zero: pick another (close_chain)"
    # the two executions of the same graph agree far inside the oracle tolerance: same kernels, the
    # filter gradient summed in one launch instead of five + a Sum
    for k in got:
        close_chain(got[k], plain[k], "lowered vs as written: " + k)


def test_f24_net_arguments_switch_the_engine_and_the_engines_agree():
    """`hip_train_f24` / `hip_frozen_f24` = 0 on a NetDef keep the F(2x2) engine (no hip_algo in the lowered list);
    with the defaults the same nets run on the split-operand engine (>= 256 wide) / F(2x4) (hip_algo = split), with
    `hip_split` = 0 on F(2x4) alone.  Losses agree to 1e-5 relative, teacher probabilities to 2e-5 of their scale,
    every logit to 1e-5 of the tensor's scale."""
    cfg, S, T, fs, ft, labs, tg, fg = small_problem()
    out = {}
    for tag, off in (("split", False), ("f24", None), ("f22", True)):
        workspace.ResetWorkspace()
        step = HeadsNetStep(cfg, N=fs[0].shape[0], shapes=SHAPES, student_init=S, teacher_init=T, update=False)
        if off:
            step.teacher.net.Proto().arg.append(core.MakeArgument("hip_frozen_f24", 0))
            step.student.net.Proto().arg.append(core.MakeArgument("hip_train_f24", 0))
        if off is None:
            step.teacher.net.Proto().arg.append(core.MakeArgument("hip_split", 0))
            step.student.net.Proto().arg.append(core.MakeArgument("hip_split", 0))
        step.feed_params()
        step.feed_inputs(fs, ft, labs, tg, fg)
        step.create()
        algos = {a.s for ops in step.lowered().values() for o in ops for a in o.arg if a.name == "hip_algo"}
        want = set() if off else {b"winograd24"} if off is None else {b"split"}
        assert algos == want or algos == {w.decode() for w in want}, algos
        step.step()
        out[tag] = fetch_all(step)
    for a, b in ((out["f24"], out["f22"]), (out["split"], out["f22"])):
      for l in step.levels:
          for stem, tol in (("fl_distill_fpn%d", 1e-5), ("fl_fpn%d", 1e-5), ("retnet_loss_bbox_fpn%d", 1e-5)):
              assert abs(float(a[stem % l]) - float(b[stem % l])) <= tol * abs(float(b[stem % l])) + 1e-12, stem % l
          for stem, tol in (("teacher/retnet_cls_prob_fpn%d", 2e-5), ("retnet_cls_pred_fpn%d", 1e-5),
                            ("retnet_bbox_pred_fpn%d", 1e-5)):
              x, y = np.asarray(a[stem % l], np.float64), np.asarray(b[stem % l], np.float64)
              assert np.abs(x - y).max() <= tol * np.abs(y).max(), (stem % l, np.abs(x - y).max(), np.abs(y).max())
          assert not np.array_equal(a["retnet_cls_pred_fpn%d" % l], b["retnet_cls_pred_fpn%d" % l])   # another engine


def test_filter_pack_cache_follows_the_blob_version():
    cfg, S, T, fs, ft, labs, tg, fg = small_problem(seed=41)
    step = HeadsNetStep(cfg, N=fs[0].shape[0], shapes=SHAPES, student_init=S, teacher_init=T, update=True, lr=0.01)
    step.feed_params()
    step.feed_inputs(fs, ft, labs, tg, fg)
    step.create()
    p0 = workspace.Counter("filter_packs")
    step.step()
    first = workspace.Counter("filter_packs") - p0
    assert first == 10 + 10 + 10, first            # teacher forward, student forward, student data gradient
    step.step()
    second = workspace.Counter("filter_packs") - p0 - first
    assert second == 20, second                    # the student's filters were updated; the teacher's were not
    w_before = workspace.FetchBlob("retnet_cls_conv_n0_fpn3_w")
    assert not np.array_equal(w_before, S["retnet_cls_conv_n0_fpn3_w"])
    # a FeedBlob is a write too: the teacher repacks exactly the filter that was fed
    workspace.FeedBlob("teacher/retnet_cls_pred_fpn3_w", T["retnet_cls_pred_fpn3_w"] * 2, device_option=step.dev)
    p1 = workspace.Counter("filter_packs")
    workspace.RunNet(step.teacher.net)
    assert workspace.Counter("filter_packs") - p1 == 1
    workspace.RunNet(step.teacher.net)
    assert workspace.Counter("filter_packs") - p1 == 1


def test_training_iterations_on_the_net_match_the_program():
    """Three iterations with the SGD update through RunNet against three of head_pipeline.DistillHeads on
    the same inputs: losses, parameters and momenta."""
    from ssad_amd.head_pipeline import DistillHeads
    cfg, S, T, fs, ft, labs, tg, fg = small_problem(seed=43)
    N = fs[0].shape[0]
    step = HeadsNetStep(cfg, N=N, shapes=SHAPES, student_init=S, teacher_init=T, update=True, lr=0.02)
    step.feed_params()
    step.feed_inputs(fs, ft, labs, tg, fg)
    step.create()
    dev = torch.device("cuda", 0)
    heads = DistillHeads(cfg, N=N, shapes=SHAPES, device=dev, student_init=S, teacher_init=T, lr=0.02)
    t = lambda arrs: [torch.from_numpy(a).to(dev) for a in arrs]
    args = (t(fs), t(ft), t(labs))
    kw = dict(update=True, bbox_targets=[tuple(t(p)) for p in tg], fg_num=torch.from_numpy(fg).to(dev))
    for it in range(3):
        step.step()
        losses = heads.step(*args, **kw)
        torch.cuda.synchronize()
        net_losses = step.losses()
        # iteration 0 sees identical parameters; later ones parameters that went through updates computed by
        # two executions of the graph (ReLU masks within round-off of zero may differ, close_chain's note)
        tol = 2e-4 if it == 0 else 2e-3
        for i, l in enumerate(step.levels):
            close(np.float32(net_losses["fl_distill_fpn%d" % l]), losses[i].cpu().numpy(), tol, 0,
                  "iteration %d distill loss" % it)
            close(np.float32(net_losses["fl_fpn%d" % l]), heads.focal_losses[i].cpu().numpy(), tol, 0,
                  "iteration %d focal loss" % it)
        if it == 0:
            errs = []
            for name in S:
                close_chain(workspace.FetchBlob(name), heads.params[name].cpu().numpy(), "param " + name, errs)
                close_chain(workspace.FetchBlob(name + "_momentum"), heads.moms[name].cpu().numpy(),
                            "momentum " + name)
            assert_typical(errs, "parameters after the first update")
    for name in S:      # still the same trajectory after three updates
        a, b = workspace.FetchBlob(name), heads.params[name].cpu().numpy()
        assert np.linalg.norm(a - b) <= 2e-3 * np.linalg.norm(b) + 1e-12, name
    # and against the oracle's update rule on the first iteration's gradients is test_fused_sgd_step_matches_oracle


def test_net_errors_surface_like_the_reference():
    from ssad_amd.caffe2_hip import _capi
    with pytest.raises(_capi.C2Error, match="does not exist yet"):
        workspace.RunNet("nope")
    cfg, S, T, fs, ft, labs, tg, fg = small_problem(seed=44, N=1)
    step = HeadsNetStep(cfg, N=1, shapes=SHAPES, student_init=S, teacher_init=T, update=False)
    with pytest.raises(_capi.C2Error, match="non-existing input blob"):
        step.create()                                     # parameters not fed yet (operator.cc:44-66)
    step.feed_params()
    step.feed_inputs(fs, ft, labs, tg, fg)
    step.create()
    with pytest.raises(_capi.C2Error, match="already exists"):
        workspace.CreateNet(step.teacher.net)
    assert sorted(workspace.Nets()) == ["student", "teacher"]
    # a shape error inside a group names the operator
    workspace.RunNet(step.teacher.net)
    # a blob the lowering fused away (the teacher's logits under Conv -> Sigmoid) is refused by name, with the remedy,
    # instead of returning stale or missing contents; the list as written produces it again
    lvl = step.levels[0]
    with pytest.raises(_capi.C2Error, match="hip_keep_blobs"):
        workspace.FetchBlob("teacher/retnet_cls_pred_fpn%d" % lvl)
    prob = workspace.FetchBlob("teacher/retnet_cls_prob_fpn%d" % lvl)
    workspace.RunNet(step.teacher.net, sync_every_op=True)
    logits = workspace.FetchBlob("teacher/retnet_cls_pred_fpn%d" % lvl)
    assert np.allclose(1.0 / (1.0 + np.exp(-logits.astype(np.float64))), prob, rtol=1e-4, atol=1e-7)
    workspace.RunNet(step.teacher.net)                    # lowered again: skipped again
    with pytest.raises(_capi.C2Error, match="fused its producer away"):
        workspace.FetchBlob("teacher/retnet_cls_pred_fpn%d" % lvl)
    workspace.FeedBlob("fpn_5", fs[2][:, :100], device_option=step.dev)
    with pytest.raises(_capi.C2Error, match="input channels does not match"):
        workspace.RunNet(step.student.net)
    workspace.DeleteNet("student")
    assert sorted(workspace.Nets()) == ["teacher", "teacher__sync_every_op"]     # (+ the as-written twin made above)


def test_lr_schedule_drives_both_routes_with_momentum_correction():
    """train_net.py:171-173: lr_policy.get_lr_at_iter -> UpdateWorkspaceLr before every iteration; a decay step
    rescales the update history (detector.py:616-648) on the net route (Scale operators) and on the program
    (one pass over the flat buffer) alike."""
    from ssad_amd.head_pipeline import DistillHeads
    from ssad_amd.utils.lr_policy import LrSchedule
    cfg, S, T, fs, ft, labs, tg, fg = small_problem(seed=45, N=1)
    sched = LrSchedule()
    step = HeadsNetStep(cfg, N=1, shapes=SHAPES, student_init=S, teacher_init=T, update=True, lr=0.0)
    step.feed_params()
    step.feed_inputs(fs, ft, labs, tg, fg)
    step.create()
    dev = torch.device("cuda", 0)
    heads = DistillHeads(cfg, N=1, shapes=SHAPES, device=dev, student_init=S, teacher_init=T, lr=0.0)
    t = lambda arrs: [torch.from_numpy(a).to(dev) for a in arrs]
    kw = dict(update=True, bbox_targets=[tuple(t(p)) for p in tg], fg_num=torch.from_numpy(fg).to(dev))
    name = "retnet_cls_conv_n3_fpn3_w"
    for it in (179998, 179999, 180000):                      # the decay step of the distillation yaml
        assert sched.apply(step, it) == sched(it) and sched.apply(heads, it) == sched(it)
        assert np.float32(workspace.FetchBlob("lr")[0]) == sched(it) == np.float32(heads.lr.item())
        if it == 180000:                                     # history rescaled by new / old = 0.1 before the step
            close(workspace.FetchBlob(name + "_momentum"), m_net * np.float32(0.1), 1e-6, 1e-12, "net momentum")
            close(heads.moms[name].cpu().numpy(), m_prog * np.float32(0.1), 1e-6, 1e-12, "program momentum")
        step.step()
        heads.step(t(fs), t(ft), t(labs), **kw)
        torch.cuda.synchronize()
        m_net, m_prog = workspace.FetchBlob(name + "_momentum"), heads.moms[name].cpu().numpy()
        close_chain(m_net, m_prog, "momentum after iteration %d" % it)
