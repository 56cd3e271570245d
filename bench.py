#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload heads|full]

One "step" = one training iteration of adaptive distillation on one batch of
synthetic COCO-shaped input per GPU: teacher forward + student forward +
PowSum + SigmoidAdaptiveDistillLoss forward/gradient + student backward
(+ gradient all-reduce over RCCL and the SGD update).  Inputs are resident in
HBM before the timed region.  Prints ONE JSON line on rank 0.

Workloads
  heads  the RetinaNet subnets + distillation losses only, fed synthetic FPN features;
  full   BASELINE config "R-50 student + R-101 teacher, bs=16/GPU, 600 px" (default): both
         ResNet-FPN backbones, subnets, losses and both SGD updates as native programs of this
         repo's HIP kernels (ssad_program_run), teacher on a second stream.  This is the
         configuration the metric is quoted on.  `--precision f16` runs every convolution of the
         net with fp16 storage / fp32 accumulation on this repo's kernels (config 5's precision).
         --backbone harness selects round 1's PyTorch backbones (tools/harness: MIOpen / rocBLAS)
         for A/B runs.

Timing: `value` from the wall clock around K steps (barrier + synchronize on both sides, max
over ranks).  Inside the timed region HIP events bracket only the kernel families the
`roofline*` objects quote; `kernels[]` comes from --profile-steps instrumented steps after it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime maps a process's streams round-robin onto GPU_MAX_HW_QUEUES hardware queues (default
# 4).  Measured on one MI355X with the collectives forced onto a one-rank RCCL communicator
# (SSAD_DP_FORCE=1): with 2, 4, 8 or 12 queues a step that issues collectives is 2.5-3 ms slower than
# one that does not (full step 107.3 vs 104.7 ms at 8; fp16 subnets 13.5 vs 11.6 ms) -- two of the
# streams involved (main / teacher / the executor's auxiliary streams / RCCL's) are created four apart
# and then share a queue -- while with 3, 5, 6 or 7 queues the collectives cost nothing (104.4 vs 105.0
# ms at 7; harness 105.7 vs 106.1).  Without collectives the count does not matter (>= 2).  A count
# that is not a multiple of 4 therefore; must be set before the runtime initialises, i.e. before
# torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "7")

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "images/sec (student+teacher fwd + student bwd) R50-FPN distill, 1/2/4/8 MI355X"
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=16)
    ap.add_argument("--workload", default="full",
                    choices=["heads", "full"])
    # BASELINE config 3 by default; config 5: --student r101 --teacher x101-64x4d --px 500
    # --precision f16
    ap.add_argument("--student", default="r50", choices=["r50", "r101"])
    # --teacher none: plain RetinaNet training of the student (BASELINE config 2: with
    # --student r50 --batch-per-gpu 2), no teacher network, no distillation loss
    ap.add_argument("--teacher", default="r101", choices=["none", "r50", "r101", "x101-64x4d"])
    ap.add_argument("--px", type=int, default=600, choices=[600, 500])
    # precision: f32 (the metric's precision, default) or fp16 storage / fp32 accumulation in every
    # convolution of the net (config 5).  An f16 line is NOT the headline number.
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"])
    # backbones: "native" (= "auto") = programs of this repo's kernels in the run's precision (ResNet-50/101
    # students, ResNet / ResNeXt-101-64x4d teachers); "harness" = round 1's PyTorch backbones under
    # tools/harness (MIOpen / rocBLAS), for A/B runs only
    ap.add_argument("--backbone", default="auto", choices=["auto", "native", "harness"])
    ap.add_argument("--profile-steps", type=int, default=3,
                    help="instrumented steps after the timed region (per-family kernel table)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the short config 5 (fp16) and config 2 (student only) legs after the headline run")
    ap.add_argument("--cpu-sample", default="auto")
    return ap.parse_args()


def kernel_report(classes, steps):
    """Per kernel family (timing class of the native program, program.KLASS): launches per
    step, average launch duration (HIP events on the launch stream, taken by
    ssad_program_run inside the timed region), achieved rate and fraction of the bound's peak.
    MFMA-bound families report EXECUTED flops (the Winograd engine runs 1/2.25 of the
    direct-form flops of SURVEY.md 8d) against the dense fp32 MFMA peak; the direct-form
    equivalent is given beside it."""
    from ssad_amd import program as PR
    out = []
    for k in sorted(classes):
        c, meta = classes[k], PR.KLASS.get(k, dict(name="class %d" % k, bound=None))
        if c["launches"] == 0 or c["ms"] <= 0:
            continue
        row = {"class": k, "kernel": meta["name"], "bound": meta["bound"],
               "launches_per_step": round(c["launches"] / float(steps), 2),
               "avg_launch_ms": round(c["ms"] / c["launches"], 4),
               "ms_per_step": round(c["ms"] / steps, 4)}
        rate = c["work"] / (c["ms"] * 1e-3)
        if meta["bound"] in ("mfma", "mfma16"):
            executed = rate / meta.get("exec_div", 2.25) if meta.get("wino") else rate
            row["exec_div"] = meta.get("exec_div", 2.25) if meta.get("wino") else 1.0
            row.update(unit="TFLOP/s", achieved=round(executed / 1e12, 2),
                       direct_equiv_tflops=round(rate / 1e12, 2), peak=PR.PEAK[meta["bound"]] / 1e12,
                       frac=round(executed / PR.PEAK[meta["bound"]], 4),
                       flops_per_launch=c["work"] / c["launches"])
        elif meta["bound"] == "hbm":
            row.update(unit="GB/s", achieved=round(rate / 1e9, 1), peak=PR.PEAK["hbm"] / 1e9,
                       frac=round(rate / PR.PEAK["hbm"], 4), bytes_per_launch=c["work"] / c["launches"])
        out.append(row)
    return out


# which kernel source a timing class's dominant kernel lives in (for the staleness check of `traffic`)
_KERNEL_SOURCE = {"wino_conv_z_kernel": "conv3x3_winograd.hip", "wino24_conv_kernel": "conv3x3_winograd24.hip",
                  "conv3x3_split_kernel": "conv3x3_split.hip", "wino_wgrad_kernel": "conv3x3_wgrad_winograd.hip",
                  "wsplit_kernel": "conv3x3_wgrad_split.hip",
                  "pow_sum_kernel": "distill_loss.hip", "cls_losses_fused_kernel": "distill_loss.hip",
                  "sgd_flat_kernel": "elementwise.hip"}


def pmc_traffic(klass):
    """HBM bytes per launch of one timing class from the committed counter passes over THIS
    command's launches (profiles/rNN_pmc_classes.json: tools/profile_round.sh runs separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --workload heads` and
    tools/pmc_by_class.py attributes the dispatches to classes; read side x 2, the gfx950
    correction calibrated in profiles/r02_pmc_fetch_calib.md).  It is a builder-side capture by
    construction (counters need rocprofv3 around the process), so the line says which round's capture it is
    (`traffic_from_profile_round`) and the figure is WITHHELD -- None, with a warning on stderr -- when the
    capture records the hashes of the kernel sources it ran (r05 onwards) and the class's source has changed
    since.  -> (bytes or None, note, round tag)"""
    try:
        import glob
        import hashlib
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_classes.json")))
        doc = json.load(open(files[-1]))
        tag = os.path.basename(files[-1]).split("_")[0]
        e = doc["classes"].get(str(klass))
        if not e or "hbm_bytes" not in e:
            return None, None, tag
        src = _KERNEL_SOURCE.get(e.get("kernel", ""))
        recorded = (doc.get("kernel_sources") or {}).get(src) if src else None
        note = ("2 x FETCH_SIZE + WRITE_SIZE per dispatch of this class, separate rocprofv3 "
                "--pmc passes over bench.py --workload heads (%s)" % os.path.basename(files[-1]))
        if recorded is None:
            return int(e["hbm_bytes"]), note + "; the capture predates source hashes (not verifiable against this build)", tag
        path = os.path.join(ROOT, "semi-supervised-adaptive-distillation_amd", "csrc", "kernels", src)
        now = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
        if now != recorded:
            sys.stderr.write("bench.py: profiles/%s was captured on another version of %s (%s then, %s now): "
                             "`traffic` of class %s withheld; re-run tools/profile_round.sh\n"
                             % (os.path.basename(files[-1]), src, recorded, now, klass))
            return None, "STALE: %s changed since %s was captured; traffic withheld" % (src, os.path.basename(files[-1])), tag
        return int(e["hbm_bytes"]), note + "; kernel source %s unchanged since the capture" % src, tag
    except Exception:
        return None, None, None


def _cpu_conv_sample(fwd, bwd, budget_s):
    """The convolutions of the subnets step on ONE image, FPN levels from the coarsest upwards while the budget lasts:
    -> (levels used, seconds, flops, pixel share)."""
    from ssad_amd import synth
    rng = np.random.default_rng(7)
    used, t_total, fl_total, px_used = [], 0.0, 0.0, 0
    last = None
    for li in range(len(synth.LEVEL_SHAPES_600) - 1, -1, -1):
        h, w = synth.LEVEL_SHAPES_600[li]
        if last is not None and t_total + 4.5 * last > budget_s:
            break
        t_level = 0.0
        x = rng.standard_normal((1, 256, h, w)).astype(np.float32)
        for M, n_fwd, n_bwd in ((256, 16, 8), (720, 2, 1), (36, 2, 1)):     # towers (student + teacher), cls_pred, bbox_pred
            Wt = (rng.standard_normal((M, 256, 3, 3)) * 0.01).astype(np.float32)
            b = np.zeros(M, np.float32)
            t0 = time.perf_counter()
            for _ in range(n_fwd):
                y = fwd(x, Wt, b)
            for _ in range(n_bwd):
                bwd(x, Wt, y)
            t_level += time.perf_counter() - t0
            fl_total += 2.0 * 9 * 256 * M * h * w * (n_fwd + 2 * n_bwd)
        t_total += t_level
        last = t_level
        px_used += h * w
        used.append("P%d" % (li + 3))
    frac = px_used / float(sum(h * w for h, w in synth.LEVEL_SHAPES_600))
    return used, t_total, fl_total, frac


def cpu_port_conv(budget_s=12.0):
    """The SAME scope and thread count as `reference_conv` for the repo's own CPU restatement (oracle/ssad_oracle.c:
    im2col + GEMM, ONE thread): the convolutions of the subnets step on one image.  Lets `reference_conv` and the port
    be compared like for like; the multi-threaded `value` above covers more (the losses) on more threads."""
    from oracle import oracle
    prev = oracle.num_threads()
    oracle.set_num_threads(1)
    try:
        used, t_total, fl_total, frac = _cpu_conv_sample(
            lambda x, w, b: oracle.conv_forward(x, w, b), lambda x, w, y: oracle.conv_backward(x, w, y), budget_s)
    finally:
        oracle.set_num_threads(prev)
    return {"value": round(frac / t_total, 5), "unit": "images/s", "cores": 1, "kind": "port",
            "scope": "convolutions of the subnets step only (= reference_conv's scope)",
            "gflops": round(fl_total / t_total / 1e9, 2),
            "sample": "1 image, FPN levels %s (%.1f%% of the pixels, scaled by pixel share), 1 thread, %.1f s" % (
                "+".join(reversed(used)), 100 * frac, t_total)}


def cpu_reference_conv(budget_s=25.0):
    """The reference's OWN CPU operators for the convolutions of the subnets step:
    oracle/_ref/libref_conv.so = ConvOp / ConvGradientOp<float, CPUContext> (conv_op_impl.h:31-202,
    358-577: per-image im2col + Eigen GEMM, one thread as the reference builds them) compiled from
    /root/reference by oracle/build_ref_conv.sh.  One image; every convolution a step runs on it (student
    towers + predictors forward and gradient, teacher forward) on the FPN levels from the coarsest
    upwards for as long as the budget lasts.  None when the library was not built."""
    from oracle import oracle
    if oracle.load_ref_conv() is None:
        return None
    used, t_total, fl_total, frac = _cpu_conv_sample(oracle.ref_conv_forward, oracle.ref_conv_backward, budget_s)
    return {"value": round(frac / t_total, 5), "unit": "images/s", "cores": 1, "kind": "reference",
            "scope": "convolutions of the subnets step only",
            "gflops": round(fl_total / t_total / 1e9, 2),
            "sample": "1 image, the convolutions of the subnets step only (no losses, no backbone): student towers + "
                      "predictors forward and gradient, teacher forward, FPN levels %s (%.1f%% of the pixels, scaled by "
                      "pixel share), the reference's compiled ConvOp / ConvGradientOp<float, CPUContext> "
                      "(oracle/_ref/libref_conv.so, Eigen GEMM, 1 thread), %.1f s" % (
                          "+".join(reversed(used)), 100 * frac, t_total)}


def cpu_baseline(args, cfg):
    """The oracle's restatement of the same step (subnets + losses; the
    reference has no CPU operators for the losses)
    timed on the host cores of this box, on a bounded sample: ONE image.  Beside it
    (`reference_conv`) the reference's own compiled CPU convolution operators on the
    convolutions of that step."""
    from oracle import oracle, head_step
    from ssad_amd import synth
    cores = min(os.cpu_count() or 1, 64)   # beyond ~64 threads the 256-row GEMMs stop scaling
    oracle.set_num_threads(cores)
    shapes = synth.LEVEL_SHAPES_600 if (args.cpu_sample == "all" or
                                        (args.cpu_sample == "auto" and cores >= 16)) \
        else synth.LEVEL_SHAPES_600[1:]
    rng = np.random.default_rng(99)
    S, T = synth.head_params(rng), synth.head_params(rng)
    f = synth.fpn_features(rng, 1, shapes)
    labs = [synth.distill_inputs(rng, 1, 9, 80, h, w)[2] for h, w in shapes]
    tg = [synth.bbox_targets(rng, l) for l in labs]
    fg = np.array([max(1, sum(t[0].shape[0] for t in tg))], np.float32)
    t0 = time.time()
    head_step.head_step(S, T, f, f, labs, scale=1.0, bbox_targets=tg, fg_num=fg)
    dt = time.time() - t0
    frac = sum(h * w for h, w in shapes) / float(sum(h * w for h, w in synth.LEVEL_SHAPES_600))
    out = {
        "value": round(frac / dt, 5), "unit": "images/s", "cores": oracle.num_threads(),
        "kind": "port",
        "scope": "subnets + losses (no backbone); like-for-like with the reference's operators: port_conv_1thread vs "
                 "reference_conv (same scope, both 1 thread)",
        "sample": "1 image, subnets+losses only (no backbone), FPN levels %s of 5 (%.1f%% of "
                  "the pixels, scaled by pixel share), OpenMP im2col+GEMM oracle, %.1f s" % (
                      "P3-P7" if len(shapes) == 5 else "P4-P7", 100 * frac, dt)}
    try:
        out["reference_conv"] = cpu_reference_conv()
    except Exception as e:          # the checker library is optional on the box
        out["reference_conv"] = {"error": repr(e)}
    try:
        out["port_conv_1thread"] = cpu_port_conv()
    except Exception as e:
        out["port_conv_1thread"] = {"error": repr(e)}
    return out


def make_workload(args, dev, world, pg, rank):
    """Everything one configuration needs: subnets (+ backbones for --workload full), synthetic
    labels / box targets / images resident in HBM, and step() = one training iteration.
    -> dict(step, heads, model, wl, N, shapes, image_hw, f16, distill, native, hkw, labels, bbox_targets,
    fg_num, gen)"""
    from ssad_amd import synth
    from ssad_amd.head_pipeline import DistillHeads, DistillHeadsF16
    from ssad_amd.modeling.retinanet_heads import HeadConfig
    N = args.batch_per_gpu
    shapes = synth.LEVEL_SHAPES_600 if args.px == 600 else synth.LEVEL_SHAPES_500
    image_hw = (640, 896) if args.px == 600 else (512, 768)
    cfg = HeadConfig(num_gpus=world)
    rng = np.random.default_rng(1234 + rank)
    f16 = args.precision == "f16"
    distill = args.teacher != "none"
    native_ok = args.student in ("r50", "r101") and args.teacher in ("none", "r50", "r101", "x101-64x4d")
    if args.backbone == "native" and not native_ok:
        sys.stderr.write("bench.py: --backbone native covers ResNet-50/101 students and ResNet-50/101 / "
                         "ResNeXt-101-64x4d teachers\n")
        sys.exit(2)
    # every precision runs on native programs of this repo's kernels by default ("harness" = round
    # 1's PyTorch backbones, kept under tools/ for A/B runs)
    native = args.workload == "full" and (args.backbone == "native" or (args.backbone == "auto" and native_ok))
    hkw = dict(blocked_io=True) if (f16 and native and True) else {}
    heads = (DistillHeadsF16 if f16 else DistillHeads)(cfg, N=N, shapes=shapes, device=dev,
                         student_init=synth.head_params(np.random.default_rng(1)),
                         teacher_init=synth.head_params(np.random.default_rng(2)) if distill else None,
                         process_group=pg, world_size=world, lr=1e-4, distill=distill, **hkw)
    heads.broadcast_params()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    labels = [torch.from_numpy(synth.distill_inputs(rng, N, 9, 80, h, w)[2]).to(dev)
              for h, w in shapes] if N <= 4 else None
    if labels is None:
        labels = []
        for h, w in shapes:
            u = torch.rand((N, 9, h, w), device=dev, generator=gen)
            lab = torch.zeros((N, 9, h, w), dtype=torch.int32, device=dev)
            lab[u < 0.05] = -1
            fg = (u >= 0.05) & (u < 0.07)
            lab[fg] = torch.randint(1, 81, (int(fg.sum()),), device=dev, generator=gen,
                                    dtype=torch.int32)
            labels.append(lab)
    # box-regression targets for every foreground anchor (SelectSmoothL1Loss inputs)
    bbox_targets, n_fg = [], 0
    for lab in labels:
        idx = torch.nonzero(lab > 0)
        Lc = torch.stack([idx[:, 0], 4 * idx[:, 1], idx[:, 2], idx[:, 3]], dim=1).float().contiguous()
        Y = (torch.randn((Lc.shape[0], 4), device=dev, generator=gen) * 0.5).contiguous()
        bbox_targets.append((Y, Lc))
        n_fg += Lc.shape[0]
    fg_num = torch.tensor([float(max(n_fg, 1))], device=dev)

    losses_txt = ("PowSum + SigmoidAdaptiveDistillLoss + SigmoidFocalLoss + SelectSmoothL1Loss"
                  if distill else "SigmoidFocalLoss + SelectSmoothL1Loss")
    model = None
    if args.workload == "heads":
        s_fpn = [torch.randn((N, 256, h, w), device=dev, generator=gen) for h, w in shapes]
        t_fpn = [torch.randn((N, 256, h, w), device=dev, generator=gen) for h, w in shapes] if distill else s_fpn

        def step():
            heads.step(s_fpn, t_fpn, labels, bbox_targets=bbox_targets, fg_num=fg_num)
        wl = ("heads-only: RetinaNet cls+bbox subnets (%sstudent fwd+bwd) + %s fwd/bwd + SGD on synthetic "
              "FPN features; the whole step is one native program of this repo's HIP kernels" % (
                  "teacher fwd, " if distill else "", losses_txt))
    else:
        if native:
            from ssad_amd.backbone_pipeline import NativeDistillModel
            model = NativeDistillModel(heads, student_arch=args.student,
                                       teacher_arch=args.teacher if distill else None, N=N, image_hw=image_hw,
                                       device=dev, process_group=pg, world_size=world)
        else:
            from tools.harness.full_model import FullDistillModel
            model = FullDistillModel(heads, student_depth=args.student,
                                     teacher_depth=args.teacher if distill else None, device=dev,
                                     backbone_f16=f16, process_group=pg, world_size=world)
        images = torch.randn((N, 3) + image_hw, device=dev, generator=gen)

        def step():
            model.step(images, labels, bbox_targets, fg_num)

        def pretty(a):
            return a.upper().replace("R", "R-", 1) if a[0] == "r" else a.upper().replace("X", "X-", 1)
        wl = ("%s-FPN student%s, %d px (3x%dx%d): " % (
              pretty(args.student), (" + %s-FPN teacher adaptive distillation" % pretty(args.teacher))
              if distill else " only (plain RetinaNet training, BASELINE config 2)",
              args.px, image_hw[0], image_hw[1]) + model.describe() +
              "; subnets, %s and subnet SGD = one native program of this repo's HIP kernels" % losses_txt)

    return dict(step=step, heads=heads, model=model, wl=wl, N=N, shapes=shapes, image_hw=image_hw, f16=f16,
                distill=distill, native=native, hkw=hkw, labels=labels, bbox_targets=bbox_targets, fg_num=fg_num,
                gen=gen, cfg=cfg)


def also_leg(dev, dom_klass, steps=5, warmup=2, **cfgkw):
    """A short run of ANOTHER BASELINE configuration on this GPU, after (and outside) the headline's
    timed region: built from scratch (make_workload), `warmup` untimed + `steps` timed iterations
    bracketed by synchronize(), the dominant convolution class bracketed by HIP events inside
    them.  Same step, same kernels, same schedule as a `bench.py` run with these flags."""
    import gc
    from ssad_amd import program as PR
    a = argparse.Namespace(workload="full", backbone="auto", **cfgkw)
    W = make_workload(a, dev, 1, None, 0)
    try:
        for _ in range(warmup):
            W["step"]()
        torch.cuda.synchronize()
        timing = PR.Timing().select([dom_klass])
        W["heads"].timing = timing
        W["model"].timing = timing
        t0 = time.perf_counter()
        for _ in range(steps):
            W["step"]()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        W["heads"].timing = None
        W["model"].timing = None
        h = W["heads"]
        losses = [float(v) for v in (h.losses if W["distill"] else h.focal_losses).cpu()]
        ok = all(np.isfinite(losses)) and bool(torch.isfinite(h.params.flat).all()) and \
            bool(torch.isfinite(W["model"].student.params_flat).all())
        dom = {r["class"]: r for r in kernel_report(timing.collect(), steps)}.get(dom_klass)
        out = {"workload": W["wl"], "batch_per_gpu": W["N"], "image": "3x%dx%d" % W["image_hw"],
               "dtype": "f16 storage / f32 accumulate" if W["f16"] else "f32", "steps": steps, "warmup": warmup,
               "ms_per_step": round(dt / steps * 1e3, 3), "images_per_s": round(W["N"] * steps / dt, 2),
               "finite": ok, ("distill_loss" if W["distill"] else "focal_loss"): losses,
               "roofline": dict(kernel=dom["kernel"], bound="mfma", achieved=dom["achieved"], peak=dom["peak"],
                                unit="TFLOP/s", frac=dom["frac"], launches_per_step=dom["launches_per_step"],
                                avg_launch_ms=dom["avg_launch_ms"], flops_per_launch=dom["flops_per_launch"],
                                direct_equiv_tflops=dom["direct_equiv_tflops"]) if dom else None}
        if W["f16"]:
            out["loss_scale"] = float(h.ls_state[0])
        return out
    finally:
        W.clear()
        gc.collect()
        torch.cuda.empty_cache()


def operator_surface_leg(dev, steps=5, warmup=2, N=16):
    """The subnets' training iteration on the DROP-IN route: the reference's graph --
    teacher net + student net with the 121 forward operators of retinanet_heads.py, their gradient operators,
    one NCCLAllreduce per parameter (no-ops without a communicator) and optimizer.py's update operators --
    created once by workspace.CreateNet from the serialized NetDefs and run with one workspace.RunNet per net
    and iteration, as detectron/tools/train_net.py:165-189 does; bs 16, 600 px, FPN features / labels /
    targets resident in workspace blobs.  Timed twice: the lowered nets (csrc/ops/net_lowering.cc; one
    synchronisation per net) and the operator lists as written (`hip_lowering` = 0); beside them the same
    iteration on the hand-built program (head_pipeline.DistillHeads, `bench.py --workload heads`)."""
    import gc
    from ssad_amd import synth
    from ssad_amd.caffe2_hip import workspace
    from ssad_amd.operator_surface import HeadsNetStep
    from ssad_amd.modeling.retinanet_heads import HeadConfig
    out = {"workload": "RetinaNet subnets (teacher + student) + PowSum + SigmoidAdaptiveDistillLoss + SigmoidFocalLoss + "
                       "SelectSmoothL1Loss + backward + NCCLAllreduce + MomentumSGDUpdate as Caffe2-shaped nets: "
                       "CreateNet once, RunNet(teacher) + RunNet(student) per iteration, through the C-ABI",
           "batch_per_gpu": N, "fpn_levels": [list(s) for s in synth.LEVEL_SHAPES_600], "dtype": "f32",
           "steps": steps, "warmup": warmup}
    try:
        for key, lowering in (("lowered", True), ("as_written", False)):
            workspace.ResetWorkspace()
            st = HeadsNetStep(HeadConfig(num_gpus=1), N=N, shapes=synth.LEVEL_SHAPES_600, update=True,
                              lowering=lowering)
            st.feed_params()
            st.feed_synthetic()
            st.create()
            for _ in range(warmup):
                st.step()
            torch.cuda.synchronize()
            p0, c0 = workspace.Counter("filter_packs"), workspace.Counter("conv_launch_calls")
            t0 = time.perf_counter()
            for _ in range(steps):
                st.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            low = st.lowered()
            losses = st.losses()
            out[key] = {"ms_per_step": round(dt / steps * 1e3, 3), "images_per_s": round(N * steps / dt, 2),
                        "operators_written": st.total_ops, "operators_run": len(low["teacher"]) + len(low["student"]),
                        "filter_packs_per_step": (workspace.Counter("filter_packs") - p0) / steps,
                        "conv_launcher_calls_per_step": (workspace.Counter("conv_launch_calls") - c0) / steps,
                        "distill_loss": [losses["fl_distill_fpn%d" % l] for l in st.levels],
                        "finite": bool(np.all(np.isfinite(list(losses.values()))))}
            del st
            workspace.ResetWorkspace()
        # the same iteration on the hand-built program, same sizes, same number of steps (bench.py --workload heads)
        a = argparse.Namespace(workload="heads", backbone="auto", student="r50", teacher="r101", px=600,
                               precision="f32", batch_per_gpu=N)
        W = make_workload(a, dev, 1, None, 0)
        try:
            for _ in range(warmup):
                W["step"]()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                W["step"]()
            torch.cuda.synchronize()
            prog = (time.perf_counter() - t0) / steps * 1e3
        finally:
            W.clear()
        out["program_ms_per_step"] = round(prog, 3)
        out["lowered_over_program"] = round(out["lowered"]["ms_per_step"] / prog, 3)
        out["as_written_over_program"] = round(out["as_written"]["ms_per_step"] / prog, 3)
        return out
    finally:
        workspace.ResetWorkspace()
        gc.collect()
        torch.cuda.empty_cache()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) of this
    same script under torch.distributed.run on the loopback rendezvous -- the command the driver
    itself uses for N > 1 -- and return their exit code.  The reference builds its N replicas from
    one process (detectron/lib/modeling/optimizer.py:33-69); here a replica is a process."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, but this node has %d visible GPU(s); "
                         "refusing to run fewer ranks than asked for\n" % (args.gpus, have))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL needs it on this stack)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if world != args.gpus or world > torch.cuda.device_count():
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d and %d visible GPU(s): one rank per "
                         "GPU, all three must agree\n" % (args.gpus, world, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or os.environ.get("SSAD_DP_FORCE") == "1":
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        pg = dist.group.WORLD

    import ssad_amd  # noqa: F401
    from ssad_amd import kernels as K, program as PR
    K.lib()   # fail loudly if the HIP extension is missing

    W = make_workload(args, dev, world, pg, rank)
    step, heads, model, wl = W["step"], W["heads"], W["model"], W["wl"]
    N, shapes, image_hw, f16, distill, native, hkw = (W[k] for k in ("N", "shapes", "image_hw", "f16", "distill",
                                                                     "native", "hkw"))
    labels, bbox_targets, fg_num, gen, cfg = W["labels"], W["bbox_targets"], W["fg_num"], W["gen"], W["cfg"]
    # The step's critical path is the stream step() is called on (student forward, subnets, data gradients); the
    # teacher, the filter gradients and the collectives run on other streams and fill the chip beside it.
    # The step therefore runs on a high-priority stream (HIP has two levels: 0 and -1): config 3 94.3 -> 93.3 ms,
    # config 5 26.7 -> 26.5 ms in same-call A/B (round 4).
    _prio = -1
    if _prio:
        _main = torch.cuda.Stream(priority=_prio)
        _main.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(_main)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # In the timed region only the families the roofline objects quote are bracketed by events (the
    # dominant convolution, the loss kernels, PowSum): events between all ~500 launches of a step keep
    # consecutive kernels from overlapping their tails and cost 1 % of the step (2.5 % with collectives
    # in flight).  `kernels[]` comes from instrumented steps after the timed region.
    ROOFLINE_CLASSES = [2, 23, 28, 18, 34, 8, 9, 15]
    timing = PR.Timing().select(ROOFLINE_CLASSES)
    heads.timing = timing
    if args.workload == "full":
        model.timing = timing
    from ssad_amd.data_parallel import BucketedAllReduce
    coll0 = BucketedAllReduce.issued_total
    t0 = time.perf_counter()
    host = 0.0                      # time the launching thread spends inside step()
    for _ in range(args.steps):
        th = time.perf_counter()
        step()
        host += time.perf_counter() - th
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    collectives_per_step = (BucketedAllReduce.issued_total - coll0) / float(args.steps)
    # every family, in `args.profile_steps` instrumented steps outside the timed region (all ranks:
    # the steps contain collectives)
    timing_all = PR.Timing()
    heads.timing = timing_all
    if args.workload == "full":
        model.timing = timing_all
    for _ in range(args.profile_steps):
        step()
    torch.cuda.synchronize()
    heads.timing = None
    if args.workload == "full":
        model.timing = None
    # What enqueueing a step costs the launching thread by itself: with an EMPTY queue (the time spent
    # inside step() during the timed region above also contains the waits on a full HIP queue -- the
    # GPU is the bottleneck -- and says nothing about the host)
    pure = []
    for _ in range(3):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step()
        pure.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax[0])
    loss_val = [float(v) for v in heads.losses.cpu()]
    assert all(np.isfinite(loss_val)), "non-finite distillation loss: %r" % (loss_val,)
    assert bool(torch.isfinite(heads.params.flat).all()), "non-finite subnet parameters after the run"
    if args.workload == "full" and native:
        assert bool(torch.isfinite(model.student.params_flat).all()), "non-finite backbone parameters after the run"

    # Every rank empties its C stdio buffer NOW (RCCL's version banner sits there until process exit otherwise and
    # would land after rank 0's result line in the launcher's merged stdout), then all meet, then rank 0 prints.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if pg is not None:
        torch.distributed.barrier()
    if rank == 0:
        rows = kernel_report(timing_all.collect(), max(args.profile_steps, 1))      # all families
        by = {r["class"]: r for r in kernel_report(timing.collect(), args.steps)}     # the timed region
        # the subnet tower forward launch: 28 on the split-operand engine (SSAD_SPLIT_CONV & 4, the default), 23 on
        # F(2x4), 2 on F(2x2), 18 on the direct kernel
        dom_k = 34 if f16 else (28 if 28 in by else 2 if 2 in by else 23 if 23 in by else 18)
        dom = by.get(dom_k)
        traffic, traffic_note, traffic_round = pmc_traffic(dom_k) if not f16 else (None, None, None)
        # algorithmic HBM bytes of one launch of the dominant class: the four towers' inputs + outputs of
        # all levels + the packed filters, each once (fp32)
        px = N * sum(h * w for h, w in shapes)
        ntow = (4 if distill else 2) if dom_k in (2, 23, 28) else 1
        dom_alg_bytes = ntow * (2 * 256 * px * 4 + 16 * 256 * 256 * 4) if dom_k in (2, 23, 28) else None
        heads_ms = sum(r["ms_per_step"] for r in rows if r["class"] < 48)
        backbone_ms = sum(r["ms_per_step"] for r in rows if r["class"] >= 48)
        out = {
            "metric": METRIC, "value": round(world * N * args.steps / dt, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(min(pure) * 1e3, 3),
            "host_enqueue_note": ("launching thread's time to enqueue one whole step into an empty queue; "
                                  "host_in_step_ms_per_step = time inside step() during the timed region, "
                                  "which includes blocking on the full HIP queue of a GPU-bound step"),
            "host_in_step_ms_per_step": round(host / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if not f16 else
                      "f16 storage / f32 accumulate (subnets: this repo's kernels" + (
                          ")" if args.workload != "full" else
                          "; backbones: this repo's fp16 kernels)" if (native and hkw) else
                          "; backbones: this repo's fp32 kernels)" if native else
                          "; backbones: torch autocast on MIOpen / rocBLAS)")),
            "dtype_note": (None if f16 else
                           "fp32 storage and fp32 accumulation everywhere.  The wide convolutions run on the split-operand "
                           "engines by default -- forward, data gradient AND filter gradient of the subnets' towers and "
                           "cls_pred, of the backbones' >= 256-wide 3x3 layers and of their pointwise layers with both "
                           "channel counts >= 256: each fp32 operand enters the fp16 matrix pipe as hi + lo (22 significant "
                           "bits under a per-tensor power-of-two scale from the tensor's measured |max|), three MFMAs per "
                           "operand pair, measured error against float64 1.2e-6 of the output scale at K = 2304 (fp32 "
                           "accumulation order; F(2x4) fp32 Winograd: 1.5-2.2e-6) and held to the direct fp32 kernels' "
                           "parity bar (tests/test_gpu_kernels.py, test_gpu_gemm_conv.py).  also.cfg3_fp32_mfma_only = the "
                           "same step with those engines off (SSAD_SPLIT_CONV=0: fp32 MFMA instructions only)."),
            "data": "synthetic",
            "config": {"workload": wl, "batch_per_gpu": N, "image": "3x%dx%d" % image_hw,
                       "fpn_levels": [list(s) for s in shapes], "anchors": 9, "classes": 80,
                       "parallelism": "dp%d" % world,
                       # bucket all-reduces this rank started per timed step (RCCL; 0 on one GPU unless
                       # SSAD_DP_FORCE=1 forces them onto a one-rank communicator)
                       "collectives_per_step": collectives_per_step,
                       "schedule": ("one process, HIP streams: student on the main stream%s, filter gradients on "
                                    "auxiliary streams, frozen teacher on a side stream" % (
                                        " (high priority)" if _prio < 0 else "") + (
                                        "; the teacher's forward pass of a step is ordered after the previous "
                                        "step's last reader of its outputs (not after the previous update), so it "
                                        "may run beside the previous step's backward pass -- every timed step "
                                        "still enqueues and executes its own teacher forward inside the timed "
                                        "region" if getattr(model, "_teacher_ahead", False) and
                                        getattr(model, "side", None) is not None else "")
                                    ) if args.workload == "full" and native else None,
                       "distill_loss": loss_val,
                       "focal_loss": [float(v) for v in heads.focal_losses.cpu()],
                       "bbox_loss": [float(v) for v in heads.bbox_losses.cpu()]},
            # the dominant kernel: frac = EXECUTED MFMA flops / dense peak (a hardware fraction)
            "roofline": dict(kernel=dom["kernel"], bound="mfma", achieved=dom["achieved"], peak=dom["peak"],
                             peak_note=("dense fp16 MFMA peak (the split-operand engine executes 3 fp16 products per "
                                        "direct-form product); against the fp32 MFMA peak of 157.3 TFLOP/s the "
                                        "direct-form rate is x%.2f" % (dom["direct_equiv_tflops"] / 157.3)
                                        if dom_k == 28 else None),
                             unit="TFLOP/s", frac=dom["frac"], traffic=traffic, traffic_note=traffic_note,
                             traffic_from_profile_round=traffic_round,
                             algorithmic_bytes=dom_alg_bytes,
                             traffic_over_algorithmic=(round(traffic / dom_alg_bytes, 2)
                                                       if (traffic and dom_alg_bytes) else None),
                             direct_equiv_tflops=dom["direct_equiv_tflops"],
                             exec_div=dom.get("exec_div"),
                             achieved_note=("executed MFMA FLOP/s: algorithmic direct-form flops (2*9*Cout*Cin "
                                            "per output pixel, SURVEY 8d) / exec_div -- 1/3 for the split-operand "
                                            "engine (three fp16 MFMA products per direct-form product, the call "
                                            "includes its split pass; the |max| words come from the pipeline's table, "
                                            "timing class 73), 3 for Winograd F(2x4,3x3) (24 "
                                            "products per 8 outputs), 2.25 for F(2x2,3x3); direct_equiv_tflops is "
                                            "the direct-form rate" if not f16
                                            else "algorithmic direct-form FLOP/s; the kernel executes exactly these"),
                             launches_per_step=dom["launches_per_step"], avg_launch_ms=dom["avg_launch_ms"],
                             flops_per_launch=dom["flops_per_launch"]) if dom else None,
            "kernels": rows,
            "kernels_note": ("every kernel family, from %d instrumented steps AFTER the timed region (events "
                             "around every launch); roofline / roofline_loss / roofline_pow_sum are measured "
                             "inside the timed region" % args.profile_steps),
            "subnets_ms_per_step": round(heads_ms, 3),
            "backbone_kernels_ms_per_step": round(backbone_ms, 3),
        }
        for key, k in (("roofline_loss", 9 if distill else 15), ("roofline_pow_sum", 8)):
            r = by.get(k)
            if r:
                tr, tnote, tround = pmc_traffic(k) if not f16 else (None, None, None)
                out[key] = dict(kernel=r["kernel"], bound="hbm", achieved=r["achieved"], peak=r["peak"],
                                unit="GB/s", frac=r["frac"], launches_per_step=r["launches_per_step"],
                                avg_launch_ms=r["avg_launch_ms"], bytes_per_launch=r["bytes_per_launch"],
                                traffic=tr, traffic_note=tnote, traffic_from_profile_round=tround,
                                traffic_over_algorithmic=(round(tr / r["bytes_per_launch"], 3) if tr else None))
        if not args.no_cpu_baseline and world == 1 and distill:     # rank 0 at N=1 only
            cb = cpu_baseline(args, cfg)
            # the same scope on the GPU: the subnets + losses + SGD step alone (outside the timed region;
            # with blocked fp16 I/O the subnets' inputs are whatever the backbones left in their buffers)
            sf = [torch.randn((N, 256, h, w), device=dev, generator=gen) for h, w in shapes] if not hkw else None
            for _ in range(2):
                heads.step(sf, sf, labels, bbox_targets=bbox_targets, fg_num=fg_num)
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(5):
                heads.step(sf, sf, labels, bbox_targets=bbox_targets, fg_num=fg_num)
            torch.cuda.synchronize()
            cb["gpu_same_scope_images_per_s"] = round(5 * N / (time.perf_counter() - th), 2)
            cb["scope_note"] = ("value and gpu_same_scope_images_per_s both cover subnets + losses (no backbone); "
                                "the headline `value` covers the whole step")
            out["cpu_baseline"] = cb
        # BASELINE's other single-GPU workloads, each a short run of its own AFTER the headline's timed region
        # (the headline `value` above is untouched by them): config 5 = R-101 student + ResNeXt-101-64x4d
        # teacher, 500 px, every convolution fp16 storage / fp32 accumulation; config 2 = R-50 student only,
        # bs 2, 600 px.  Only from the default headline run on one GPU.
        default_cfg3 = (args.workload == "full" and native and not f16 and distill and args.student == "r50" and
                        args.teacher == "r101" and args.px == 600 and N == 16)
        if world == 1 and pg is None and default_cfg3 and not args.no_also:
            del step, heads, model, W
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            also = {}
            for key, kl, kw in (("cfg5_f16", 34, dict(student="r101", teacher="x101-64x4d", px=500, precision="f16",
                                                      batch_per_gpu=16)),
                                ("cfg2", 28, dict(student="r50", teacher="none", px=600, precision="f32",
                                                 batch_per_gpu=2))):
                try:
                    also[key] = also_leg(dev, kl, **kw)
                except Exception as e:      # never lose the headline line to a side leg
                    also[key] = {"error": repr(e)}
            try:
                also["operator_surface"] = operator_surface_leg(dev)
            except Exception as e:
                also["operator_surface"] = {"error": repr(e)}
            # the headline configuration with every 3x3 convolution on fp32 MFMA instructions only (the split-operand
            # engine off: SSAD_SPLIT_CONV=0 -> Winograd F(2x4) / F(2x2) as in round 5), for whoever wants the step
            # without fp16-pipe products
            prev = os.environ.get("SSAD_SPLIT_CONV")
            os.environ["SSAD_SPLIT_CONV"] = "0"
            try:
                also["cfg3_fp32_mfma_only"] = also_leg(dev, 23, student="r50", teacher="r101", px=600, precision="f32",
                                                       batch_per_gpu=16)
            except Exception as e:
                also["cfg3_fp32_mfma_only"] = {"error": repr(e)}
            finally:
                if prev is None:
                    os.environ.pop("SSAD_SPLIT_CONV", None)
                else:
                    os.environ["SSAD_SPLIT_CONV"] = prev
            out["also"] = also
            out["also_note"] = ("other BASELINE configs on the same GPU, each built from scratch and run for a few "
                                "steps AFTER the timed region of the headline, on the same high-priority stream; "
                                "`value` / `ms_per_step` above are config 3 only.  operator_surface = the subnets' "
                                "iteration through workspace.CreateNet / RunNet (the drop-in route) beside the program")
        # the ONE result line, last on rank 0's stdout (with NCCL_DEBUG=VERSION in the environment
        # RCCL prints its version banner to stdout when the communicator is created, i.e. earlier)
        # ... into C stdio's buffer, which is flushed at process exit, i.e. AFTER anything Python prints: measured on
        # the box with a forced one-rank communicator, the five banner lines followed the JSON line in the file.
        # Flush the C streams first so that the result line is the last line of stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
