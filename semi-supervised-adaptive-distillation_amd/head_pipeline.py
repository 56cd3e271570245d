"""Fused execution of the hot path: teacher heads forward, student heads
forward, PowSum + SigmoidAdaptiveDistillLoss, student heads backward, gradient
all-reduce and SGD -- the per-GPU work of one training iteration of
`build_generic_retinanet_model_dissstillation`
(detectron/lib/modeling/model_builder.py:373-411) restricted to the subnets.

Same arithmetic as running the operator graph of modeling/retinanet_heads.py
through the workspace (tests check that), but scheduled MI355X-first:

  * the five FPN levels that share a filter run as ONE launch per layer
    (the reference runs 5 cuDNN calls; its DAG executor overlaps them at best);
  * Relu / ReluGradient / Sigmoid live in conv epilogues, the 5-way gradient
    Sum over levels (caffe2/python/core.py:706-741) in the wgrad reduction;
  * filters are repacked once per step, not once per level;
  * all activations, gradients, parameters, parameter gradients and momenta
    are pre-allocated once (HBM is 288 GB; nothing is allocated in the step);
    parameter gradients live in two flat buckets (cls subnet, bbox subnet) so
    the data-parallel exchange is two large RCCL all-reduces that overlap the
    remaining backward instead of ~20 per-tensor ones
    (detectron/lib/modeling/optimizer.py:72-92).

torch provides device memory, streams and torch.distributed only.
"""
from collections import OrderedDict

import numpy as np
import torch

import ctypes as C

from . import kernels as K
from . import program as PR
from . import synth
from .data_parallel import BucketedAllReduce
from .modeling.retinanet_heads import HeadConfig


def head_param_specs(cfg):
    """(name, shape, is_bias, bucket) in the order the backward pass finishes
    them: prediction layers, then tower layers from the deepest to the first,
    cls and bbox subnet interleaved (their layers of equal depth run in the
    same launches).  Bucket "late" (ready first) = predictions + upper half of
    the towers, bucket "early" = lower half."""
    A, C, D = cfg.num_anchors, cfg.num_classes - 1, cfg.fpn_dim
    specs = []

    def add(stem, cout, bucket):
        specs.append((stem + "_w", (cout, D, 3, 3), False, bucket))
        specs.append((stem + "_b", (cout,), True, bucket))
    add("retnet_cls_pred_fpn%d" % cfg.k_min, A * C, "late")
    add("retnet_bbox_pred_fpn%d" % cfg.k_min, 4 * A, "late")
    for i in range(cfg.num_convs - 1, -1, -1):
        bucket = "late" if i >= cfg.num_convs // 2 else "early"
        for tower in ("cls", "bbox"):
            add("retnet_%s_conv_n%d_fpn%d" % (tower, i, cfg.k_min), D, bucket)
    return specs


class FlatParams(object):
    """Parameters (or gradients / momenta) of both subnets in one flat
    buffer; `bucket[name]` is the contiguous slice of one all-reduce bucket."""

    def __init__(self, cfg, device, init=None):
        self.specs = head_param_specs(cfg)
        total = sum(int(np.prod(s)) for _, s, _, _ in self.specs)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.views = OrderedDict()
        self.offsets = {}
        self.bucket = {}
        off = 0
        starts = {}
        for name, shape, _, tower in self.specs:
            n = int(np.prod(shape))
            starts.setdefault(tower, off)
            self.views[name] = self.flat[off:off + n].view(shape)
            self.offsets[name] = off
            off += n
            self.bucket[tower] = self.flat[starts[tower]:off]
        if init is not None:
            for name, arr in init.items():
                self.views[name].copy_(torch.as_tensor(arr))

    def __getitem__(self, name):
        return self.views[name]


class DistillHeads(object):
    """One training iteration of the RetinaNet subnets as a native program
    (program.Program -> ssad_program_run).  distill=False is plain RetinaNet training
    (BASELINE config 2: model_builder.py:98-100,413 without the distillation wrapper): no
    teacher, no PowSum / SigmoidAdaptiveDistillLoss, only SigmoidFocalLoss + SelectSmoothL1Loss."""

    F16 = False

    def __init__(self, cfg=None, N=2, shapes=synth.LEVEL_SHAPES_600, device="cuda",
                 student_init=None, teacher_init=None, teacher_bbox_tower=True,
                 lr=0.01, momentum=0.9, weight_decay=1e-4, process_group=None, world_size=1,
                 distill=True, overlap_wgrad=None, blocked_io=False):
        self.cfg = cfg or HeadConfig()
        self.N, self.shapes, self.device = N, list(shapes), device
        self.distill = bool(distill)
        self.teacher_bbox_tower = teacher_bbox_tower and self.distill
        import os
        self.wino = os.environ.get("SSAD_CONV_ENGINE", "winograd").lower() != "direct"
        # The frozen teacher on the F(2x4, 3x3) engine (conv3x3_winograd24.hip: 3 multiplies per output instead of
        # F(2x2)'s 4, fp32 error ~2e-6 of the output scale -- admissible where nothing back-propagates):
        #   1 (default) its cls_pred layer;  2: also its tower layers, which then leave the launch they shared with the
        #   student's (SSAD_TEACHER_F24; 0 = everything on the F(2x2) engine as in rounds 1-4)
        self.teacher_f24 = int(os.environ.get("SSAD_TEACHER_F24", "1")) if (self.wino and not self.F16) else 0
        # the TRAINED subnets on the F(2x4) engine, a bit mask: 1 = data gradients, 2 = cls_pred forward,
        # 4 = tower forward (with the teacher's towers, one launch)
        self.student_f24 = int(os.environ.get("SSAD_STUDENT_F24", "15")) & 7 if (self.wino and not self.F16) else 0
        # The towers of one depth (teacher + student) are ONE launch on ONE engine, and a filter is packed in the
        # layout of the engine that reads it: decide once.  Bit 4 takes the teacher's towers along; with the teacher
        # pinned to F(2x2) (SSAD_TEACHER_F24=0) a distillation step keeps the shared launch there and bit 4 is void.
        if self.student_f24 & 4:
            if self.teacher_f24:
                self.teacher_f24 = 2
            elif self.distill:
                self.student_f24 &= ~4
        # The split-operand engine (conv3x3_split.hip: fp32 operands as hi + lo fp16, three fp16 MFMAs per pair; meets the
        # direct kernel's parity floor where F(2x4) needs 2e-5) where it measured faster than F(2x4) at config 3's size --
        # bit mask SSAD_SPLIT_CONV: 1 = cls_pred forward (student and teacher), 2 = its data gradient, 4 = tower forward
        # (both networks, one launch per depth), 8 = tower data gradients (16: the backbones' >= 256-wide 3x3 layers,
        # backbone_pipeline.py), 32 = the subnets' >= 128-wide filter gradients (conv3x3_wgrad_split.hip; 64: the
        # backbones' >= 256-wide ones; 128: the backbones' pointwise layers with K, M >= 256 on the split-operand GEMM,
        # 256: their filter gradients, gemm_split.hip); 0 = off.  Default 511: same-box A/B of the step 86.0 -> 83.9 ms
        # for bits 1-16 (every bit pays in the step although the isolated launches are level with F(2x4): the step is
        # power-bound, and the engine spends less of it), 82.6 -> 78.3 ms for bits 32 + 64, 78.9 -> 73.5 ms for bit 128,
        # 72.6 -> 70.4 ms for bit 256 (profiles/r06_experiments.md).
        self.split_conv = int(os.environ.get("SSAD_SPLIT_CONV", "511")) if (self.wino and not self.F16) else 0
        self._split_ops, self._split_ws_need = [], 0
        self.momentum, self.weight_decay = momentum, weight_decay
        self.pg, self.world_size = process_group, world_size
        self.dp = BucketedAllReduce(process_group, world_size)
        cfg = self.cfg
        self.A, self.C, self.D = cfg.num_anchors, cfg.num_classes - 1, cfg.fpn_dim
        self.params = FlatParams(cfg, device, student_init)
        self.teacher = FlatParams(cfg, device, teacher_init) if self.distill else None
        self.grads = FlatParams(cfg, device)
        self.moms = FlatParams(cfg, device)
        nlev = len(self.shapes)
        f32 = dict(dtype=torch.float32, device=device)
        self.lr = torch.full((1,), lr, **f32)
        self.one = torch.ones(nlev, **f32)
        self.losses = torch.zeros(nlev, **f32)           # distillation loss per level
        self.focal_losses = torch.zeros(nlev, **f32)
        self.bbox_losses = torch.zeros(nlev, **f32)
        self.normalizer = torch.zeros(1, **f32)
        self.fg_num = torch.ones(1, **f32)               # bound copy of the step's fg_num
        self.preserved = OrderedDict()     # blobs of a loaded weights file the subnets do not own
        self.timing = None                 # program.Timing while bench.py measures
        self.t_packed = None
        self._teacher_packed = False
        self._overlap_wgrad = overlap_wgrad      # None: environment (SSAD_OVERLAP_WGRAD, default on)
        # fp16 pipeline only: the FPN levels arrive (and their gradients leave) channel-blocked fp16
        # in this object's own buffers -- in_blk[...] is written by the backbone program, dbuf[t][0]
        # read by it (backbone_f16.NativeResNetFPNF16): no NCHW fp32 round trip at the boundary
        self.blocked_io = bool(blocked_io) and self.F16
        self._alloc_buffers()
        self._build_programs()

    # -- buffers ------------------------------------------------------------------
    def _lv(self, ch):
        return [torch.empty((self.N, ch, h, w), dtype=torch.float32, device=self.device)
                for (h, w) in self.shapes]

    def _alloc_buffers(self):
        cfg, lv, D = self.cfg, self._lv, self.D
        # inputs are bound per step (the FPN tensors belong to the caller); these placeholders
        # give the level tables valid addresses until the first bind
        self.fpn_in = lv(D)
        self.t_fpn_in = self.fpn_in
        self.labels = [torch.zeros((self.N, self.A, h, w), dtype=torch.int32, device=self.device)
                       for (h, w) in self.shapes]
        # student activations (kept for backward) and their gradients
        self.act = {t: [lv(D) for _ in range(cfg.num_convs)] for t in ("cls", "bbox")}
        self.cls_logits, self.bbox_pred = lv(self.A * self.C), lv(4 * self.A)
        self.d_cls_logits = lv(self.A * self.C)
        self.d_bbox_pred = lv(4 * self.A)
        # one gradient set per tower depth (not ping-pong): the filter gradient of a layer runs on an
        # auxiliary stream while the data-gradient chain continues, so its dY must stay intact
        self.dbuf = {t: [lv(D) for _ in range(cfg.num_convs + 1)] for t in ("cls", "bbox")}
        self.d_fpn = {t: lv(D) for t in ("cls", "bbox")}
        if self.distill:
            # teacher scratch: two ping-pong feature sets per tower + probabilities
            self.t_buf = {"cls": [lv(D), lv(D)], "bbox": [lv(D), lv(D)]}
            self.t_prob = lv(self.A * self.C)
            self.t_bbox = lv(4 * self.A) if self.teacher_bbox_tower else None

    # -- layer bookkeeping ----------------------------------------------------------
    def _layers(self, tower):
        cfg = self.cfg
        names = ["retnet_%s_conv_n%d_fpn%d" % (tower, i, cfg.k_min) for i in range(cfg.num_convs)]
        names.append("retnet_%s_pred_fpn%d" % (tower, cfg.k_min))
        return names

    # Engine choice per convolution: the Winograd F(2x2,3x3) kernel wherever the output is
    # >= 32 channels wide (every layer of the reference configuration), the direct kernel
    # below that.  SSAD_CONV_ENGINE=direct forces the direct kernel everywhere.
    def _use_wino(self, cout):
        return self.wino and cout >= 32

    def _px(self):
        return self.N * sum(h * w for h, w in self.shapes)

    # -- program construction ---------------------------------------------------------
    def _conv_table(self, problems):
        """problems: [(xs, outs, masks or None, packed or None, bias or None)] -> ssad_conv_level[]"""
        n = sum(len(p[0]) for p in problems)
        arr = (K.ConvLevel * n)()
        k = 0
        for xs, outs, masks, packed, bias in problems:
            for l, x in enumerate(xs):
                arr[k] = K.ConvLevel(x.data_ptr(), outs[l].data_ptr() if outs is not None else 0,
                                     masks[l].data_ptr() if masks is not None else 0,
                                     x.shape[0], x.shape[2], x.shape[3],
                                     packed.data_ptr() if packed is not None else 0,
                                     bias.data_ptr() if bias is not None else 0)
                k += 1
        return arr

    # -- |max| words of the split-operand engines (as backbone_pipeline._measure / _produces) ------------------------
    # One table per pipeline, zeroed by one fill at the start of the forward segment.  A launch's words are one
    # contiguous block in its problem order; the level list of every problem is registered too, so that a filter
    # gradient (one tower's five levels) finds the words of the launch that produced or first read its tensors.
    # Every activation / gradient buffer of the subnets is written by exactly one launch per step.
    AMAX_WORDS = 2048

    def _amax_block(self, lists):
        n = sum(len(l) for l in lists)
        base = self._amax_next
        self._amax_next += n
        if self._amax_next > self.AMAX_WORDS:
            raise K.KernelError("|max| table full")
        self._amax_groups[tuple(t.data_ptr() for l in lists for t in l)] = base
        k = base
        for l in lists:
            self._amax_groups[tuple(t.data_ptr() for t in l)] = k
            k += len(l)
        return base

    def _amax_addr(self, base):
        return self.amax.data_ptr() + 4 * base

    def _amax_known(self, lists):
        return self._amax_groups.get(tuple(t.data_ptr() for l in lists for t in l))

    def _amax_measure(self, P, arr, lists, channels, field=0):
        """Words of the tensors a launch reads (its level table's field 0 = x, 1 = aux): found, or measured here on
        the main stream into a fresh block."""
        base = self._amax_known(lists)
        if base is None:
            base = self._amax_block(lists)
            n = sum(len(l) for l in lists)
            P.add(PR.SPLIT_ABSMAX_LEVELS, 73, i=(n, channels, field), p=(arr, self._amax_addr(base)),
                  work=4.0 * sum(t.numel() for l in lists for t in l), keep=[t for l in lists for t in l], stream=0)
        return base

    def _emit_conv(self, P, problems, Cout, Cin, flags, klass, f24=False, split=False):
        """One launch of independent convolutions of equal (Cout, Cin); f24: on the F(2x4, 3x3) engine; split: on the
        split-operand engine (its workspace is bound by _finish_workspaces)."""
        arr = self._conv_table(problems)
        px = sum(x.shape[0] * x.shape[2] * x.shape[3] for p in problems for x in p[0])
        wino = self._use_wino(Cout)
        if not wino and klass in (2, 3, 4, 16):
            klass = 18
        if split:
            nb = K.lib().ssad_conv3x3_split_workspace_bytes(arr, len(arr), Cin)
            self._split_ws_need = max(self._split_ws_need, nb)
            xa = ya = None
            if self._amax_on:
                xa = self._amax_addr(self._amax_measure(P, arr, [list(p[0]) for p in problems], Cin))
                ya = self._amax_addr(self._amax_block([list(p[1]) for p in problems]))     # folded in by the epilogue
            idx = P.add(PR.CONV3X3, klass, i=(len(arr), Cout, Cin, flags, 3), l=(nb,), p=(arr, None, None, None, xa, ya),
                        work=2.0 * 9 * Cout * Cin * px,
                        keep=[t for p in problems for t in (list(p[0]) + list(p[1] or []) + list(p[2] or []))
                              ] + [t for p in problems for t in p[3:] if t is not None])
            self._split_ops.append((P, idx))
            return idx, arr
        idx = P.add(PR.CONV3X3, klass, i=(len(arr), Cout, Cin, flags, 2 if (f24 and wino) else int(wino)), p=(arr, None, None),
                    work=2.0 * 9 * Cout * Cin * px,
                    keep=[t for p in problems for t in (list(p[0]) + list(p[1] or []) + list(p[2] or []))
                          ] + [t for p in problems for t in p[3:] if t is not None])
        return idx, arr

    def _emit_wgrad(self, P, xs, dys, name, Cout, klass):
        arr = self._conv_table([(xs, None, dys, None, None)])
        # SSAD_SPLIT_CONV bit 32 (default): the >= 128-wide filter gradients on the split-operand engine
        split = bool(self.split_conv & 32) and Cout >= 128 and self.D >= 64
        if split and not K.lib().ssad_conv3x3_wgrad_split_workspace_bytes(arr, len(arr), Cout, self.D):
            split = False           # (a level of 2 GiB or more: the exact engine)
        size_fn = K.lib().ssad_conv3x3_wgrad_split_workspace_bytes if split else K.lib().ssad_conv3x3_wgrad_workspace_bytes
        nb = size_fn(arr, len(arr), Cout, self.D)
        self._wgrad_ws_need = max(getattr(self, "_wgrad_ws_need", 0), nb)
        px = sum(x.shape[0] * x.shape[2] * x.shape[3] for x in xs)
        xa = da = None
        if split and self._amax_on:     # (on the main stream, before the fork)
            xa = self._amax_addr(self._amax_measure(P, arr, [list(xs)], self.D, 0))
            da = self._amax_addr(self._amax_measure(P, arr, [list(dys)], Cout, 1))
        if self._wstream:
            P.fork(self._wstream)       # behind everything enqueued so far (the producer of dys)
        if split:
            klass = {5: 68, 6: 69}.get(klass, klass)
        idx = P.add(PR.CONV3X3_WGRAD, klass if self._use_wino(Cout) else 19,
                    i=(len(arr), Cout, self.D, 0, 1 if split else 0), l=(nb,),
                    p=(arr, self.grads[name + "_w"], self.grads[name + "_b"], None, xa, da),
                    work=2.0 * 9 * Cout * self.D * px, keep=list(xs) + list(dys), stream=self._wstream)
        self._wgrad_ops.append(idx)
        return idx, arr

    def _f24_layer(self, name, cout):
        """Does the TEACHER's layer `name` run on the F(2x4, 3x3) engine?"""
        if not self.teacher_f24 or cout < 128:
            return False
        return "_pred_" in name or self.teacher_f24 >= 2

    def _f24_use(self, who, name, cout, cin, which):
        """Engine of one convolution: who = "teacher" | "student", which = "fwd" | "dgrad" (M = cout / cin)."""
        if who == "teacher":
            return which == "fwd" and self._f24_layer(name, cout)
        m = self.student_f24
        if which == "dgrad":
            return bool(m & 1) and cin >= 128
        if cout < 128:
            return False
        return bool(m & 2) if "_pred_" in name else bool(m & 4)

    def _split_use(self, name, cout, cin, which):
        """Does this convolution run on the split-operand engine?  SSAD_SPLIT_CONV bits: 1 cls_pred forward (both
        networks), 2 its data gradient, 4 tower forward (both networks: one launch per depth), 8 tower data gradient."""
        m = self.split_conv
        if not m or "bbox_pred" in name:
            return False
        if "_cls_pred_" in name:
            return bool(m & (1 if which == "fwd" else 2))
        return bool(m & (4 if which == "fwd" else 8))

    def _alloc_packed(self, params, want_dgrad, f24=None):
        """Packed-filter buffers per layer in the layout of the engine that consumes them:
        -> ({name: (fwd, dgrad)}, wino pack entries, direct pack ops[, F(2x4) pack entries]); f24 = "teacher" |
        "student": whose layers these are (None: no F(2x4) engine, three results)."""
        L = K.lib()
        packed, entries, direct, entries24, entries_sp = {}, [], [], [], []
        for tower in ("cls", "bbox"):
            for name in self._layers(tower):
                w = params[name + "_w"]
                cout, cin = w.shape[0], w.shape[1]
                pf = pd = None
                fw, dw = self._use_wino(cout), self._use_wino(cin)
                f_sp = bool(f24) and fw and self._split_use(name, cout, cin, "fwd")
                d_sp = bool(f24) and want_dgrad and dw and self._split_use(name, cout, cin, "dgrad")
                f_24 = bool(f24) and fw and not f_sp and self._f24_use(f24, name, cout, cin, "fwd")
                d_24 = bool(f24) and want_dgrad and dw and not d_sp and self._f24_use(f24, name, cout, cin, "dgrad")
                nf = (L.ssad_conv_split_filter_floats if f_sp else L.ssad_conv_wino24_filter_floats if f_24
                      else L.ssad_conv_wino_filter_floats if fw else L.ssad_conv_packed_filter_floats)(cout, cin)
                pf = torch.empty(nf, dtype=torch.float32, device=self.device)
                if want_dgrad:
                    nd = (L.ssad_conv_split_filter_floats if d_sp else L.ssad_conv_wino24_filter_floats if d_24
                          else L.ssad_conv_wino_filter_floats if dw else L.ssad_conv_packed_filter_floats)(cin, cout)
                    pd = torch.empty(nd, dtype=torch.float32, device=self.device)
                packed[name] = (pf, pd)
                if f_sp or d_sp:
                    entries_sp.append((w, cout, cin, pf if f_sp else None, pd if d_sp else None))
                xf, xd = (pf if f_24 else None), (pd if d_24 else None)
                if xf is not None or xd is not None:
                    entries24.append((w, cout, cin, xf, xd))
                wf = pf if (fw and not f_24 and not f_sp) else None
                wd = pd if (dw and want_dgrad and not d_24 and not d_sp) else None
                if wf is not None or wd is not None:
                    entries.append((w, cout, cin, wf, wd))
                df, dd = (pf if not fw else None), (pd if (not dw and want_dgrad) else None)
                if df is not None or dd is not None:
                    direct.append((w, cout, cin, df, dd))
        if f24:
            self._entries_split[f24] = entries_sp
        return (packed, entries, direct, entries24) if f24 else (packed, entries, direct)

    def _emit_pack(self, P, entries, direct, entries24=(), entries_sp=()):
        if entries_sp:
            tab = (K.PackEntry * len(entries_sp))()
            nbytes = 0
            for k, (w, cout, cin, pf, pd) in enumerate(entries_sp):
                tab[k] = K.PackEntry(w.data_ptr(), cout, cin, pf.data_ptr() if pf is not None else 0,
                                     pd.data_ptr() if pd is not None else 0)
                nbytes += 4 * (w.numel() + (pf.numel() if pf is not None else 0) + (pd.numel() if pd is not None else 0))
            P.add(PR.WINO_PACK_FILTERS, 1, i=(len(entries_sp), 3), p=(tab,), work=nbytes,
                  keep=[t for e in entries_sp for t in e if isinstance(t, torch.Tensor)])
        if entries24:
            tab = (K.PackEntry * len(entries24))()
            nbytes = 0
            for k, (w, cout, cin, pf, pd) in enumerate(entries24):
                tab[k] = K.PackEntry(w.data_ptr(), cout, cin, pf.data_ptr() if pf is not None else 0,
                                     pd.data_ptr() if pd is not None else 0)
                nbytes += 4 * (w.numel() + (pf.numel() if pf is not None else 0)
                               + (pd.numel() if pd is not None else 0))
            P.add(PR.WINO_PACK_FILTERS, 1, i=(len(entries24), 2), p=(tab,), work=nbytes,
                  keep=[t for e in entries24 for t in e if isinstance(t, torch.Tensor)])
        if entries:
            tab = (K.PackEntry * len(entries))()
            nbytes = 0
            for k, (w, cout, cin, pf, pd) in enumerate(entries):
                tab[k] = K.PackEntry(w.data_ptr(), cout, cin, pf.data_ptr() if pf is not None else 0,
                                     pd.data_ptr() if pd is not None else 0)
                nbytes += 4 * (w.numel() + (pf.numel() if pf is not None else 0)
                               + (pd.numel() if pd is not None else 0))
            P.add(PR.WINO_PACK_FILTERS, 1, i=(len(entries),), p=(tab,), work=nbytes,
                  keep=[t for e in entries for t in e if isinstance(t, torch.Tensor)])
        for w, cout, cin, pf, pd in direct:
            P.add(PR.PACK_FILTER, 1, i=(cout, cin), p=(w, pf, pd),
                  work=4 * (w.numel() + (pf.numel() if pf is not None else 0)
                            + (pd.numel() if pd is not None else 0)))

    def _build_programs(self):
        import os
        # filter gradients on an auxiliary stream beside the data-gradient chain (program.py FORK / JOIN)
        ov = self._overlap_wgrad
        self._wstream = 1 if (os.environ.get("SSAD_OVERLAP_WGRAD", "1") == "1" if ov is None else ov) else 0
        self._wgrad_ops, self._wgrad_ws_need = [], 0
        # the table of |max| words serves the fp32 split engines (not the fp16-storage subclass)
        self._amax_on = bool(self.split_conv) and not self.F16
        self.amax = torch.zeros(self.AMAX_WORDS, dtype=torch.int32, device=self.device)
        self._amax_groups, self._amax_next = {}, 0
        self._in_slots = []          # (table, index, which): entries that read the bound inputs
        self._entries_split = {}
        # filters (the teacher's are frozen: packed by a program of their own, run when they change)
        if self.F16:
            self.packed, s_entries, s_direct = self._alloc_packed(self.params, True)
            s_extra = ()
        else:
            self.packed, s_entries, s_direct, s_entries24 = self._alloc_packed(self.params, True, f24="student")
            s_extra = (s_entries24, self._entries_split.get("student", []))
        if self.distill:
            if self.F16:            # (the fp16 subclass has its own pack layouts and no F(2x4) engine)
                self.t_packed_pairs, t_entries, t_direct = self._alloc_packed(self.teacher, False)
                t_extra = ()
            else:
                self.t_packed_pairs, t_entries, t_direct, t_entries24 = self._alloc_packed(self.teacher, False, f24="teacher")
                t_extra = (t_entries24, self._entries_split.get("teacher", []))
            self.t_packed = {k: v[0] for k, v in self.t_packed_pairs.items()}
            T = self.prog_teacher_pack = PR.Program()
            self._emit_pack(T, t_entries, t_direct, *t_extra)
            T.build()
        P = self.prog = PR.Program()
        P.mark("pack")
        self._emit_pack(P, s_entries, s_direct, *s_extra)
        P.mark("forward")
        if self._amax_on:
            P.add(PR.FILL, 11, p=(self.amax,), f=(0.0,), l=(self.AMAX_WORDS,), work=4.0 * self.AMAX_WORDS)
        self._emit_forward(P)
        P.mark("losses")
        self._emit_losses(P)
        P.mark("backward")
        self._emit_backward(P)
        P.mark("sgd")
        self._emit_sgd(P)
        P.mark("end")
        # the distillation-only variant of the loss segment (no supervised losses: the caller
        # supplies the gradient of the box predictions)
        if self.distill:
            Q = self.prog_distill_only = PR.Program()
            self._emit_losses(Q, supervised=False)
            Q.build()
        self._finish_workspaces(P)
        P.build()

    def _finish_workspaces(self, P):
        # one workspace for every split-engine convolution: they run one after the other on the program's stream
        self.split_ws = torch.empty(max(self._split_ws_need, 16), dtype=torch.uint8, device=self.device)
        for prog, idx in self._split_ops:
            prog.set_ptr(idx, 3, self.split_ws)
        self.wgrad_ws = torch.empty(max(self._wgrad_ws_need, 16), dtype=torch.uint8, device=self.device)
        for idx in self._wgrad_ops:
            P.set_ptr(idx, 3, self.wgrad_ws)

    # -- forward --------------------------------------------------------------------------
    def _emit_forward(self, P):
        """Teacher (test mode) and student subnets.  The tower layers of equal depth
        (teacher/student x cls/bbox) are independent convolutions of the same shape, so each
        depth is ONE launch of up to 20 (level, filter) problems: 12 480 equal workgroups fill
        the 256 CUs to 99 % where five separate launches would each leave a partial last wave.
        Teacher cls_pred carries the Sigmoid epilogue (retinanet_heads.py:153-163)."""
        cfg, D = self.cfg, self.D
        AC, A4 = self.A * self.C, 4 * self.A
        tx = {"cls": self.t_fpn_in, "bbox": self.t_fpn_in}
        sx = {"cls": self.fpn_in, "bbox": self.fpn_in}
        for i in range(cfg.num_convs):
            probs, who = [], []
            for t in ("cls", "bbox"):
                name = self._layers(t)[i]
                if self.distill and (t == "cls" or self.teacher_bbox_tower):
                    out = self.t_buf[t][i & 1]
                    probs.append((tx[t], out, None, self.t_packed_for(name), self.teacher[name + "_b"]))
                    who.append("teacher")
                    tx[t] = out
                out = self.act[t][i]
                probs.append((sx[t], out, None, self.packed[name][0], self.params[name + "_b"]))
                who.append("student")
                sx[t] = out
            if self.split_conv & 4:
                # every tower of this depth on the split-operand engine, one launch (class 28)
                _, arr = self._emit_conv(P, probs, D, D, K.CONV_RELU, 28, split=True)
                if i == 0:
                    k = 0
                    for (xs, _, _, _, _), w in zip(probs, who):
                        for l in range(len(xs)):
                            self._in_slots.append((arr, k, w, l))
                            k += 1
                continue
            if self.student_f24 & 4 and (not self.distill or self._f24_layer(self._layers("cls")[i], D)):
                # every tower of this depth on the F(2x4) engine, one launch (class 23)
                _, arr = self._emit_conv(P, probs, D, D, K.CONV_RELU, 23, f24=True)
                if i == 0:
                    k = 0
                    for (xs, _, _, _, _), w in zip(probs, who):
                        for l in range(len(xs)):
                            self._in_slots.append((arr, k, w, l))
                            k += 1
                continue
            if self.distill and self._f24_layer(self._layers("cls")[i], D):
                # the teacher's towers on the F(2x4) engine: a launch of their own (class 21), the student's on F(2x2)
                tp = [p for p, w in zip(probs, who) if w == "teacher"]
                sp_ = [p for p, w in zip(probs, who) if w == "student"]
                _, arr_t = self._emit_conv(P, tp, D, D, K.CONV_RELU, 21, f24=True)
                # (the student's half alone is 19.5 + 4.5 rounds of the 256 CUs where the four towers together were
                # 39 + 9: these launches have the chip to themselves, so their partial rounds are split)
                _, arr_s = self._emit_conv(P, sp_, D, D, K.CONV_RELU | K.CONV_SPLIT_TAIL, 2)
                if i == 0:
                    for arr, plist, w in ((arr_t, tp, "teacher"), (arr_s, sp_, "student")):
                        k = 0
                        for (xs, _, _, _, _) in plist:
                            for l in range(len(xs)):
                                self._in_slots.append((arr, k, w, l))
                                k += 1
                continue
            _, arr = self._emit_conv(P, probs, D, D, K.CONV_RELU, 2)
            if i == 0:
                k = 0
                for (xs, _, _, _, _), w in zip(probs, who):
                    for l in range(len(xs)):
                        self._in_slots.append((arr, k, w, l))
                        k += 1
        cp, bp = self._layers("cls")[-1], self._layers("bbox")[-1]
        sp = self._split_use(cp, AC, D, "fwd")
        if self.distill:
            f24 = self._f24_layer(cp, AC)
            self._emit_conv(P, [(tx["cls"], self.t_prob, None, self.t_packed_for(cp), self.teacher[cp + "_b"])],
                            AC, D, K.CONV_SIGMOID, 25 if sp else 20 if f24 else 3, f24=f24, split=sp)
        f24 = self._f24_use("student", cp, AC, D, "fwd")
        self._emit_conv(P, [(sx["cls"], self.cls_logits, None, self.packed[cp][0], self.params[cp + "_b"])],
                        AC, D, 0, 26 if sp else 22 if f24 else 3, f24=f24, split=sp)
        probs = [(sx["bbox"], self.bbox_pred, None, self.packed[bp][0], self.params[bp + "_b"])]
        if self.teacher_bbox_tower:
            probs.append((tx["bbox"], self.t_bbox, None, self.t_packed_for(bp), self.teacher[bp + "_b"]))
        self._emit_conv(P, probs, A4, D, 0, 4)

    def t_packed_for(self, name):
        return self.t_packed[name]

    # -- losses -----------------------------------------------------------------------------
    def _distill_kw(self):
        cfg = self.cfg
        return dict(gamma=cfg.distill_gamma, alpha=cfg.distill_alpha, beta=cfg.distill_beta,
                    num_classes=self.C, ignored_label=cfg.ignored_label,
                    scale=cfg.loss_scale * cfg.temperature * cfg.temperature)

    def _cls_table(self, outs):
        arr = (K.DistillLevel * len(self.shapes))()
        for l, x in enumerate(self.cls_logits):
            N, Dd, H, W = x.shape
            arr[l] = K.DistillLevel(x.data_ptr(), self.t_prob[l].data_ptr() if self.distill else 0,
                                    self.labels[l].data_ptr(), outs[l].data_ptr(), N, Dd, H, W)
        self._label_tables.append(arr)
        return arr

    def _emit_losses(self, P, supervised=True):
        """PowSum normaliser (adaptive: computed from the teacher's probabilities,
        retinanet_heads.py:325-329) and the classification / box losses with their gradients
        w.r.t. the predictions (loss gradients = 1.0, utils/blob.py:166-172).  With the
        supervised losses present, both classification losses of the student and their summed
        gradient come from ONE pass over the logits (the reference: 2 forward ops + 2 gradient
        ops + an autograd Sum per level)."""
        cfg, L = self.cfg, K.lib()
        if not hasattr(self, "_label_tables"):
            self._label_tables, self._sl1_tables = [], []
        nlev = len(self.shapes)
        n_logits = sum(x.numel() for x in self.cls_logits)
        n_labels = sum(x.numel() for x in self.labels)
        if self.distill:
            ptrs = (C.c_void_p * nlev)(*[t.data_ptr() for t in self.t_prob])
            sizes = (C.c_int64 * nlev)(*[t.numel() for t in self.t_prob])
            nb = L.ssad_pow_sum_workspace_bytes(nlev)
            ws = torch.zeros(nb, dtype=torch.uint8, device=self.device)     # arrival counters: zero once
            P.add(PR.POW_SUM, 8, i=(nlev,), f=(cfg.logits_power,), l=(nb,), p=(ptrs, sizes, self.normalizer, ws),
                  work=4.0 * n_logits, keep=list(self.t_prob))
        DP = K.DistillParams(**{k: v for k, v in self._distill_kw().items()})
        FP = K.FocalParams(cfg.focal_gamma, cfg.focal_alpha, self.C, cfg.loss_scale)
        if self.distill and supervised:
            if cfg.focal_gamma != 2.0:
                raise K.KernelError("the fused distillation + focal pass specialises focal gamma == 2")
            arr = self._cls_table(self.d_cls_logits)
            nb = L.ssad_cls_losses_fused_workspace_bytes(nlev)
            ws = torch.zeros(nb, dtype=torch.uint8, device=self.device)     # arrival counters: zero once
            P.add(PR.CLS_LOSSES_FUSED, 9, i=(nlev,), l=(nb,),
                  p=(arr, self.normalizer, self.fg_num, DP, FP, self.losses, self.focal_losses, ws),
                  work=12.0 * n_logits + 4.0 * n_labels)
        elif self.distill:
            nb = L.ssad_distill_loss_workspace_bytes(nlev)
            ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
            arr_f = self._cls_table([self.losses[l:l + 1] for l in range(nlev)])
            P.add(PR.DISTILL_FWD, 12, i=(nlev,), l=(nb,), p=(arr_f, self.normalizer, DP, ws),
                  work=8.0 * n_logits + 4.0 * n_labels)
            arr_b = self._cls_table(self.d_cls_logits)
            P.add(PR.DISTILL_BWD, 13, i=(nlev, 1), p=(arr_b, self.normalizer, self.one, DP),
                  work=12.0 * n_logits + 4.0 * n_labels)
        else:
            nb = L.ssad_distill_loss_workspace_bytes(nlev)
            ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
            arr_f = self._cls_table([self.focal_losses[l:l + 1] for l in range(nlev)])
            P.add(PR.FOCAL_FWD, 14, i=(nlev,), l=(nb,), p=(arr_f, self.fg_num, FP, ws),
                  work=4.0 * n_logits + 4.0 * n_labels)
            arr_b = self._cls_table(self.d_cls_logits)
            P.add(PR.FOCAL_BWD, 15, i=(nlev, 1), p=(arr_b, self.fg_num, self.one, FP),
                  work=8.0 * n_logits + 4.0 * n_labels)
        if supervised:
            # SelectSmoothL1Loss per level (retinanet_heads.py:268-280) and its gradient
            tab = (K.SmoothL1Level * nlev)()
            for l, pred in enumerate(self.bbox_pred):
                N, Dd, H, W = pred.shape
                tab[l] = K.SmoothL1Level(pred.data_ptr(), 0, 0, self.bbox_losses[l:l + 1].data_ptr(),
                                         self.d_bbox_pred[l].data_ptr(), N, Dd, H, W, 0)
            self._sl1_tables.append(tab)
            nb = L.ssad_select_smooth_l1_workspace_bytes(nlev)
            ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
            P.add(PR.SMOOTH_L1, 10, i=(nlev, 1), f=(cfg.bbox_reg_beta, cfg.loss_scale * cfg.bbox_reg_weight),
                  l=(nb,), p=(tab, self.fg_num, self.one, ws),
                  work=4.0 * sum(x.numel() for x in self.d_bbox_pred),
                  keep=list(self.bbox_pred) + list(self.d_bbox_pred))

    # -- backward -------------------------------------------------------------------------------
    def _emit_backward(self, P):
        """Backward of both subnets, depth by depth from the prediction layers down.  Per depth:
        the two weight gradients (each already one workgroup per CU) and ONE data-gradient
        launch for both towers.  For tower layers the data gradient carries the ReluGradient
        mask, which is the layer's own post-ReLU input.  The program is cut where a gradient
        bucket is complete (mark "backward_late_done"): its all-reduce is issued there."""
        cfg, D = self.cfg, self.D
        nl = cfg.num_convs
        dy = {"cls": self.d_cls_logits, "bbox": self.d_bbox_pred}
        for t, klass_w in (("cls", 6), ("bbox", 7)):
            name = self._layers(t)[-1]
            x_in = self.act[t][nl - 1]
            Cout = self.params[name + "_b"].numel()
            self._emit_wgrad(P, x_in, dy[t], name, Cout, klass_w)
            out = self.dbuf[t][nl]
            sp = self._split_use(name, Cout, D, "dgrad")
            f24 = self._f24_use("student", name, Cout, D, "dgrad")
            self._emit_conv(P, [(dy[t], out, x_in, self.packed[name][1], None)], D, Cout, K.CONV_MASK_AUX,
                            27 if sp else 24 if f24 else 16, f24=f24, split=sp)
            dy[t] = out
        for li in range(nl - 1, -1, -1):
            probs = []
            for t in ("cls", "bbox"):
                name = self._layers(t)[li]
                x_in = self.act[t][li - 1] if li > 0 else self.fpn_in
                _, arr = self._emit_wgrad(P, x_in, dy[t], name, D, 5)
                if li == 0:
                    for l in range(len(x_in)):
                        self._in_slots.append((arr, l, "student", l))
                out = self.dbuf[t][li] if li > 0 else self.d_fpn[t]
                probs.append((dy[t], out, x_in if li > 0 else None, self.packed[name][1], None))
                dy[t] = out
            sp = self._split_use(name, D, D, "dgrad")
            f24 = self._f24_use("student", name, D, D, "dgrad")
            self._emit_conv(P, probs, D, D, K.CONV_MASK_AUX if li > 0 else 0, 29 if sp else 24 if f24 else 16, f24=f24, split=sp)
            if li == nl // 2:
                P.mark("backward_late_done")
        if "backward_late_done" not in P.marks:
            P.mark("backward_late_done")

    # -- update -----------------------------------------------------------------------------------
    def _emit_sgd(self, P):
        specs = self.params.specs
        tab = (K.SgdSegment * len(specs))()
        for k, (name, shape, is_bias, _) in enumerate(specs):
            tab[k] = K.SgdSegment(self.params.offsets[name], int(np.prod(shape)), int(is_bias), 0, None)
        if "sgd_update" not in P.marks:
            P.mark("sgd_update")            # fp32: nothing precedes the update ("sgd" == "sgd_update")
        P.add(PR.SGD_FLAT, 11, i=(len(specs),), f=(self.momentum, self.weight_decay),
              p=(self.params.flat, self.grads.flat, self.moms.flat, self.lr, tab, None),
              work=4.0 * 6 * self.params.flat.numel())
        P.mark("ls_update")                 # fp32: nothing follows the update

    # -- input binding ------------------------------------------------------------------------------
    def _bind(self, student_fpn=None, teacher_fpn=None, labels=None, bbox_targets=None, fg_num=None):
        """Point the program at this step's input tensors (they belong to the caller and may
        move between steps); only entries whose address changed are rewritten."""
        if student_fpn is not None or teacher_fpn is not None:
            s = list(student_fpn) if student_fpn is not None else self.fpn_in
            t = list(teacher_fpn) if teacher_fpn is not None else self.t_fpn_in
            dev_type = torch.device(self.device).type
            for x in list(s) + list(t if self.distill else []):
                if x.dtype != torch.float32 or not x.is_contiguous() or x.device.type != dev_type:
                    raise K.KernelError("FPN levels must be contiguous float32 tensors on %s" % dev_type)
            if any(tuple(a.shape) != tuple(b.shape) for a, b in zip(s, self.fpn_in)):
                raise K.KernelError("FPN levels do not match the shapes this pipeline was built for")
            self._rebind_inputs(s, t)
            self.fpn_in, self.t_fpn_in = s, t
            self._bound_in = (s, t)                     # keep the tensors alive
        if labels is not None:
            for l, g in enumerate(labels):
                if g.dtype != torch.int32 or not g.is_contiguous() or tuple(g.shape) != tuple(self.labels[l].shape):
                    raise K.KernelError("labels[%d] must be contiguous int32 %r" % (l, tuple(self.labels[l].shape)))
            for arr in self._label_tables:
                for l, g in enumerate(labels):
                    arr[l].labels = g.data_ptr()
            self.labels = list(labels)
        if fg_num is not None:
            self.fg_num.copy_(fg_num.reshape(1), non_blocking=True)
        if bbox_targets is not None:
            for tab in self._sl1_tables:
                for l, (Y, Lc) in enumerate(bbox_targets):
                    M = Y.shape[0] if Y.numel() else 0
                    if M:
                        K._f32c(Y, "Y"); K._f32c(Lc, "L")
                        if tuple(Y.shape) != (M, 4) or tuple(Lc.shape) != (M, 4):
                            raise K.KernelError("bbox targets must be (Y [M,4], L [M,4])")
                    tab[l].Y = Y.data_ptr() if M else 0
                    tab[l].L = Lc.data_ptr() if M else 0
                    tab[l].M = M
            self._bound_targets = bbox_targets

    def _rebind_inputs(self, s, t):
        for arr, k, who, l in self._in_slots:
            arr[k].x = (s if who == "student" else t)[l].data_ptr()

    # -- segments (the names the full model drives) ---------------------------------------------------
    def pack_student(self, want_dgrad=True):
        self.prog.run("pack", "forward", timing=self.timing)

    def pack_teacher(self):
        """The teacher is frozen: packed once (and again after its weights are loaded)."""
        if self.distill:
            self.prog_teacher_pack.run()
            self._teacher_packed = True

    def forward_all(self, teacher_fpn, student_fpn):
        if self.distill and not self._teacher_packed:
            self.pack_teacher()
        self._bind(student_fpn=student_fpn, teacher_fpn=teacher_fpn if self.distill else None)
        self.prog.run("forward", "losses", timing=self.timing)
        return self.cls_logits, self.bbox_pred

    def cls_losses(self, labels, fg_num):
        """With bbox_losses_fwd_bwd: the full reference loss set (kept as two calls for the
        full model's driver; both run in the one "losses" segment)."""
        self._pending_labels, self._pending_fg = labels, fg_num
        return self.losses, self.focal_losses

    def bbox_losses_fwd_bwd(self, bbox_targets, fg_num):
        self._bind(labels=self._pending_labels, bbox_targets=bbox_targets, fg_num=fg_num)
        self.prog.run("losses", "backward", timing=self.timing)
        return self.d_bbox_pred

    def distill_loss(self, labels):
        """Distillation loss only (forward + gradient w.r.t. the logits)."""
        self._bind(labels=labels)
        self.prog_distill_only.run(timing=self.timing)
        return self.losses

    def backward(self, d_bbox_pred=None):
        if d_bbox_pred is not None and d_bbox_pred is not self.d_bbox_pred:
            for dst, src in zip(self.d_bbox_pred, d_bbox_pred):
                dst.copy_(src)
        self.prog.run("backward", "backward_late_done", timing=self.timing)
        self._allreduce_async("late")
        self.prog.run("backward_late_done", "sgd", timing=self.timing)
        self._allreduce_async("early")
        return self.d_fpn

    # -- data parallel ------------------------------------------------------------------
    def _allreduce_async(self, tower):
        if tower in self.grads.bucket:
            self.dp.issue(self.grads.bucket[tower])

    def wait_gradients(self):
        self.dp.wait()

    def broadcast_params(self, src=0):
        """Initial parameter sync (detectron/lib/utils/net.py:185-208 walks all of model.params, which in a
        distillation model holds the teacher's blobs too): student parameters, their history and the frozen
        teacher's parameters."""
        if not self.dp.active:
            return
        tensors = [self.params.flat, self.moms.flat]
        if self.teacher is not None:
            tensors.append(self.teacher.flat)
            self._teacher_packed = False         # repack from the received values
        self.dp.broadcast(tensors, src=src)

    # -- learning rate (detector.py:594-648) ------------------------------------------
    SCALE_MOMENTUM = True             # cfg.SOLVER.SCALE_MOMENTUM (config.py:634)
    SCALE_MOMENTUM_THRESHOLD = 1.1    # config.py:638

    def update_lr(self, new_lr):
        """UpdateWorkspaceLr: set the step's learning rate; when it changes by more
        than the threshold the update history V (= mu*V + lr*grad, so it carries
        the old lr) is rescaled by new/old in one pass over the flat momentum
        buffer (_CorrectMomentum runs one Scale op per parameter)."""
        cur_lr = float(self.lr.item())
        new_lr = float(np.float32(new_lr))
        if cur_lr == new_lr:
            return new_lr
        eps = 1e-10
        ratio = max(new_lr / max(cur_lr, eps), cur_lr / max(new_lr, eps))
        self.lr.fill_(new_lr)
        if self.SCALE_MOMENTUM and cur_lr > 1e-7 and ratio > self.SCALE_MOMENTUM_THRESHOLD:
            K.scale_(self.moms.flat, new_lr / cur_lr)
        return new_lr

    def sgd_step(self):
        """Wait for the reduced gradients, then the whole model's update in one launch."""
        self.wait_gradients()
        self.prog.run("sgd", "end", timing=self.timing)

    # -- one iteration --------------------------------------------------------------------
    def step(self, student_fpn, teacher_fpn, labels, d_bbox_pred=None, update=True,
             bbox_targets=None, fg_num=None):
        """One iteration.  With `bbox_targets` and `fg_num` the student's supervised losses are
        part of the step (the full reference graph); without them only the distillation loss
        drives the cls subnet and `d_bbox_pred` must supply the box-subnet gradient."""
        if bbox_targets is None and not self.distill:
            raise K.KernelError("student-only training needs bbox_targets and fg_num")
        self.pack_student()
        self.forward_all(teacher_fpn, student_fpn)
        if bbox_targets is not None:
            self.cls_losses(labels, fg_num)
            self.bbox_losses_fwd_bwd(bbox_targets, fg_num)
            d_bbox_pred = None
        else:
            self.distill_loss(labels)
        self.backward(d_bbox_pred)
        if update:
            self.sgd_step()
        else:
            self.wait_gradients()
        return self.losses


class DistillHeadsF16(DistillHeads):
    """The same iteration with fp16 storage and fp32 accumulation in the subnet convolutions
    (BASELINE config 5's precision; the reference's only fp16 route is CudnnConvOp<float16>
    with fp32 math, caffe2/operators/conv_op_cudnn.cc:631-636).  Mixed precision in the usual
    arrangement: fp32 master parameters, momentum and parameter gradients (FlatParams, SGD and
    the all-reduce are unchanged); filters re-rounded to fp16 every step; activations between
    the layers channel-blocked fp16; prediction layers write NCHW fp32 for the fp32 loss
    kernels; the gradient of the logits is multiplied by a loss scale before it is rounded to
    fp16 (its elements are ~1e-6) and every result leaving the fp16 domain -- filter / bias
    gradients, the gradient w.r.t. the FPN levels -- is divided by it again.

    The loss scale is DYNAMIC and lives on the device (ssad_loss_scale_update): after the
    gradient all-reduce one pass checks the flat fp32 gradient buffer for Inf / NaN; on
    overflow the SGD launch drops the step (every rank sees the same reduced buffer, so all
    ranks drop it together) and the scale is halved, after LOSS_SCALE_GROWTH_INTERVAL clean
    steps it is doubled.  Nothing of this visits the host."""

    F16 = True
    LOSS_SCALE = 8192.0                 # initial value
    LOSS_SCALE_GROWTH_INTERVAL = 2000
    LOSS_SCALE_MIN, LOSS_SCALE_MAX = 1.0, 65536.0

    def _blk(self, ch):
        return [torch.empty((self.N, (ch + 7) // 8, h, w, 8), dtype=torch.float16, device=self.device)
                for (h, w) in self.shapes]

    def _alloc_buffers(self):
        cfg, lv, blk, D = self.cfg, self._lv, self._blk, self.D
        self.fpn_in = lv(D)
        self.t_fpn_in = self.fpn_in
        self.labels = [torch.zeros((self.N, self.A, h, w), dtype=torch.int32, device=self.device)
                       for (h, w) in self.shapes]
        self.act = {t: [blk(D) for _ in range(cfg.num_convs)] for t in ("cls", "bbox")}
        self.cls_logits, self.bbox_pred = lv(self.A * self.C), lv(4 * self.A)
        self.d_cls_logits, self.d_bbox_pred = lv(self.A * self.C), lv(4 * self.A)
        self.dy_pred = {"cls": blk(self.A * self.C), "bbox": blk(4 * self.A)}
        self.dbuf = {t: [blk(D) for _ in range(cfg.num_convs + 1)] for t in ("cls", "bbox")}
        self.d_fpn = {t: lv(D) for t in ("cls", "bbox")}
        self.in_blk = {"student": blk(D)}
        if self.distill:
            self.in_blk["teacher"] = blk(D)
            self.t_buf = {"cls": [blk(D), blk(D)], "bbox": [blk(D), blk(D)]}
            self.t_prob = lv(self.A * self.C)
            self.t_bbox = lv(4 * self.A) if self.teacher_bbox_tower else None
        # {scale, 1/scale} and {overflow flag, clean steps}
        self.ls_state = torch.tensor([self.LOSS_SCALE, 1.0 / self.LOSS_SCALE], dtype=torch.float32,
                                     device=self.device)
        self.ls_counters = torch.zeros(2, dtype=torch.int32, device=self.device)

    @property
    def loss_scale(self):
        return float(self.ls_state[0])

    def _alloc_packed(self, params, want_dgrad):
        packed, ops = {}, []
        for tower in ("cls", "bbox"):
            for name in self._layers(tower):
                w = params[name + "_w"]
                M, Cc = w.shape[0], w.shape[1]
                n = K.lib().ssad_f16_filter_halves(M, Cc)
                wf = torch.empty(n, dtype=torch.float16, device=self.device)
                wd = torch.empty(n, dtype=torch.float16, device=self.device) if want_dgrad else None
                packed[name] = (wf, wd)
                ops.append((w, M, Cc, wf, wd))
        return packed, ops, []

    def _emit_pack(self, P, entries, direct):
        """All of a network's filters in one launch (ssad_f16_pack_filters)."""
        if not entries:
            return
        tab = (K.F16PackEntry * len(entries))()
        for i, (w, M, Cc, wf, wd) in enumerate(entries):
            tab[i] = K.F16PackEntry(w.data_ptr(), wf.data_ptr(), wd.data_ptr() if wd is not None else None, M, Cc, 9, 0)
        P.add(PR.F16_PACK_FILTERS, 33, i=(len(entries),), p=(tab,),
              work=sum(4.0 * w.numel() + 2.0 * (wf.numel() + (wd.numel() if wd is not None else 0))
                       for w, M, Cc, wf, wd in entries),
              keep=[t for e in entries for t in (e[0], e[3], e[4]) if t is not None])

    def _f16_table(self, problems):
        """problems: [(xs, outs, masks or None, packed or None, bias or None)] -> ssad_f16_level[]"""
        n = sum(len(p[0]) for p in problems)
        arr = (K.F16Level * n)()
        k = 0
        for xs, outs, masks, packed, bias in problems:
            for l, xb in enumerate(xs):
                arr[k].x, arr[k].y = xb.data_ptr(), outs[l].data_ptr()
                arr[k].aux = masks[l].data_ptr() if masks is not None else None
                arr[k].N, arr[k].H, arr[k].W = xb.shape[0], xb.shape[2], xb.shape[3]
                arr[k].packed = packed.data_ptr() if packed is not None else None
                arr[k].bias = bias.data_ptr() if bias is not None else None
                k += 1
        return arr

    def _emit_conv16(self, P, problems, Cin, Cout, flags, klass):
        arr = self._f16_table(problems)
        px = sum(x.shape[0] * x.shape[2] * x.shape[3] for p in problems for x in p[0])
        P.add(PR.F16_CONV3X3, klass, i=(len(arr), Cin, Cout, flags), p=(arr, None, None),
              work=2.0 * 9 * Cout * Cin * px,
              keep=[t for p in problems for t in (list(p[0]) + list(p[1]) + list(p[2] or []))] +
                   [t for p in problems for t in p[3:] if t is not None])

    def _emit_forward(self, P):
        cfg, D = self.cfg, self.D
        AC, A4 = self.A * self.C, 4 * self.A
        self._in_ops = []
        act_bytes = lambda x: 6.0 * x.numel()
        for who in (("student", "teacher") if self.distill else ("student",)) if not self.blocked_io else ():
            src = self.fpn_in if who == "student" else self.t_fpn_in
            for l, (x, xb) in enumerate(zip(src, self.in_blk[who])):
                idx = P.add(PR.F16_PACK_ACT, 32, i=(x.shape[0], x.shape[1], x.shape[2], x.shape[3]), f=(1.0,),
                            p=(x, None, xb), work=act_bytes(x))
                self._in_ops.append((idx, who, l))
        tx = {"cls": self.in_blk.get("teacher"), "bbox": self.in_blk.get("teacher")}
        sx = {"cls": self.in_blk["student"], "bbox": self.in_blk["student"]}
        for i in range(cfg.num_convs):
            # the tower layers of equal depth (teacher / student x cls / bbox) are independent
            # convolutions of one shape: ONE launch of up to 20 (level, filter) problems
            probs = []
            for t in ("cls", "bbox"):
                name = self._layers(t)[i]
                if self.distill and (t == "cls" or self.teacher_bbox_tower):
                    out = self.t_buf[t][i & 1]
                    probs.append((tx[t], out, None, self.t_packed_for(name), self.teacher[name + "_b"]))
                    tx[t] = out
                out = self.act[t][i]
                probs.append((sx[t], out, None, self.packed[name][0], self.params[name + "_b"]))
                sx[t] = out
            self._emit_conv16(P, probs, D, D, K.CONV_RELU, 34)
        cp, bp = self._layers("cls")[-1], self._layers("bbox")[-1]
        F32 = K.F16_OUT_NCHW_F32
        if self.distill:
            self._emit_conv16(P, [(tx["cls"], self.t_prob, None, self.t_packed_for(cp), self.teacher[cp + "_b"])],
                              D, AC, K.CONV_SIGMOID | F32, 35)
        self._emit_conv16(P, [(sx["cls"], self.cls_logits, None, self.packed[cp][0], self.params[cp + "_b"])],
                          D, AC, F32, 35)
        self._emit_conv16(P, [(sx["bbox"], self.bbox_pred, None, self.packed[bp][0], self.params[bp + "_b"])],
                          D, A4, F32, 36)
        if self.teacher_bbox_tower:
            self._emit_conv16(P, [(tx["bbox"], self.t_bbox, None, self.t_packed_for(bp), self.teacher[bp + "_b"])],
                              D, A4, F32, 36)

    def _emit_wgrad16(self, P, xbs, dybs, name, Cin, Cout):
        n = len(xbs)
        arr = (K.F16WgradLevel * n)()
        for l, (xb, dyb) in enumerate(zip(xbs, dybs)):
            arr[l].x, arr[l].dy = xb.data_ptr(), dyb.data_ptr()
            arr[l].N, arr[l].H, arr[l].W = xb.shape[0], xb.shape[2], xb.shape[3]
        nb = K.lib().ssad_conv3x3_wgrad_f16_levels_workspace_bytes(arr, n, Cin, Cout)
        self._wgrad_ws_need = max(self._wgrad_ws_need, nb)
        px = sum(x.shape[0] * x.shape[2] * x.shape[3] for x in xbs)
        if self._wstream:
            P.fork(self._wstream)
        idx = P.add(PR.F16_WGRAD, 37, i=(n, Cin, Cout, 0), f=(1.0,), l=(nb,),
                    p=(arr, self.ls_state[1:2], self.grads[name + "_w"], self.grads[name + "_b"], None),
                    work=2.0 * 9 * Cout * Cin * px, keep=list(xbs) + list(dybs), stream=self._wstream)
        self._wgrad_ops.append(idx)

    def _finish_workspaces(self, P):
        self.wgrad_ws = torch.empty(max(self._wgrad_ws_need, 16), dtype=torch.uint8, device=self.device)
        for idx in self._wgrad_ops:
            P.set_ptr(idx, 4, self.wgrad_ws)

    def _emit_backward(self, P):
        cfg, D = self.cfg, self.D
        nl, nlev = cfg.num_convs, len(self.shapes)
        S, Sinv = self.ls_state[0:1], self.ls_state[1:2]
        dy = {}
        for t, src in (("cls", self.d_cls_logits), ("bbox", self.d_bbox_pred)):
            for l in range(nlev):
                x = src[l]
                P.add(PR.F16_PACK_ACT, 32, i=tuple(x.shape), f=(1.0,), p=(x, S, self.dy_pred[t][l]),
                      work=6.0 * x.numel())
            dy[t] = self.dy_pred[t]
        for t, klass in (("cls", 35), ("bbox", 36)):
            name = self._layers(t)[-1]
            x_in = self.act[t][nl - 1]
            Cout = self.params[name + "_b"].numel()
            self._emit_wgrad16(P, x_in, dy[t], name, D, Cout)
            out = self.dbuf[t][nl]
            self._emit_conv16(P, [(dy[t], out, x_in, self.packed[name][1], None)], Cout, D,
                              K.CONV_MASK_AUX, klass)
            dy[t] = out
        for li in range(nl - 1, -1, -1):
            probs = []
            for t in ("cls", "bbox"):
                name = self._layers(t)[li]
                x_in = self.act[t][li - 1] if li > 0 else self.in_blk["student"]
                self._emit_wgrad16(P, x_in, dy[t], name, D, D)
                out = self.dbuf[t][li]
                probs.append((dy[t], out, x_in if li > 0 else None, self.packed[name][1], None))
                dy[t] = out
            self._emit_conv16(P, probs, D, D, K.CONV_MASK_AUX if li > 0 else 0, 34)
            if li == nl // 2:
                P.mark("backward_late_done")
        for t in ("cls", "bbox") if not self.blocked_io else ():
            for l in range(nlev):
                xb, x = dy[t][l], self.d_fpn[t][l]
                P.add(PR.F16_UNPACK_ACT, 32, i=tuple(x.shape), f=(1.0,), p=(xb, Sinv, x), work=6.0 * x.numel())
        if "backward_late_done" not in P.marks:
            P.mark("backward_late_done")

    def _emit_sgd(self, P):
        # "sgd": finiteness check of the reduced gradients | "sgd_update": the update, dropped on
        # overflow | "ls_update": the scale moves and the flag is cleared.  A model that owns more
        # gradients than the subnets' (backbone_pipeline.NativeDistillModel) runs its own checks on
        # the same flag between "sgd" and "sgd_update" and its own updates before "ls_update".
        n = self.grads.flat.numel()
        P.add(PR.CHECK_FINITE, 38, l=(n,), p=(self.grads.flat, self.ls_counters), work=4.0 * n)
        P.mark("sgd_update")
        idx = len(P.ops)
        DistillHeads._emit_sgd(self, P)
        P.set_ptr(idx, 5, self.ls_counters)            # the update is skipped on overflow
        P.add(PR.LOSS_SCALE_UPDATE, 38, i=(self.LOSS_SCALE_GROWTH_INTERVAL,),
              f=(2.0, 0.5, self.LOSS_SCALE_MIN, self.LOSS_SCALE_MAX), p=(self.ls_state, self.ls_counters))

    def _rebind_inputs(self, s, t):
        for idx, who, l in self._in_ops:
            self.prog.set_ptr(idx, 0, (s if who == "student" else t)[l])
