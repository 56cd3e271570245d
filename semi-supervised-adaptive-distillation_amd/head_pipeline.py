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

from . import kernels as K
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
        self.bucket = {}
        off = 0
        starts = {}
        for name, shape, _, tower in self.specs:
            n = int(np.prod(shape))
            starts.setdefault(tower, off)
            self.views[name] = self.flat[off:off + n].view(shape)
            off += n
            self.bucket[tower] = self.flat[starts[tower]:off]
        if init is not None:
            for name, arr in init.items():
                self.views[name].copy_(torch.as_tensor(arr))

    def __getitem__(self, name):
        return self.views[name]


class DistillHeads(object):
    def __init__(self, cfg=None, N=2, shapes=synth.LEVEL_SHAPES_600, device="cuda",
                 student_init=None, teacher_init=None, teacher_bbox_tower=True,
                 lr=0.01, momentum=0.9, weight_decay=1e-4, process_group=None, world_size=1):
        self.cfg = cfg or HeadConfig()
        self.N, self.shapes, self.device = N, list(shapes), device
        self.teacher_bbox_tower = teacher_bbox_tower
        import os
        self.wino = os.environ.get("SSAD_CONV_ENGINE", "winograd").lower() != "direct"
        self.momentum, self.weight_decay = momentum, weight_decay
        self.pg, self.world_size = process_group, world_size
        self.dp = BucketedAllReduce(process_group, world_size)
        cfg = self.cfg
        self.A, self.C, self.D = cfg.num_anchors, cfg.num_classes - 1, cfg.fpn_dim
        self.params = FlatParams(cfg, device, student_init)
        self.teacher = FlatParams(cfg, device, teacher_init)
        self.grads = FlatParams(cfg, device)
        self.moms = FlatParams(cfg, device)
        self.lr = torch.full((1,), lr, dtype=torch.float32, device=device)
        self.one = torch.ones(len(self.shapes), dtype=torch.float32, device=device)
        self.focal_losses = None
        self.bbox_losses = None
        self.preserved = OrderedDict()     # blobs of a loaded weights file the subnets do not own

        self._alloc_buffers()

    def _lv(self, ch):
        return [torch.empty((self.N, ch, h, w), dtype=torch.float32, device=self.device)
                for (h, w) in self.shapes]

    def _alloc_buffers(self):
        cfg, lv, D = self.cfg, self._lv, self.D
        # student activations (kept for backward) and their gradients
        self.act = {t: [lv(D) for _ in range(cfg.num_convs)] for t in ("cls", "bbox")}
        self.cls_logits, self.bbox_pred = lv(self.A * self.C), lv(4 * self.A)
        self.d_cls_logits = lv(self.A * self.C)
        self.d_bbox_pred = lv(4 * self.A)
        self.dbuf = {"cls": [lv(D), lv(D)], "bbox": [lv(D), lv(D)]}   # ping-pong tower gradients
        self.d_fpn = {t: lv(D) for t in ("cls", "bbox")}
        # teacher scratch: two ping-pong feature sets per tower + probabilities
        self.t_buf = {"cls": [lv(D), lv(D)], "bbox": [lv(D), lv(D)]}
        self.t_prob = lv(self.A * self.C)
        self.t_bbox = lv(4 * self.A) if self.teacher_bbox_tower else None
        # packed filters (rebuilt every step from the current weights)
        self.packed = {}
        self.t_packed = None
        self.losses = None
        self.normalizer = None

    # -- parameters -----------------------------------------------------------
    def _layers(self, tower):
        cfg = self.cfg
        names = ["retnet_%s_conv_n%d_fpn%d" % (tower, i, cfg.k_min) for i in range(cfg.num_convs)]
        names.append("retnet_%s_pred_fpn%d" % (tower, cfg.k_min))
        return names

    # Engine choice per convolution: the Winograd F(2x2,3x3) kernel wherever the
    # output is >= 128 channels wide (all tower layers, cls_pred, every data
    # gradient), the direct kernel for the 36-channel bbox_pred forward.
    # SSAD_CONV_ENGINE=direct forces the direct kernel everywhere.
    def _use_wino(self, cout):
        return self.wino and cout >= 32

    def _pack(self, w, want_fwd, want_dgrad):
        """-> (fwd_packed, dgrad_packed) in the layout of the engine that will
        consume each (forward: Cout outputs; data gradient: Cin outputs)."""
        cout, cin = w.shape[0], w.shape[1]
        pf = pd = None
        if want_fwd:
            pf = (K.conv_wino_pack_filter(w, True, False)[0] if self._use_wino(cout)
                  else K.conv_pack_filter(w, True, False)[0])
        if want_dgrad:
            pd = (K.conv_wino_pack_filter(w, False, True)[1] if self._use_wino(cin)
                  else K.conv_pack_filter(w, False, True)[1])
        return pf, pd

    def pack_student(self, want_dgrad=True):
        for tower in ("cls", "bbox"):
            for name in self._layers(tower):
                self.packed[name] = self._pack(self.params[name + "_w"], True, want_dgrad)

    def pack_teacher(self):
        """The teacher is frozen: pack once."""
        self.t_packed = {}
        for tower in ("cls", "bbox"):
            for name in self._layers(tower):
                self.t_packed[name] = self._pack(self.teacher[name + "_w"], True, False)[0]

    # -- forward ----------------------------------------------------------------
    def forward_all(self, teacher_fpn, student_fpn):
        """Teacher (test mode) and student subnets.  The four tower layers of
        equal depth (teacher/student x cls/bbox) are independent convolutions
        of the same shape, so each depth is ONE launch of 20 (level, filter)
        problems: 12 480 equal workgroups fill the 256 CUs to 99 % where five
        separate launches would each leave a partial last wave.  Teacher
        cls_pred carries the Sigmoid epilogue (retinanet_heads.py:153-163)."""
        if self.t_packed is None:
            self.pack_teacher()
        self.fpn_in = student_fpn
        cfg = self.cfg
        tx = {"cls": teacher_fpn, "bbox": teacher_fpn}
        sx = {"cls": student_fpn, "bbox": student_fpn}
        for i in range(cfg.num_convs):
            probs = []
            for t in ("cls", "bbox"):
                name = self._layers(t)[i]
                if t == "cls" or self.teacher_bbox_tower:
                    out = self.t_buf[t][i & 1]
                    probs.append(dict(xs=tx[t], packed=self.t_packed[name],
                                      bias=self.teacher[name + "_b"], out=out))
                    tx[t] = out
                out = self.act[t][i]
                probs.append(dict(xs=sx[t], packed=self.packed[name][0],
                                  bias=self.params[name + "_b"], out=out))
                sx[t] = out
            K.conv3x3_forward_multi(probs, self.D, relu=True, wino=self._use_wino(self.D))
        cp = self._layers("cls")[-1]
        wn = self._use_wino(self.A * self.C)
        K.conv3x3_forward(tx["cls"], self.t_packed[cp], self.teacher[cp + "_b"], self.A * self.C,
                          sigmoid=True, out=self.t_prob, wino=wn)
        K.conv3x3_forward(sx["cls"], self.packed[cp][0], self.params[cp + "_b"], self.A * self.C,
                          out=self.cls_logits, wino=wn)
        bp = self._layers("bbox")[-1]
        probs = [dict(xs=sx["bbox"], packed=self.packed[bp][0], bias=self.params[bp + "_b"],
                      out=self.bbox_pred)]
        if self.teacher_bbox_tower:
            probs.append(dict(xs=tx["bbox"], packed=self.t_packed[bp],
                              bias=self.teacher[bp + "_b"], out=self.t_bbox))
        K.conv3x3_forward_multi(probs, 4 * self.A, wino=self._use_wino(4 * self.A))
        return self.cls_logits, self.bbox_pred

    # -- losses -------------------------------------------------------------------
    def distill_loss(self, labels):
        """PowSum normaliser + the five SigmoidAdaptiveDistillLoss, forward and
        gradient w.r.t. the student logits (loss gradient = 1.0,
        utils/blob.py:166-172)."""
        cfg = self.cfg
        kw = dict(gamma=cfg.distill_gamma, alpha=cfg.distill_alpha, beta=cfg.distill_beta,
                  num_classes=self.C, ignored_label=cfg.ignored_label,
                  scale=cfg.loss_scale * cfg.temperature * cfg.temperature)
        self.normalizer = K.pow_sum(self.t_prob, cfg.logits_power).reshape(1)
        levels = list(zip(self.cls_logits, self.t_prob, labels))
        self.losses = K.distill_loss_forward(levels, self.normalizer, **kw)
        K.distill_loss_backward(levels, self.normalizer, self.one, out=self.d_cls_logits, **kw)
        return self.losses

    def _distill_kw(self):
        cfg = self.cfg
        return dict(gamma=cfg.distill_gamma, alpha=cfg.distill_alpha, beta=cfg.distill_beta,
                    num_classes=self.C, ignored_label=cfg.ignored_label,
                    scale=cfg.loss_scale * cfg.temperature * cfg.temperature)

    def cls_losses(self, labels, fg_num):
        """Both classification losses of the student (SigmoidFocalLoss,
        retinanet_heads.py:282-297, and SigmoidAdaptiveDistillLoss, :331-348) and
        their summed gradient w.r.t. the logits in ONE pass (the reference:
        2 forward ops + 2 gradient ops + an autograd Sum per level)."""
        cfg = self.cfg
        self.normalizer = K.pow_sum(self.t_prob, cfg.logits_power).reshape(1)
        levels = list(zip(self.cls_logits, self.t_prob, labels))
        focal_kw = dict(gamma=cfg.focal_gamma, alpha=cfg.focal_alpha, num_classes=self.C,
                        scale=cfg.loss_scale)
        self.losses, self.focal_losses, _ = K.cls_losses_fused(
            levels, self.normalizer, fg_num, self._distill_kw(), focal_kw, out=self.d_cls_logits)
        return self.losses, self.focal_losses

    def bbox_losses_fwd_bwd(self, bbox_targets, fg_num):
        """SelectSmoothL1Loss per level (retinanet_heads.py:268-280) and its
        gradient w.r.t. the box predictions.  bbox_targets: [(Y [M,4], L [M,4])]."""
        cfg = self.cfg
        kw = dict(beta=cfg.bbox_reg_beta, scale=cfg.loss_scale * cfg.bbox_reg_weight)
        losses = []
        for pred, (Y, Lc), dst in zip(self.bbox_pred, bbox_targets, self.d_bbox_pred):
            losses.append(K.select_smooth_l1_forward(pred, Y, Lc, fg_num, **kw))
            K.select_smooth_l1_backward(pred, Y, Lc, fg_num, self.one[:1], out=dst, **kw)
        self.bbox_losses = torch.stack(losses)
        return self.d_bbox_pred

    # -- backward -------------------------------------------------------------------
    def backward(self, d_bbox_pred):
        """Backward of both subnets, depth by depth from the prediction layers
        down.  Per depth: the two weight gradients (each already one workgroup
        per CU) and ONE data-gradient launch for both towers.  For tower
        layers the data gradient carries the ReluGradient mask, which is the
        layer's own post-ReLU input.  Each gradient bucket is all-reduced as
        soon as its last weight gradient is enqueued."""
        cfg = self.cfg
        nl = cfg.num_convs
        dy = {"cls": self.d_cls_logits, "bbox": d_bbox_pred}
        # prediction layers (different widths: separate launches)
        for t in ("cls", "bbox"):
            name = self._layers(t)[-1]
            x_in = self.act[t][nl - 1]
            Cout = self.params[name + "_b"].numel()
            K.conv3x3_wgrad(x_in, dy[t], Cout, dW=self.grads[name + "_w"], db=self.grads[name + "_b"])
            dy[t] = K.conv3x3_forward(dy[t], self.packed[name][1], None, self.D, mask_by=x_in,
                                      out=self.dbuf[t][nl & 1], wino=self._use_wino(self.D))
        for li in range(nl - 1, -1, -1):
            probs = []
            for t in ("cls", "bbox"):
                name = self._layers(t)[li]
                x_in = self.act[t][li - 1] if li > 0 else self.fpn_in
                K.conv3x3_wgrad(x_in, dy[t], self.D, dW=self.grads[name + "_w"],
                                db=self.grads[name + "_b"])
                out = self.dbuf[t][li & 1] if li > 0 else self.d_fpn[t]
                probs.append(dict(xs=dy[t], packed=self.packed[name][1], bias=None, out=out,
                                  mask_by=x_in if li > 0 else None))
                dy[t] = out
            K.conv3x3_forward_multi(probs, self.D, wino=self._use_wino(self.D))
            if li == nl // 2:
                self._allreduce_async("late")
        self._allreduce_async("early")
        return self.d_fpn

    # -- data parallel ------------------------------------------------------------------
    def _allreduce_async(self, tower):
        if tower in self.grads.bucket:
            self.dp.issue(self.grads.bucket[tower])

    def wait_gradients(self):
        self.dp.wait()

    def broadcast_params(self, src=0):
        """Initial parameter sync (detectron/lib/utils/net.py:185-208)."""
        self.dp.broadcast([self.params.flat, self.moms.flat], src=src)

    # -- update -------------------------------------------------------------------------
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
        self.wait_gradients()
        for name, _, is_bias, _ in self.params.specs:
            K.momentum_sgd_update_(self.params[name], self.grads[name], self.moms[name], self.lr,
                                   self.momentum, self.weight_decay, is_bias)

    # -- one iteration --------------------------------------------------------------------
    def step(self, student_fpn, teacher_fpn, labels, d_bbox_pred=None, update=True,
             bbox_targets=None, fg_num=None):
        """One iteration.  With `bbox_targets` and `fg_num` the student's
        supervised losses are part of the step (the full reference graph);
        without them only the distillation loss drives the cls subnet and
        `d_bbox_pred` must supply the box-subnet gradient."""
        self.pack_student()
        self.forward_all(teacher_fpn, student_fpn)
        if bbox_targets is not None:
            self.cls_losses(labels, fg_num)
            d_bbox_pred = self.bbox_losses_fwd_bwd(bbox_targets, fg_num)
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
    kernels; the gradient of the logits is multiplied by LOSS_SCALE before it is rounded to
    fp16 (its elements are ~1e-6) and every result leaving the fp16 domain -- filter / bias
    gradients, the gradient w.r.t. the FPN levels -- is divided by it again."""

    LOSS_SCALE = 8192.0

    def _blk(self, ch):
        return [torch.empty((self.N, (ch + 7) // 8, h, w, 8), dtype=torch.float16, device=self.device)
                for (h, w) in self.shapes]

    def _alloc_buffers(self):
        cfg, lv, blk, D = self.cfg, self._lv, self._blk, self.D
        self.act = {t: [blk(D) for _ in range(cfg.num_convs)] for t in ("cls", "bbox")}
        self.cls_logits, self.bbox_pred = lv(self.A * self.C), lv(4 * self.A)
        self.d_cls_logits, self.d_bbox_pred = lv(self.A * self.C), lv(4 * self.A)
        self.dy_pred = {"cls": blk(self.A * self.C), "bbox": blk(4 * self.A)}
        self.dbuf = {"cls": [blk(D), blk(D)], "bbox": [blk(D), blk(D)]}
        self.d_fpn = {t: lv(D) for t in ("cls", "bbox")}
        self.t_buf = {"cls": [blk(D), blk(D)], "bbox": [blk(D), blk(D)]}
        self.in_blk = {"student": blk(D), "teacher": blk(D)}
        self.t_prob = lv(self.A * self.C)
        self.t_bbox = lv(4 * self.A) if self.teacher_bbox_tower else None
        self.packed = {}
        self.t_packed = None
        self.losses = None
        self.normalizer = None

    def _pack(self, w, want_fwd, want_dgrad):
        return K.f16_pack_filter(w, want_fwd, want_dgrad)

    def forward_all(self, teacher_fpn, student_fpn):
        if self.t_packed is None:
            self.pack_teacher()
        cfg, D = self.cfg, self.D
        for x, xb in zip(student_fpn, self.in_blk["student"]):
            K.f16_pack_activations(x, out=xb)
        for x, xb in zip(teacher_fpn, self.in_blk["teacher"]):
            K.f16_pack_activations(x, out=xb)
        self.fpn_in = self.in_blk["student"]
        tx = {"cls": self.in_blk["teacher"], "bbox": self.in_blk["teacher"]}
        sx = {"cls": self.fpn_in, "bbox": self.fpn_in}
        F = K.conv3x3_forward_f16_levels          # all five levels of a layer in one launch
        for i in range(cfg.num_convs):
            # the four tower layers of equal depth (teacher / student x cls / bbox) are independent
            # convolutions of one shape: ONE launch of 20 (level, filter) problems, as on the fp32 path
            probs = []
            for t in ("cls", "bbox"):
                name = self._layers(t)[i]
                if t == "cls" or self.teacher_bbox_tower:
                    out = self.t_buf[t][i & 1]
                    probs.append(dict(xs=tx[t], packed=self.t_packed[name], bias=self.teacher[name + "_b"],
                                      out=out))
                    tx[t] = out
                out = self.act[t][i]
                probs.append(dict(xs=sx[t], packed=self.packed[name][0], bias=self.params[name + "_b"],
                                  out=out))
                sx[t] = out
            K.conv3x3_forward_f16_multi(probs, D, D, relu=True)
        cp, bp = self._layers("cls")[-1], self._layers("bbox")[-1]
        AC, A4 = self.A * self.C, 4 * self.A
        F(tx["cls"], self.t_packed[cp], self.teacher[cp + "_b"], D, AC, self.t_prob, sigmoid=True,
          out_nchw_f32=True)
        F(sx["cls"], self.packed[cp][0], self.params[cp + "_b"], D, AC, self.cls_logits, out_nchw_f32=True)
        F(sx["bbox"], self.packed[bp][0], self.params[bp + "_b"], D, A4, self.bbox_pred, out_nchw_f32=True)
        if self.teacher_bbox_tower:
            F(tx["bbox"], self.t_packed[bp], self.teacher[bp + "_b"], D, A4, self.t_bbox, out_nchw_f32=True)
        return self.cls_logits, self.bbox_pred

    def backward(self, d_bbox_pred):
        cfg, D, S = self.cfg, self.D, self.LOSS_SCALE
        nl, nlev = cfg.num_convs, len(self.shapes)
        dy = {}
        for t, src in (("cls", self.d_cls_logits), ("bbox", d_bbox_pred)):
            for l in range(nlev):
                K.f16_pack_activations(src[l], scale=S, out=self.dy_pred[t][l])
            dy[t] = self.dy_pred[t]
        for t in ("cls", "bbox"):
            name = self._layers(t)[-1]
            x_in = self.act[t][nl - 1]
            Cout = self.params[name + "_b"].numel()
            K.conv3x3_wgrad_f16(x_in, dy[t], D, Cout, scale=1.0 / S, dW=self.grads[name + "_w"],
                                db=self.grads[name + "_b"])
            out = self.dbuf[t][nl & 1]
            K.conv3x3_forward_f16_levels(dy[t], self.packed[name][1], None, Cout, D, out, mask_bys=x_in)
            dy[t] = out
        for li in range(nl - 1, -1, -1):
            probs = []
            for t in ("cls", "bbox"):
                name = self._layers(t)[li]
                x_in = self.act[t][li - 1] if li > 0 else self.fpn_in
                K.conv3x3_wgrad_f16(x_in, dy[t], D, D, scale=1.0 / S, dW=self.grads[name + "_w"],
                                    db=self.grads[name + "_b"])
                out = self.dbuf[t][li & 1]
                probs.append(dict(xs=dy[t], packed=self.packed[name][1], bias=None, out=out,
                                  mask_by=x_in if li > 0 else None))
                dy[t] = out
            K.conv3x3_forward_f16_multi(probs, D, D)      # both towers' data gradients: one launch
            if li == nl // 2:
                self._allreduce_async("late")
        for t in ("cls", "bbox"):
            for l in range(nlev):
                K.f16_unpack_activations(dy[t][l], D, scale=1.0 / S, out=self.d_fpn[t][l])
        self._allreduce_async("early")
        return self.d_fpn
