"""ResNet-FPN backbones (torch, MIOpen convs) + the HIP subnet pipeline.

Structure follows detectron/lib/modeling/ResNet.py:85-130 (bottleneck stages
3-4-{6,23}-3, stride on the first 1x1 as with the MSRA weights, frozen BN as a
per-channel affine = AffineChannel) and FPN.py:116-250 for RetinaNet
(laterals on res3..res5, 3x3 output convs, P6 = conv3x3/2 on res5, P7 =
conv3x3/2 on relu(P6); levels P3..P7 at 256 channels).  Random weights.
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F


def _K():
    from ssad_amd import kernels
    return kernels


class _HipConv3x3Fn(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 convolution (+bias, optional fused ReLU) on this repo's
    kernels: Winograd forward and data gradient, direct-form weight gradient.  NCHW
    inside; channels-last callers pay one layout copy each way."""

    @staticmethod
    def forward(ctx, x, w, b, relu, cl):
        K = _K()
        xc = x.contiguous()
        need_dx = x.requires_grad
        wf, wd = K.conv_wino_pack_filter(w.detach().contiguous(), True, need_dx)
        y = K.conv3x3_forward([xc], wf, b.detach().contiguous(), w.shape[0], relu=relu, wino=True)[0]
        ctx.relu, ctx.cl, ctx.need_dx = relu, cl, need_dx
        ctx.wd = wd
        ctx.save_for_backward(xc, y if relu else None)
        ctx.cout, ctx.cin = w.shape[0], w.shape[1]
        return y.contiguous(memory_format=torch.channels_last) if cl else y

    @staticmethod
    def backward(ctx, dy):
        K = _K()
        xc, y = ctx.saved_tensors
        dz = dy.contiguous()
        if ctx.relu:
            dz = K.relu_grad(y, dz, out=dz if dz.data_ptr() != dy.data_ptr() else None)
        dx = None
        if ctx.need_dx:
            dx = K.conv3x3_forward([dz], ctx.wd, None, ctx.cin, wino=True)[0]
            if ctx.cl:
                dx = dx.contiguous(memory_format=torch.channels_last)
        dW, db = K.conv3x3_wgrad([xc], [dz], ctx.cout)
        return dx, dW, db, None, None


class _BiasActFn(torch.autograd.Function):
    """y = act(z + bias[c] (+ residual)) in ONE in-place pass over the conv output z
    (torch runs bias add, residual add and ReLU as three passes).  z must not be
    needed by its producer's backward (a convolution needs only input and weight)."""

    @staticmethod
    def forward(ctx, z, bias, residual, relu):
        K = _K()
        y = K.affine_channel_(z, bias.detach().contiguous(), residual=residual, relu=relu)
        ctx.mark_dirty(z)
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = _K()
        (y,) = ctx.saved_tensors
        dz = dy.contiguous()
        db = None
        if ctx.needs_input_grad[1]:
            # ReluGradient and the bias gradient's plane sums in one pass over dy
            dz, rs = K.relu_grad_rowsum(y if ctx.relu else None, dz)
            db = rs.sum(0)
        elif ctx.relu:
            dz = K.relu_grad(y, dz)
        return dz, db, (dz if ctx.has_res else None), None


class _Conv1x1Fn(torch.autograd.Function):
    """Stride-1 pointwise convolution without bias as three strided-batched GEMMs on the
    NCHW tensors (batch = image): Y[n] = W . X[n], dX[n] = W^T . dY[n],
    dW = sum_n dY[n] . X[n]^T.  MIOpen runs the same rocBLAS GEMMs for forward and data
    gradient but picks its own solution, and its weight gradient is an NHWC igemm that
    costs two layout transposes per call; going through torch's BLAS front end lets
    TunableOp's per-shape pick (harness/tunableop_gfx950.csv) apply to all three."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _mm1x1(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        N, M = dy.shape[0], dy.shape[1]
        Cc = x.shape[1]
        if ctx.needs_input_grad[0]:
            # explicit batched form: torch.matmul would fold 2-D x 3-D into one GEMM over
            # a transposed COPY of dy
            dx = torch.empty_like(x)
            torch.bmm(w.view(1, M, Cc).transpose(1, 2).expand(N, Cc, M), dy.view(N, M, -1),
                      out=dx.view(N, Cc, -1))
        if ctx.needs_input_grad[1]:
            dw = torch.bmm(dy.view(N, M, -1), x.view(N, Cc, -1).transpose(1, 2)).sum(0).view_as(w)
        return dx, dw


class _Conv1x1ShortcutFn(torch.autograd.Function):
    """c1 of a bottleneck together with the tap for the block's shortcut: returns
    (conv1x1(x, w), x).  Autograd would add the two gradients reaching x -- the shortcut's and
    c1's data gradient -- in a pass of its own (read two tensors, write a third); here c1's data
    gradient GEMM accumulates onto the shortcut's gradient in place (beta = 1)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _mm1x1(x, w), x.view_as(x)

    @staticmethod
    def backward(ctx, dz, dsc):
        x, w = ctx.saved_tensors
        N, M = dz.shape[0], dz.shape[1]
        Cc = x.shape[1]
        dz = dz.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = w.view(1, M, Cc).transpose(1, 2).expand(N, Cc, M)
            if dsc is None:
                dx = torch.empty_like(x)
                torch.bmm(wt, dz.view(N, M, -1), out=dx.view(N, Cc, -1))
            else:
                dx = dsc if dsc.is_contiguous() else dsc.contiguous()
                dx.view(N, Cc, -1).baddbmm_(wt, dz.view(N, M, -1))
        if ctx.needs_input_grad[1]:
            dw = torch.bmm(dz.view(N, M, -1), x.view(N, Cc, -1).transpose(1, 2)).sum(0).view_as(w)
        return dx, dw


class _Conv1x1TailFn(torch.autograd.Function):
    """The last layer of a bottleneck in one kernel: relu(conv1x1(x, w) + bias + shortcut)
    (ssad_conv1x1_bias_act).  At the early stages this layer is HBM bound and the library route
    pays its output twice (GEMM writes Z; the tail pass reads Z and the shortcut, writes Y)."""

    @staticmethod
    def forward(ctx, x, w, bias, res):
        y = _K().conv1x1_bias_act(x, w.detach(), bias.detach().contiguous(), res, relu=True)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = _K()
        x, w, y = ctx.saved_tensors
        N, Cc = x.shape[0], x.shape[1]
        M = w.shape[0]
        dz, rs = K.relu_grad_rowsum(y, dy.contiguous())
        db = rs.sum(0) if ctx.needs_input_grad[2] else None
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            torch.bmm(w.view(1, M, Cc).transpose(1, 2).expand(N, Cc, M), dz.view(N, M, -1),
                      out=dx.view(N, Cc, -1))
        if ctx.needs_input_grad[1]:
            dw = torch.bmm(dz.view(N, M, -1), x.view(N, Cc, -1).transpose(1, 2)).sum(0).view_as(w)
        return dx, dw, db, (dz if ctx.needs_input_grad[3] else None)


def conv1x1_tail(y, w, bias, sc):
    """relu(conv1x1(y, w) + bias + sc): fused kernel where it wins (64 / 128 input channels,
    i.e. res2 / res3), GEMM + fused tail pass elsewhere."""
    cin, cout = w.shape[1], w.shape[0]
    if (_FUSED_PW and cin in _FUSED_PW_CIN and cout % 128 == 0 and (y.shape[2] * y.shape[3]) % 4 == 0
            and y.is_contiguous() and sc.is_contiguous()):
        if y.requires_grad or w.requires_grad or bias.requires_grad or sc.requires_grad:
            return _Conv1x1TailFn.apply(y, w, bias, sc)
        return _K().conv1x1_bias_act(y, w.detach(), bias.detach().contiguous(), sc, relu=True)
    return bias_act(conv1x1(y, w), bias, residual=sc)


def _mm1x1(x, w):
    N, Cc, H, W = x.shape
    M = w.shape[0]
    y = x.new_empty((N, M, H, W))       # a fresh tensor, not a view: the tail pass writes it in place
    torch.bmm(w.detach().view(1, M, Cc).expand(N, M, Cc), x.detach().view(N, Cc, H * W),
              out=y.view(N, M, H * W))
    return y


def conv1x1(x, w):
    if _GEMM_1X1 and x.is_contiguous():
        if w.requires_grad or x.requires_grad:
            return _Conv1x1Fn.apply(x, w)
        return _mm1x1(x, w)
    return F.conv2d(x, w, None)


def conv3x3_s2_gemm(x, w, b):
    """3x3 / stride 2 / pad 1 convolution (FPN's P6 on res5: 2048 -> 256 at 20x28) as
    im2col + ONE GEMM with the batch folded into the columns (K = 9*Cin = 18 432);
    MIOpen's stride-2 Winograd runs this geometry at ~20 TFLOP/s."""
    N, Cc, H, W = x.shape
    M = w.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    cols = F.unfold(x, 3, padding=1, stride=2)                      # [N, 9*Cin, Ho*Wo]
    cols = cols.permute(1, 0, 2).reshape(Cc * 9, N * Ho * Wo)
    y = torch.mm(w.view(M, Cc * 9), cols).view(M, N, Ho, Wo).permute(1, 0, 2, 3)
    return (y + b.view(1, M, 1, 1)).contiguous()


def bias_act(z, bias, residual=None, relu=True):
    if not (z.requires_grad or bias.requires_grad or (residual is not None and residual.requires_grad)):
        return _K().affine_channel_(z, bias.detach().contiguous(), residual=residual, relu=relu)
    return _BiasActFn.apply(z, bias, residual, relu)


class HipConv3x3(nn.Module):
    def __init__(self, cin, cout, relu=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.relu = relu
        self._packed = None          # frozen (teacher) weights: pack once

    def forward(self, x):
        cl = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        if not (self.weight.requires_grad or x.requires_grad):
            K = _K()
            if self._packed is None:
                self._packed = K.conv_wino_pack_filter(self.weight.detach().contiguous(), True, False)[0]
            y = K.conv3x3_forward([x.contiguous()], self._packed, self.bias.detach(),
                                  self.weight.shape[0], relu=self.relu, wino=True)[0]
            return y.contiguous(memory_format=torch.channels_last) if cl else y
        return _HipConv3x3Fn.apply(x, self.weight, self.bias, self.relu, cl)


# Default: every stride-1 3x3 convolution of the bottlenecks (res2..res5) and the FPN output
# convs (55 forward + 16 backward layers per step) run on this repo's Winograd /
# wgrad kernels with bias and ReLU fused, and the harness stays NCHW.  Measured on one
# MI355X, bs 16: 143.9 ms/step against 150.7 ms for MIOpen-only channels-last
# (SSAD_HARNESS_HIP3X3=0), 155.0 ms MIOpen-only NCHW, 149.3 ms HIP 3x3 inside a
# channels-last harness (layout copies around every call).
_HIP3X3 = os.environ.get("SSAD_HARNESS_HIP3X3", "1") == "1"
_HIP3X3_MIN = int(os.environ.get("SSAD_HARNESS_HIP3X3_MIN", "64"))
# bias + residual + ReLU after the MIOpen / rocBLAS convolutions of a bottleneck as one
# fused AffineChannel pass of this repo (NCHW only) instead of three torch passes
_FUSE_TAIL = os.environ.get("SSAD_HARNESS_FUSE_TAIL", "1") == "1"
# stride-1 pointwise convolutions as torch strided-batched GEMMs (see _Conv1x1Fn)
_GEMM_1X1 = os.environ.get("SSAD_HARNESS_GEMM_1X1", "1") == "1"
# last layer of the res2 / res3 bottlenecks as one fused kernel (see _Conv1x1TailFn)
_FUSED_PW = os.environ.get("SSAD_HARNESS_FUSED_PW", "1") == "1"
_FUSED_PW_CIN = tuple(int(v) for v in os.environ.get("SSAD_HARNESS_FUSED_PW_CIN", "64,128").split(","))
# TunableOp: "1" = use the committed per-shape GEMM picks when the file matches this
# stack (its validator lines name torch / ROCm / rocBLAS / hipBLASLt / gfx arch; on a
# mismatch torch ignores it), "tune" = search and write SSAD_TUNABLEOP_OUT, "0" = off
_TUNABLEOP = os.environ.get("SSAD_HARNESS_TUNABLEOP", "1")
_TUNABLEOP_CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def setup_tunableop():
    import torch.cuda.tunable as T
    if _TUNABLEOP == "tune":
        T.enable(True)
        T.tuning_enable(True)
        T.set_max_tuning_duration(int(os.environ.get("SSAD_TUNABLEOP_MS", "60")))
        T.set_max_tuning_iterations(int(os.environ.get("SSAD_TUNABLEOP_ITERS", "30")))
        T.set_filename(os.environ.get("SSAD_TUNABLEOP_OUT", "gpurun_out/tunableop_gfx950.csv"))
    elif _TUNABLEOP == "1" and os.path.exists(_TUNABLEOP_CSV):
        T.enable(True)
        T.tuning_enable(False)
        T.set_filename(_TUNABLEOP_CSV)


def conv_frozen_bn(cin, cout, k, stride=1, padding=0):
    """Conv followed by a frozen-BN AffineChannel (y = x*s + b with constant s, b;
    detectron/lib/modeling/ResNet.py uses AffineChannel after every conv).  With
    s and b constant the affine folds exactly into the conv: W' = s*W, bias = b,
    so the harness runs one conv-with-bias instead of conv + two elementwise ops."""
    return nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=True)


class Bottleneck(nn.Module):
    """ResNet.py:223-283.  groups > 1 with stride_1x1=False is the ResNeXt block
    (RESNETS.NUM_GROUPS / STRIDE_1X1 of retinanet_X-101-64x4d-FPN_1x_teacher.yaml:19-24):
    the grouped 3x3 carries the stride and runs on MIOpen."""

    def __init__(self, cin, cmid, cout, stride, groups=1, stride_1x1=True):
        super().__init__()
        s1, s3 = (stride, 1) if stride_1x1 else (1, stride)
        self.c1 = conv_frozen_bn(cin, cmid, 1, stride=s1)
        self.hip2 = _HIP3X3 and cmid >= _HIP3X3_MIN and groups == 1 and s3 == 1
        if self.hip2:
            self.c2 = HipConv3x3(cmid, cmid, relu=True)
        else:
            self.c2 = nn.Conv2d(cmid, cmid, 3, stride=s3, padding=1, groups=groups, bias=True)
        self.c3 = conv_frozen_bn(cmid, cout, 1)
        self.proj = conv_frozen_bn(cin, cout, 1, stride=stride) if (cin != cout or stride != 1) else None
        self._w12 = self._b12 = None      # frozen [W3 | Wproj] and bias sum (see forward)

    def forward(self, x):
        if _FUSE_TAIL and x.is_contiguous():
            xs = x
            if self.proj is not None and self.proj.stride != (1, 1):
                # a strided pointwise convolution is the pointwise convolution of the
                # subsampled map; c1 and the projection share the one gather
                xs = x[:, :, ::self.proj.stride[0], ::self.proj.stride[1]].contiguous()
            if self.c1.stride != (1, 1):
                x = xs
            b3 = self.c3.bias
            frozen = not (x.requires_grad or self.c3.weight.requires_grad)
            if (_FUSED_PW and frozen and self.proj is not None and x is xs and self.hip2
                    and self.c3.weight.shape[1] + self.proj.weight.shape[1] == 128
                    and self.c3.weight.shape[0] % 128 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0):
                # res2's first block, frozen: last layer and projection shortcut as ONE product
                # [W3 | Wproj] . [y2 ; x] -- the projection's output is never written
                if self._w12 is None:
                    M = self.c3.weight.shape[0]
                    self._w12 = torch.cat([self.c3.weight.detach().view(M, -1),
                                           self.proj.weight.detach().view(M, -1)], 1).contiguous()
                    self._b12 = (self.c3.bias + self.proj.bias).detach().contiguous()
                y = bias_act(conv1x1(x, self.c1.weight), self.c1.bias)
                y = self.c2(y)
                return _K().conv1x1_bias_act2(y, x, self._w12, self._b12, relu=True)
            if _GEMM_1X1 and x.requires_grad and x is xs:
                # c1 and the shortcut tap in one node: their gradients meet inside c1's GEMM
                z1, xs = _Conv1x1ShortcutFn.apply(x, self.c1.weight)
            else:
                z1 = conv1x1(x, self.c1.weight)
            if self.proj is None:
                sc = xs
            else:
                # the projection's bias rides along with c3's in the block's last pass
                sc = conv1x1(xs, self.proj.weight)
                b3 = b3 + self.proj.bias
            # convolution without bias, then bias (+ residual) + ReLU in one pass
            y = bias_act(z1, self.c1.bias)
            y = self.c2(y) if self.hip2 else bias_act(
                F.conv2d(y, self.c2.weight, None, self.c2.stride, 1, 1, self.c2.groups), self.c2.bias)
            return conv1x1_tail(y, self.c3.weight, b3, sc)
        sc = x if self.proj is None else self.proj(x)
        y = F.relu(self.c1(x), inplace=True)
        y = self.c2(y) if self.hip2 else F.relu(self.c2(y), inplace=True)
        y = self.c3(y)
        return F.relu(y.add_(sc), inplace=True)


ARCHS = {   # name -> (block counts, groups, width per group, stride on the 1x1)
    "r50": ((3, 4, 6, 3), 1, 64, True),
    "r101": ((3, 4, 23, 3), 1, 64, True),
    "x101-64x4d": ((3, 4, 23, 3), 64, 4, False),
}


class ResNetFPN(nn.Module):
    def __init__(self, depth=50, fpn_dim=256):
        super().__init__()
        arch = depth if isinstance(depth, str) else "r%d" % depth
        blocks, groups, width, stride_1x1 = ARCHS[arch]
        self.stem = nn.Sequential(conv_frozen_bn(3, 64, 7, stride=2, padding=3),
                                  nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2, padding=1))
        stages, cin = [], 64
        for i, n in enumerate(blocks):
            cmid, cout = groups * width * 2 ** i, 256 * 2 ** i
            layers = []
            for j in range(n):
                layers.append(Bottleneck(cin, cmid, cout, 2 if (j == 0 and i > 0) else 1,
                                         groups=groups, stride_1x1=stride_1x1))
                cin = cout
            stages.append(nn.Sequential(*layers))
        self.res2, self.res3, self.res4, self.res5 = stages
        self.lat = nn.ModuleList([nn.Conv2d(c, fpn_dim, 1) for c in (2048, 1024, 512)])
        self.out = nn.ModuleList([HipConv3x3(fpn_dim, fpn_dim) if _HIP3X3 else
                                  nn.Conv2d(fpn_dim, fpn_dim, 3, padding=1) for _ in range(3)])
        self.p6 = nn.Conv2d(2048, fpn_dim, 3, stride=2, padding=1)
        self.p7 = nn.Conv2d(fpn_dim, fpn_dim, 3, stride=2, padding=1)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, HipConv3x3)):
                nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                # frozen-BN scales carry no statistics with random weights: damp
                # the residual branch (folded scale 0.25) so activations stay
                # O(1) through 16 / 33 blocks (a trained model's BN does this job)
                with torch.no_grad():
                    m.c3.weight.mul_(0.25)
        for m in list(self.lat) + list(self.out) + [self.p6, self.p7]:
            nn.init.xavier_uniform_(m.weight)
        # the stem and res2 are frozen in Detectron (TRAIN.FREEZE_CONV_BODY / FREEZE_AT = 2)
        for p in list(self.stem.parameters()) + list(self.res2.parameters()):
            p.requires_grad_(False)

    def forward(self, x):
        if _FUSE_TAIL and x.is_contiguous() and not x.requires_grad:
            # frozen stem: 7x7/2 convolution, then bias + ReLU + 3x3/2 pool in one pass
            z = F.conv2d(x, self.stem[0].weight, None, 2, 3)
            c1 = _K().max_pool3x3s2_bias_relu(z, self.stem[0].bias.detach(), relu=True)
        else:
            c1 = self.stem(x)
        c2 = self.res2(c1)
        c3 = self.res3(c2)
        c4 = self.res4(c3)
        c5 = self.res5(c4)
        if _FUSE_TAIL and c5.is_contiguous():
            # laterals as GEMMs; bias and the top-down sum in one pass
            t5 = bias_act(conv1x1(c5, self.lat[0].weight), self.lat[0].bias, relu=False)
            t4 = bias_act(conv1x1(c4, self.lat[1].weight), self.lat[1].bias, relu=False,
                          residual=F.interpolate(t5, scale_factor=2, mode="nearest"))
            t3 = bias_act(conv1x1(c3, self.lat[2].weight), self.lat[2].bias, relu=False,
                          residual=F.interpolate(t4, scale_factor=2, mode="nearest"))
            p6 = conv3x3_s2_gemm(c5, self.p6.weight, self.p6.bias)
        else:
            t5 = self.lat[0](c5)
            t4 = self.lat[1](c4) + F.interpolate(t5, scale_factor=2, mode="nearest")
            t3 = self.lat[2](c3) + F.interpolate(t4, scale_factor=2, mode="nearest")
            p6 = self.p6(c5)
        p5, p4, p3 = self.out[0](t5), self.out[1](t4), self.out[2](t3)
        p7 = self.p7(F.relu(p6))
        return [p3, p4, p5, p6, p7]      # finest first, matching synth.LEVEL_SHAPES_600


class FullDistillModel(object):
    """One distillation iteration of the whole detector on one GPU."""

    def __init__(self, heads, student_depth=50, teacher_depth=101, device="cuda",
                 process_group=None, world_size=1, lr=1e-5, momentum=0.9, weight_decay=1e-4,
                 backbone_f16=False):
        """backbone_f16: run both backbones under torch.autocast(float16) on MIOpen / rocBLAS
        (fp32 master weights; config 5's precision for the part of the model this repo does not
        own).  This repo's fp32 backbone kernels are then out of the picture: the plain
        nn.Conv2d route, channels-last."""
        global _HIP3X3, _FUSE_TAIL, _GEMM_1X1
        self.backbone_f16 = backbone_f16
        if backbone_f16:
            _HIP3X3 = _FUSE_TAIL = _GEMM_1X1 = False
        self.heads = heads
        self.timing = None
        self.pg, self.world = process_group, world_size
        # teacher_depth None / "none": plain RetinaNet training of the student (BASELINE config 2,
        # detectron/lib/modeling/model_builder.py:98-100,413 without the distillation wrapper)
        self.has_teacher = teacher_depth not in (None, "none")
        assert self.has_teacher == bool(getattr(heads, "distill", True)), \
            "the subnet pipeline and the full model must agree on distillation"
        with torch.random.fork_rng():
            torch.manual_seed(7)
            self.student = ResNetFPN(student_depth).to(device)
            self.teacher = ResNetFPN(teacher_depth).to(device).eval() if self.has_teacher else None
        if self.has_teacher:
            for p in self.teacher.parameters():
                p.requires_grad_(False)
        # harness tuning knobs (A/B via env): MIOpen solver search and NHWC layout
        if os.environ.get("SSAD_HARNESS_BENCHMARK", "0") == "1":
            torch.backends.cudnn.benchmark = True
        self.channels_last = os.environ.get("SSAD_HARNESS_NHWC", "0" if _HIP3X3 else "1") == "1"
        self._cast = (lambda: torch.autocast("cuda", dtype=torch.float16)) if backbone_f16 else \
            (lambda: torch.autocast("cuda", enabled=False))
        if self.channels_last:
            self.student = self.student.to(memory_format=torch.channels_last)
            if self.has_teacher:
                self.teacher = self.teacher.to(memory_format=torch.channels_last)
        setup_tunableop()
        self.two_streams = os.environ.get("SSAD_HARNESS_TWO_STREAMS", "1") == "1" and self.has_teacher
        self.side = torch.cuda.Stream() if self.two_streams else None
        self.trainable = [p for p in self.student.parameters() if p.requires_grad]
        self.opt = torch.optim.SGD(self.trainable, lr=lr, momentum=momentum,
                                   weight_decay=weight_decay)
        self.dist_on = self.pg is not None and (world_size > 1 or os.environ.get("SSAD_DP_FORCE") == "1")
        if self.dist_on:
            import torch.distributed as dist
            for p in self.student.parameters():
                dist.broadcast(p.data, src=0, group=self.pg)
            # flat gradient buckets in backward-completion order (FPN, res5, res4, res3):
            # few, large all-reduces, each started by the hook of its last gradient
            from ssad_amd.data_parallel import BucketedAllReduce, GradBuckets
            st = self.student
            fpn = list(st.lat.parameters()) + list(st.out.parameters()) + \
                list(st.p6.parameters()) + list(st.p7.parameters())
            groups = [[p for p in g if p.requires_grad] for g in
                      (fpn, st.res5.parameters(), st.res4.parameters(), st.res3.parameters(),
                       st.res2.parameters(), st.stem.parameters())]
            assert sum(len(g) for g in groups) == len(self.trainable)
            self.buckets = GradBuckets(groups, BucketedAllReduce(self.pg, world_size))

    def describe(self):
        """What runs where in the backbones (for bench.py's workload string)."""
        if self.backbone_f16:
            return "backbones = PyTorch harness under autocast(float16) on MIOpen / rocBLAS"
        return ("backbones = PyTorch harness (1x1 convs as rocBLAS / hipBLASLt GEMMs, MIOpen 7x7 stem, "
                "P7 and grouped convs; its stride-1 3x3 convs, bias/residual/ReLU tails and stem pool "
                "on this repo's kernels)")

    def _mark(self, name):
        if self._timing is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._timing.append((name, e))

    def step(self, images, labels, bbox_targets, fg_num):
        h = self.heads
        self._timing = [] if os.environ.get("SSAD_HARNESS_TIMING") == "1" else None
        self._mark("start")
        if self.channels_last:
            images = images.contiguous(memory_format=torch.channels_last)
        h.pack_student()
        if not self.has_teacher:
            t_fpn = None
            with self._cast():
                s_fpn = self.student(images)
            s_in = [t.detach().float().contiguous() for t in s_fpn]
            self._mark("student backbone fwd")
        elif self.two_streams:
            # teacher and student backbones are independent: on two streams the
            # last partial round of CUs of one network's kernel is filled by the
            # other network's next kernel (persistent / few-round launches leave
            # 5-20 % of a launch idle at these feature-map sizes)
            cur = torch.cuda.current_stream()
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                with torch.no_grad(), self._cast():
                    t_fpn = [t.float().contiguous() for t in self.teacher(images)]
            with self._cast():
                s_fpn = self.student(images)
            s_in = [t.detach().float().contiguous() for t in s_fpn]
            cur.wait_stream(self.side)
            for t in t_fpn:
                t.record_stream(cur)
            self._mark("teacher + student backbone fwd (two streams)")
        else:
            with torch.no_grad(), self._cast():
                t_fpn = [t.float().contiguous() for t in self.teacher(images)]
            self._mark("teacher backbone fwd")
            with self._cast():
                s_fpn = self.student(images)
            s_in = [t.detach().float().contiguous() for t in s_fpn]
            self._mark("student backbone fwd")
        h.forward_all(t_fpn, s_in)
        self._mark("subnets fwd (teacher+student)")
        h.cls_losses(labels, fg_num)
        d_bbox = h.bbox_losses_fwd_bwd(bbox_targets, fg_num)
        self._mark("losses")
        d_fpn = h.backward(d_bbox)
        self._mark("subnets bwd")
        # gradient w.r.t. each FPN level = cls-subnet part + bbox-subnet part,
        # in the memory format of the forward output (a mismatched format sends
        # MIOpen's backward to its slow non-packed fallback kernels)
        grads = []
        for a, b, f in zip(d_fpn["cls"], d_fpn["bbox"], s_fpn):
            g = (a + b).to(f.dtype)
            if self.channels_last:
                g = g.contiguous(memory_format=torch.channels_last)
            grads.append(g)
        if self.dist_on:
            self.buckets.begin()
        else:
            self.opt.zero_grad(set_to_none=True)
        torch.autograd.backward(s_fpn, grads)
        self._mark("student backbone bwd")
        if self.dist_on:
            self.buckets.finish()
        h.sgd_step()
        self.opt.step()
        self._mark("all-reduce + SGD")
        if self._timing is not None:
            torch.cuda.synchronize()
            print("harness timing: " + ", ".join(
                "%s %.1f ms" % (self._timing[i][0], self._timing[i - 1][1].elapsed_time(self._timing[i][1]))
                for i in range(1, len(self._timing))), file=sys.stderr)
        return h.losses
