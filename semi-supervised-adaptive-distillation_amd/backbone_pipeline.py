"""ResNet-FPN backbone (row f1) as a native program of this repo's kernels -- forward, backward
through res3..res5 + FPN, and the SGD update, with no torch operator in the step.

Structure of the network: detectron/lib/modeling/ResNet.py:85-130,221-283 (bottleneck stages
3-4-{6,23}-3, stride on the first 1x1 as with the MSRA weights, frozen BN = AffineChannel,
folded into the convolution that precedes it: W' = s W, bias = b; stem + res2 frozen,
TRAIN.FREEZE_CONV_BODY / FREEZE_AT = 2).  The body's convolutions have no bias of their own
(ResNet.py:270-283, no_bias=1) and AffineChannel's scale and bias are never trained
(caffe2/modules/detectron/affine_channel_op.cc: the gradient operator produces dX only), so
here the folded biases of res3..res5 are frozen values, not parameters, and the update of a
folded filter multiplies its gradient rows by s^2 (ssad_sgd_segment.row_scale) -- W' = s W
then follows the reference's update of W exactly: s (W - lr (s dW' + wd W)) = W' - lr (s^2 dW'
+ wd W').  Only the FPN's own convolutions (FPN.py:116-250) carry trained biases. and FPN.py:116-250 for RetinaNet (laterals on
res3..res5, top-down nearest upsampling + Sum, 3x3 output convs, P6 = conv3x3/2 on res5,
P7 = conv3x3/2 on relu(P6)).

Kernels:
  * every pointwise convolution (bottleneck 1x1s, projection shortcuts, laterals): the fp32-MFMA
    GEMM of csrc/kernels/gemm_conv.hip with bias / shortcut Sum / ReLU in its epilogue (forward),
    the ReluGradient mask and gradient accumulation in its epilogue (data gradient), and the
    split-reduction filter gradient;
  * every 3x3 / stride 1 convolution: the Winograd engine of the subnets (forward, data
    gradient with the fused ReluGradient mask, filter + bias gradient);
  * the stride-2 pointwise layers run on ssad_subsample's output (shared by c1 and the
    projection); P6 / P7 (3x3 / stride 2) run at their own size: implicit GEMM with split-K forward,
    flattened-batch GEMMs + gather-form col2im for the gradients (csrc/kernels/conv_strided.hip;
    SSAD_STRIDED_3X3=winograd: the stride-1 Winograd layer + subsampling of rounds 1-2, 4x the flops);
  * the 7x7 / stride 2 stem: batched im2col + the same GEMM, then bias + ReLU + 3x3/2 max pool
    in one pass (ssad_max_pool3x3s2_bias_relu).

All activations, gradients and packed filters are allocated once; the step is
Program.run() calls (ssad_program_run) -- a few segments when data parallel, so that a stage's
gradient bucket is all-reduced while the earlier stages are still in backward.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import kernels as K
from . import program as PR
from .data_parallel import BucketedAllReduce

# pixels (all images) a pointwise launch needs before the split-operand GEMM pays (see NativeResNetFPN._gemm)
GEMM_SPLIT_MIN_PIXELS = 8192

ARCHS = {"r50": (3, 4, 6, 3), "r101": (3, 4, 23, 3), "x101-64x4d": (3, 4, 23, 3)}
# ResNeXt (ResNet.py:247-258): name -> (groups, width per group); the stride sits on the 3x3
# (RESNETS.STRIDE_1X1 = False).  Forward only: it is the frozen teacher of BASELINE config 5.
GROUPED = {"x101-64x4d": (64, 4)}


class _Layer(object):
    __slots__ = ("name", "k", "cin", "cout", "stride", "train", "group", "affine", "w", "b", "gw", "gb", "wt",
                 "pf", "pd", "s2", "f24")

    def __init__(self, name, k, cin, cout, stride, train, group=1, affine=False):
        self.name, self.k, self.cin, self.cout, self.stride, self.train = name, k, cin, cout, stride, train
        self.group = group
        self.affine = affine       # followed by a frozen AffineChannel (folded): the bias is not a parameter
        self.w = self.b = self.gw = self.gb = self.wt = self.pf = self.pd = self.s2 = None
        self.f24 = False           # 3x3 forward on the F(2x4, 3x3) engine (frozen networks, >= 128 outputs)

    @property
    def wcin(self):
        """Input channels one filter sees ([cout][cin / group][k][k], conv_op_impl.h:43-47)."""
        return self.cin // self.group


class NativeResNetFPN(object):
    def __init__(self, arch="r50", N=2, image_hw=(640, 896), device="cuda", train=True, src=None,
                 fpn_dim=256, lr=1e-5, momentum=0.9, weight_decay=1e-4, process_group=None, world_size=1,
                 affine_scales=None, skip_flag=None, overlap_wgrad=None):
        if arch not in ARCHS:
            raise K.KernelError("native backbone: architectures %s" % sorted(ARCHS))
        if arch in GROUPED and train:
            raise K.KernelError("native backbone: %s is forward only (the frozen teacher); its grouped 3x3 has no "
                                "gradient kernels" % arch)
        self.arch, self.N, self.hw, self.device, self.train = arch, N, tuple(image_hw), device, train
        self.D = fpn_dim
        self.momentum, self.weight_decay = momentum, weight_decay
        self.dp = BucketedAllReduce(process_group, world_size)
        self.timing = None
        H, W = self.hw
        if H % 128 or W % 128:
            raise K.KernelError("native backbone: image sides must be multiples of 128 (16-pixel-multiple "
                                "maps down to res5; 640x896 and 512x768 are)")
        self._layers = OrderedDict()
        self._bufs = []
        self._define_layers()
        self._alloc_params(src, affine_scales)
        self.lr = torch.full((1,), lr, dtype=torch.float32, device=device)
        # device int (or None): when non-zero at execution time the update is dropped (a
        # mixed-precision step whose gradients overflowed, head_pipeline.DistillHeadsF16)
        self.skip_flag = skip_flag
        self._overlap_wgrad = overlap_wgrad
        self._build()

    # -- network definition -------------------------------------------------------------
    def _define_layers(self):
        L = self._layers

        def add(name, k, cin, cout, stride=1, train=True, group=1):
            L[name] = _Layer(name, k, cin, cout, stride, train and self.train, group,
                             affine=name.startswith(("stem", "res")))
        groups, width = GROUPED.get(self.arch, (1, 64))
        add("stem.0", 7, 3, 64, 2, train=False)
        cin = 64
        self.blocks = []                       # (stage, j, cin, cmid, cout, stride, has_proj, trainable)
        for si, nblk in enumerate(ARCHS[self.arch]):
            stage = si + 2
            cmid, cout = groups * width * 2 ** si, 256 * 2 ** si
            tr = stage > 2                     # FREEZE_AT = 2: the stem and res2 stay frozen
            for j in range(nblk):
                stride = 2 if (j == 0 and si > 0) else 1
                pre = "res%d.%d" % (stage, j)
                s1, s3 = (stride, 1) if groups == 1 else (1, stride)        # STRIDE_1X1
                add(pre + ".c1", 1, cin, cmid, s1, tr)
                add(pre + ".c2", 3, cmid, cmid, s3, tr, groups)
                add(pre + ".c3", 1, cmid, cout, 1, tr)
                proj = cin != cout or stride != 1
                if proj:
                    add(pre + ".proj", 1, cin, cout, stride, tr)
                self.blocks.append((stage, j, cin, cmid, cout, stride, proj, tr and self.train))
                cin = cout
        for i, c in enumerate((2048, 1024, 512)):
            add("lat.%d" % i, 1, c, self.D)
        for i in range(3):
            add("out.%d" % i, 3, self.D, self.D)
        add("p6", 3, 2048, self.D, 2)
        add("p7", 3, self.D, self.D, 2)

    def _bucket_order(self):
        """Trainable layers in the order the backward pass finishes them, grouped into the
        all-reduce buckets FPN, res5, res4, res3 ("backbone 4-6 buckets")."""
        groups = OrderedDict([("fpn", []), ("res5", []), ("res4", []), ("res3", [])])
        for name in ("p7", "p6", "out.0", "out.1", "out.2", "lat.0", "lat.1", "lat.2"):
            groups["fpn"].append(name)
        for stage in (5, 4, 3):
            n = ARCHS[self.arch][stage - 2]
            for j in range(n - 1, -1, -1):
                for part in ("c3", "proj", "c2", "c1"):
                    name = "res%d.%d.%s" % (stage, j, part)
                    if name in self._layers:
                        groups["res%d" % stage].append(name)
        return groups

    # Folded AffineChannel scale of the random initialisation: the last layer of every bottleneck
    # carries s = 0.25 (frozen-BN scales have no statistics with random weights; the damped residual
    # branch keeps activations O(1) through 16 / 33 blocks, the job a trained model's BN does).
    INIT_C3_SCALE = 0.25

    def _default_init(self):
        """Random weights of the network's shapes (there are no checkpoints on the box): He-normal
        filters (fan in) with zero bias for the body, Xavier-uniform for the FPN's own layers
        (FPN.py:116-250 uses XavierFill there).  Returns ({name.weight / name.bias: tensor},
        {name: folded affine scale})."""
        gen = torch.Generator().manual_seed(7)
        sd, scales = {}, {}
        for l in self._layers.values():
            shape = (l.cout, l.wcin, l.k, l.k)
            fan_in = l.wcin * l.k * l.k
            if l.affine:
                w = torch.randn(shape, generator=gen) * float(np.sqrt(2.0 / fan_in))
                if l.name.endswith(".c3"):
                    w *= self.INIT_C3_SCALE
                    scales[l.name] = self.INIT_C3_SCALE
            else:
                bound = float(np.sqrt(6.0 / (fan_in + l.cout * l.k * l.k)))
                w = (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound
            sd[l.name + ".weight"] = w
            sd[l.name + ".bias"] = torch.zeros(l.cout)
        return sd, scales

    def _alloc_params(self, src, affine_scales=None):
        """src: None (random initialisation), a {name.weight / name.bias: tensor} dict or a module
        whose state_dict() has those names (filters with the AffineChannel scale folded in);
        affine_scales: {layer name: s (float or [cout] tensor)} of the folded scales (default 1)."""
        dev = self.device
        L = self._layers
        scales = dict(affine_scales or {})
        # weights from outside (a checkpoint, utils/net.py): every trainable folded filter gets a scale slot, so
        # that load_from() can install other AffineChannel scales later without rebuilding the program.  The
        # random initialisation keeps slots only where its scale is not 1 (bench.py's path, unchanged).
        all_slots = src is not None or affine_scales is not None
        if src is None:
            sd, init_scales = self._default_init()
            for k, v in init_scales.items():
                scales.setdefault(k, v)
        else:
            sd = src if isinstance(src, dict) else src.state_dict()
            sd = {k: v.detach() for k, v in sd.items()}
            if any((l.name + ".weight") not in sd for l in L.values()):
                # a weights file that holds only part of the network (the reference's standard TRAIN.WEIGHTS
                # is an ImageNet body without fpn_* / retnet_* blobs): what it lacks keeps the initialisation
                init_sd, init_scales = self._default_init()
                for l in L.values():
                    if (l.name + ".weight") not in sd:
                        sd[l.name + ".weight"], sd[l.name + ".bias"] = init_sd[l.name + ".weight"], init_sd[l.name + ".bias"]
                        if l.name in init_scales:
                            scales.setdefault(l.name, init_scales[l.name])
        # the folded scales as given (utils/net.py writes them back as <conv>_bn_s when saving)
        self.affine_scale_values = {k: torch.as_tensor(v, dtype=torch.float32).reshape(-1).cpu().clone()
                                    for k, v in scales.items() if k in L and L[k].affine}
        groups = self._bucket_order() if self.train else OrderedDict()
        order = [L[n] for g in groups.values() for n in g]
        assert all(l.train for l in order) and (not self.train or len(order) == sum(l.train for l in L.values()))

        # frozen values: filters + biases of frozen layers, and the folded affine biases of the
        # trainable body layers
        def fsize(l):
            return (0 if l.train else l.cout * l.wcin * l.k * l.k) + (l.cout if (not l.train or l.affine) else 0)

        f32 = dict(dtype=torch.float32, device=dev)
        self.frozen_flat = torch.empty(sum(fsize(l) for l in L.values()), **f32)
        off = 0
        for l in L.values():
            if not l.train:
                nw = l.cout * l.wcin * l.k * l.k
                l.w = self.frozen_flat[off:off + nw].view(l.cout, l.wcin, l.k, l.k)
                off += nw
            if not l.train or l.affine:
                l.b = self.frozen_flat[off:off + l.cout]
                off += l.cout
        n_train = sum(l.cout * l.cin * l.k * l.k + (0 if l.affine else l.cout) for l in order)
        self.params_flat = torch.empty(n_train, **f32)
        self.grads_flat = torch.zeros(n_train, **f32)
        self.moms_flat = torch.zeros(n_train, **f32)
        self.bucket, self.segments = OrderedDict(), []
        off = 0
        for gname, names in groups.items():
            start = off
            for n in names:
                l = L[n]
                nw = l.cout * l.cin * l.k * l.k
                l.w = self.params_flat[off:off + nw].view(l.cout, l.cin, l.k, l.k)
                l.gw = self.grads_flat[off:off + nw].view(l.cout, l.cin, l.k, l.k)
                sc = scales.get(n)
                if sc is None and all_slots and l.affine:
                    sc = 1.0
                if sc is not None and l.affine:
                    sc = torch.as_tensor(sc, dtype=torch.float32).reshape(-1).to(dev)
                    l.s2 = (sc * sc).expand(l.cout).contiguous() if sc.numel() == 1 else (sc * sc).contiguous()
                self.segments.append((off, nw, 0, l.cin * l.k * l.k, l.s2))
                off += nw
                if not l.affine:
                    l.b = self.params_flat[off:off + l.cout]
                    l.gb = self.grads_flat[off:off + l.cout]
                    self.segments.append((off, l.cout, 1, 0, None))
                    off += l.cout
            self.bucket[gname] = self.grads_flat[start:off]
        for l in L.values():
            l.w.copy_(sd[l.name + ".weight"].to(device=dev, dtype=torch.float32))
            l.b.copy_(sd[l.name + ".bias"].to(device=dev, dtype=torch.float32))

    def load_from(self, src, affine_scales=None, strict=True):
        """Copy parameters from a {name.weight / name.bias: tensor} dict or a module with such a
        state_dict() (filters with the AffineChannel scale folded in).  strict=False: a layer the
        dict does not hold keeps its current values (detectron/lib/utils/net.py:96-99 logs
        "<name> not found" and leaves the initialised blob alone).  affine_scales: {layer: s} of
        the folded scales when they differ from the ones the network was built with -- the update of
        a trainable folded filter multiplies its gradient rows by s^2 (see the module docstring), so
        the network needs a scale slot for that layer (built with src= / affine_scales=)."""
        sd = src if isinstance(src, dict) else src.state_dict()
        for l in self._layers.values():
            if not strict and (l.name + ".weight") not in sd:
                continue
            l.w.copy_(sd[l.name + ".weight"].detach().to(device=self.device, dtype=torch.float32))
            l.b.copy_(sd[l.name + ".bias"].detach().to(device=self.device, dtype=torch.float32))
        if affine_scales is not None:
            for name, sc in affine_scales.items():
                l = self._layers.get(name)
                if l is None or not l.affine:
                    raise K.KernelError("load_from: %r is not a layer followed by an AffineChannel" % (name,))
                sc = torch.as_tensor(sc, dtype=torch.float32).reshape(-1)
                if sc.numel() == 1:
                    sc = sc.expand(l.cout)
                if sc.numel() != l.cout:
                    raise K.KernelError("load_from: %d scales for the %d channels of %s" % (sc.numel(), l.cout, name))
                self.affine_scale_values[name] = sc.cpu().clone()
                if not l.train:
                    continue
                if l.s2 is None:
                    if bool((sc == 1).all()):
                        continue
                    raise K.KernelError("load_from: %s has no scale slot (the network was built from the random "
                                        "initialisation); build it with src= / affine_scales=" % name)
                l.s2.copy_((sc * sc).to(self.device))
        self._packed_frozen = False

    # -- small emit helpers -------------------------------------------------------------------
    def _t(self, *shape):
        t = torch.empty(shape, dtype=torch.float32, device=self.device)
        self._bufs.append(t)
        return t

    def _like(self, t):
        u = torch.empty_like(t)
        self._bufs.append(u)
        return u

    # -- |max| words of the split-operand engines ------------------------------------------------------------
    # One table of words per network, zeroed by ONE fill at the start of the forward pass.  A tensor is measured once,
    # on the main stream, in front of its first split-engine consumer (SPLIT_ABSMAX / SPLIT_ABSMAX_LEVELS), and every
    # later consumer -- the filter gradient on an auxiliary stream in particular -- is handed the word: no call measures
    # or zeroes anything itself.  Tensors are identified by address: every activation and gradient buffer of a
    # program is allocated once (_t / _like) and written once per step before its consumers.
    AMAX_WORDS = 4096

    def _measure(self, P, tensors, channels):
        """Slot of the first of len(tensors) consecutive |max| words of these tensors (measured together, once)."""
        key = tuple(t.data_ptr() for t in tensors)
        if key in self._amax_groups:
            return self._amax_groups[key]
        base = self._amax_next
        self._amax_next += len(tensors)
        if self._amax_next > self.AMAX_WORDS:
            raise K.KernelError("|max| table full")
        addr = self.amax.data_ptr() + 4 * base
        if len(tensors) == 1:
            t = tensors[0]
            P.add(PR.SPLIT_ABSMAX, 73, p=(t, addr), l=(t.numel(),), work=4.0 * t.numel(), keep=[t])
        else:
            arr = (K.ConvLevel * len(tensors))()
            for i, t in enumerate(tensors):
                arr[i] = K.ConvLevel(t.data_ptr(), 0, 0, t.shape[0], t.shape[2], t.shape[3], 0, 0)
            P.add(PR.SPLIT_ABSMAX_LEVELS, 73, i=(len(tensors), channels, 0), p=(arr, addr),
                  work=4.0 * sum(t.numel() for t in tensors), keep=list(tensors))
        self._amax_groups[key] = base
        for k, t in enumerate(tensors):
            self._amax_groups[(t.data_ptr(),)] = base + k          # (the freshest measurement serves later readers)
        return base

    def _produces(self, *tensors):
        """Slots for tensors whose PRODUCER (a split-engine kernel's epilogue) folds their |max| in: nobody has to
        measure them.  The forward pass (every activation is written exactly once there) and, in the backward pass, the
        three gradients inside a bottleneck block (fold=True at their call sites); the other gradient buffers are also
        summed in place by element-wise kernels, which would leave a folded word stale."""
        base = self._amax_next
        self._amax_next += len(tensors)
        if self._amax_next > self.AMAX_WORDS:
            raise K.KernelError("|max| table full")
        self._amax_groups[tuple(t.data_ptr() for t in tensors)] = base
        for k, t in enumerate(tensors):
            self._amax_groups[(t.data_ptr(),)] = base + k
        return base

    def _amax_addr(self, slot):
        return self.amax.data_ptr() + 4 * slot

    def poison(self, value=float("nan")):
        """Fill every activation / gradient / scratch buffer (debugging aid: anything the step
        reads before it writes shows up as NaN in the results)."""
        for t in self._bufs:
            t.fill_(value)
        for w in self.wss.values():
            w.view(torch.float32)[: w.numel() // 4].fill_(value)
        self._packed_frozen = False

    def _gemm(self, P, a, lda, x, y, Kc, M, bias=None, res=None, mask=None, relu=False, acc=False, klass=50, final=True,
              fold=False):
        """final=False: y is modified in place afterwards (the top-down sum of the FPN): its |max| is not folded in.
        fold=True: a backward-pass output that is written once and read only by split-engine calls (the gradients inside
        a bottleneck block): folded like a forward activation."""
        d = K.gemm_conv_desc(a, lda, x, y, Kc, M, bias, res, mask, relu, acc)
        px = x.numel() // Kc
        # SSAD_SPLIT_CONV bit 128: the compute-bound pointwise layers (K, M >= 256: res4, res5, the laterals) on the
        # split-operand GEMM that splits x on the fly (gemm_split.hip, gemm_fly_kernel).  (With a split PASS over x in
        # front of the GEMM the step got 1.2-1.5 ms slower, profiles/r06_experiments.md section 3.)  Thresholds from
        # same-box A/B pairs: K, M >= 128 costs the step +1.6 ms, >= 64 +3.2 ms (those layers are HBM-bound); at batch 2
        # (config 2: < 8192 pixels per launch from res4 up) the call's three extra small launches cost more than the
        # GEMM saves (+1.0 ms on a 12 ms step).
        if self._gemm_split and Kc >= 256 and M >= 256 and px >= GEMM_SPLIT_MIN_PIXELS and klass == 50:
            nb = K.lib().ssad_conv1x1_gemm_split_workspace_bytes(C.byref(d))
            if nb:
                self._split_need = max(self._split_need, nb)
                xa = self._amax_addr(self._measure(P, [x], Kc))
                ya = self._amax_addr(self._produces(y)) if (self._fwd or fold) and final and not acc else None
                idx = P.add(PR.GEMM_CONV_SPLIT, 71, p=(d, None, self._packed_a(a, lda, Kc, M), xa, ya), l=(nb,),
                            work=2.0 * px * Kc * M, keep=[t for t in (a, x, y, bias, res, mask) if t is not None])
                self._gemm_split_ops.append(idx)
                return
        P.add(PR.GEMM_CONV, klass, p=(d,), work=2.0 * px * Kc * M,
              keep=[t for t in (a, x, y, bias, res, mask) if t is not None])

    def _packed_a(self, a, lda, Kc, M):
        """The split copy of a pointwise filter operand a[Kc][lda] (made by the program's one GEMM_SPLIT_PACK op: per
        step for a trained layer, once for a frozen one)."""
        key = (a.data_ptr(), lda, Kc, M)
        if key not in self._gemm_packs:
            dst = self._t(K.lib().ssad_gemm_split_filter_floats(Kc, M))
            train = self._a_train[a.data_ptr()]
            self._gemm_packs[key] = dst
            (self._gpack_train if train else self._gpack_frozen).append((a, lda, Kc, M, dst))
        return self._gemm_packs[key]

    def _conv3(self, P, probs, Cout, Cin, flags, klass=48, f24=False, fold=False):
        """probs: [(x, y, mask or None, packed, bias or None)]: independent 3x3 convolutions of one
        (Cout, Cin) in one launch; f24: truthy = on the F(2x4, 3x3) engine (packs from ssad_conv_wino24_pack_filters),
        3 = on the split-operand engine (conv3x3_split.hip; its workspace is bound when the program is finished)."""
        if f24 == 3:
            arr = (K.ConvLevel * len(probs))()
            for i, (x, y, mask, packed, bias) in enumerate(probs):
                arr[i] = K.ConvLevel(x.data_ptr(), y.data_ptr(), mask.data_ptr() if mask is not None else 0,
                                     x.shape[0], x.shape[2], x.shape[3], packed.data_ptr(),
                                     bias.data_ptr() if bias is not None else 0)
            nb = K.lib().ssad_conv3x3_split_workspace_bytes(arr, len(probs), Cin)
            self._split_need = max(getattr(self, "_split_need", 0), nb)
            px = sum(p[0].shape[0] * p[0].shape[2] * p[0].shape[3] for p in probs)
            xa = self._amax_addr(self._measure(P, [p[0] for p in probs], Cin))
            ya = self._amax_addr(self._produces(*[p[1] for p in probs])) if (self._fwd or fold) else None
            idx = P.add(PR.CONV3X3, 66 if self.train else 67, i=(len(probs), Cout, Cin, flags, 3), l=(nb,),
                        p=(arr, None, None, None, xa, ya), work=2.0 * 9 * Cout * Cin * px,
                        keep=[t for p in probs for t in p if t is not None])
            self._split_ops.append(idx)
            return
        if f24:
            klass = 46 if self.train else 47
        arr = (K.ConvLevel * len(probs))()
        for i, (x, y, mask, packed, bias) in enumerate(probs):
            arr[i] = K.ConvLevel(x.data_ptr(), y.data_ptr(), mask.data_ptr() if mask is not None else 0,
                                 x.shape[0], x.shape[2], x.shape[3], packed.data_ptr(),
                                 bias.data_ptr() if bias is not None else 0)
        px = sum(p[0].shape[0] * p[0].shape[2] * p[0].shape[3] for p in probs)
        P.add(PR.CONV3X3, klass, i=(len(probs), Cout, Cin, flags, 2 if f24 else 1), p=(arr, None, None),
              work=2.0 * 9 * Cout * Cin * px, keep=[t for p in probs for t in p if t is not None])

    def _wgrad3(self, P, x, dy, layer):
        arr = (K.ConvLevel * 1)()
        arr[0] = K.ConvLevel(x.data_ptr(), 0, dy.data_ptr(), x.shape[0], x.shape[2], x.shape[3], 0, 0)
        # SSAD_SPLIT_CONV bit 64 (default): the >= 256-wide 3x3 filter gradients on the split-operand engine (the
        # 128-wide res3 layers are level with the F(3x3, 2x2) engine there: 2 blocks of dW, 128 slabs to reduce)
        split = (int(os.environ.get("SSAD_SPLIT_CONV", "511")) & 64) != 0 and layer.cout >= 256 and layer.cin >= 256
        if split and not K.lib().ssad_conv3x3_wgrad_split_workspace_bytes(arr, 1, layer.cout, layer.cin):
            split = False           # (a tensor of 2 GiB or more: the exact engine)
        size_fn = K.lib().ssad_conv3x3_wgrad_split_workspace_bytes if split else K.lib().ssad_conv3x3_wgrad_workspace_bytes
        nb = size_fn(arr, 1, layer.cout, layer.cin)
        self._ws_need = max(self._ws_need, nb)
        xa = da = None
        if split:            # (measured on the main stream, before the fork)
            xa = self._amax_addr(self._measure(P, [x], layer.cin))
            da = self._amax_addr(self._measure(P, [dy], layer.cout))
        self._aux(P)
        idx = P.add(PR.CONV3X3_WGRAD, 70 if split else 49, i=(1, layer.cout, layer.cin, 0, 1 if split else 0), l=(nb,),
                    p=(arr, layer.gw, layer.gb, None, xa, da),
                    work=2.0 * 9 * layer.cout * layer.cin * x.shape[0] * x.shape[2] * x.shape[3], keep=[x, dy],
                    stream=self._wstream)
        self._ws_ops.append((idx, 3, self._wstream))

    def _wgrad1(self, P, x, dy, layer):
        N, Cc = x.shape[0], x.shape[1]
        pix = x.shape[2] * x.shape[3]
        # SSAD_SPLIT_CONV bit 256: the compute-bound pointwise filter gradients (C, M >= 256, the launch's pixels as for
        # the forward GEMM) on the split-operand engine (gemm_split.hip, wpoint_split_kernel)
        split = ((int(os.environ.get("SSAD_SPLIT_CONV", "511")) & 256) != 0 and Cc >= 256 and layer.cout >= 256
                 and N * pix >= GEMM_SPLIT_MIN_PIXELS and pix % 8 == 0)
        if split and not K.lib().ssad_conv1x1_wgrad_split_workspace_bytes(N, Cc, pix, layer.cout):
            split = False           # (a tensor of 2 GiB or more: the exact engine)
        size_fn = K.lib().ssad_conv1x1_wgrad_split_workspace_bytes if split else K.lib().ssad_conv1x1_wgrad_workspace_bytes
        nb = size_fn(N, Cc, pix, layer.cout)
        self._ws_need = max(self._ws_need, nb)
        xa = da = None
        if split:
            xa = self._amax_addr(self._measure(P, [x], Cc))
            da = self._amax_addr(self._measure(P, [dy], layer.cout))
        self._aux(P)
        idx = P.add(PR.CONV1X1_WGRAD, 72 if split else 52, i=(N, Cc, pix, layer.cout, 0, 1 if split else 0), l=(nb,),
                    p=(x, dy, layer.gw, None, xa, da), work=2.0 * N * pix * Cc * layer.cout, keep=[x, dy],
                    stream=self._wstream)
        self._ws_ops.append((idx, 3, self._wstream))

    def _bias_grad(self, P, dz, layer, rowsum=None):
        """db[c] = sum over n, pixels of dz; from the [N][C] plane sums when a ReluGradient pass
        already produced them."""
        self._aux(P)
        if rowsum is not None:
            P.add(PR.CHANNEL_SUM, 51, i=(rowsum.shape[0], rowsum.shape[1], 1, 0), p=(rowsum, layer.gb),
                  work=4.0 * rowsum.numel(), stream=self._wstream)
        else:
            # plane sums with one workgroup per (image, channel) -- N x C workgroups instead of the C of
            # ssad_channel_sum -- then the sum over the images of the tiny [N][C] table
            N, Cc = dz.shape[0], dz.shape[1]
            rows = self._t(N, Cc)
            P.add(PR.RELU_GRAD_ROWSUM, 51, i=(N, Cc, dz.shape[2] * dz.shape[3]), p=(None, dz, None, rows),
                  work=4.0 * dz.numel(), stream=self._wstream, keep=[dz, rows])
            P.add(PR.CHANNEL_SUM, 51, i=(N, Cc, 1, 0), p=(rows, layer.gb), work=4.0 * rows.numel(),
                  stream=self._wstream)

    def _conv_s2(self, P, x, y, layer):
        """y = conv3x3 / stride 2 / pad 1 (x) + bias at the layer's own size (FPN.py:193-224)."""
        N, Cc, H, W = x.shape
        oh, ow = y.shape[2], y.shape[3]
        d = K.gemm_conv_desc(layer.wt, layer.cout, x, y, Cc * 9, layer.cout, bias=layer.b)
        d.P = oh * ow
        nb = K.lib().ssad_conv_implicit_gemm_workspace_bytes(N, layer.cout, Cc, H, W, 3, 2, 1)
        ws = self._t(max(nb // 4, 4))
        P.add(PR.CONV_IMPLICIT_WS, 64, i=(Cc, H, W, 3, 2, 1), p=(d, ws), l=(nb,),
              work=2.0 * 9 * Cc * layer.cout * N * oh * ow, keep=[layer.wt, x, y, layer.b])

    def _wgrad_s2(self, P, x, dy, layer):
        N, Cc, H, W = x.shape
        nb = K.lib().ssad_conv_kxk_wgrad_workspace_bytes(N, Cc, H, W, layer.cout, 3, 2, 1)
        assert nb > 0
        self._ws_need = max(self._ws_need, nb)
        self._aux(P)
        idx = P.add(PR.CONV_KXK_WGRAD, 65, i=(N, Cc, H, W, layer.cout, 3, 2, 1), l=(nb, 0),
                    p=(x, dy, layer.gw, None), keep=[x, dy],
                    work=2.0 * 9 * Cc * layer.cout * N * dy.shape[2] * dy.shape[3], stream=self._wstream)
        self._ws_ops.append((idx, 3, self._wstream))
        self._bias_grad(P, dy, layer)

    def _dgrad_s2(self, P, layer, dy, dx, mask=None):
        N, Cc, H, W = dx.shape
        nb = K.lib().ssad_conv_kxk_dgrad_workspace_bytes(N, Cc, H, W, layer.cout, 3, 2, 1)
        assert nb > 0
        if self._dgrad_ws is None or self._dgrad_ws.numel() * 4 < nb:
            self._dgrad_ws = self._t(nb // 4)          # main stream only: the two layers run one after the other
        P.add(PR.CONV_KXK_DGRAD, 64, i=(N, Cc, H, W, layer.cout, 3, 2, 1), l=(nb, 0),
              p=(layer.w, dy, dx, self._dgrad_ws, mask), keep=[dy, dx],
              work=2.0 * 9 * Cc * layer.cout * N * dy.shape[2] * dy.shape[3])

    def _aux(self, P):
        """Filter / bias gradients do not feed the data-gradient chain: they run on an auxiliary
        stream behind a FORK (everything they read has been enqueued on the main stream), so that
        they fill the last partial round of workgroups of the chain's kernels and vice versa.  They
        share one workspace, which is safe because they are serialised on that one stream; nothing
        they read is modified by the main stream before the segment's JOIN (gradients that meet are
        written to fresh buffers, not accumulated in place)."""
        self._wstream = self._wstreams[self._wnext % len(self._wstreams)]
        self._wnext += 1
        if self._wstream:
            P.fork(self._wstream)

    def _ew(self, P, code, i=(), p=(), l=(), f=(), nbytes=0.0):
        P.add(code, 51, i=i, p=p, l=l, f=f, work=nbytes)

    # -- program construction ----------------------------------------------------------------------
    def _build(self):
        self._ws_need, self._ws_ops = 0, []
        import os
        ov = self._overlap_wgrad
        on = os.environ.get("SSAD_OVERLAP_WGRAD", "1") == "1" if ov is None else ov
        # filter / bias gradients go round-robin over this many auxiliary streams (each with its own
        # workspace): the small reduce launch that ends one filter gradient then runs beside the next
        # one's main kernel instead of in front of it
        nws = 2      # 1 was indistinguishable, 3 worse
        self._wstreams = list(range(1, nws + 1)) if on else [0]
        self._wstream, self._wnext = self._wstreams[0], 0
        # FPN's stride-2 3x3 layers (P6, P7) at their own size: implicit GEMM with split-K forward, flattened-batch
        # GEMMs for the gradients (conv_strided.hip).  SSAD_STRIDED_3X3=winograd: rounds 1-2's stride-1 Winograd
        # layer + subsampling (4x the direct-form flops), kept for A/B runs.
        self._strided_own = os.environ.get("SSAD_STRIDED_3X3", "own") != "winograd"
        self._dgrad_ws = None
        L = self._layers
        dev = self.device
        lib = K.lib()
        # packed filters: frozen layers once (prepare program), trainable layers every step (pack segment)
        prep, P = PR.Program(), PR.Program()
        self.prep, self.prog = prep, P
        wino_frozen, wino_train, wino24_frozen, wino24_train = [], [], [], []
        # A network that is only evaluated (the distillation step's frozen teacher) runs its 3x3 layers of >= 128
        # outputs on the F(2x4, 3x3) engine: 3 multiplies per output instead of 4, fp32 error ~2e-6 of the output
        # scale (conv3x3_winograd24.hip; SSAD_TEACHER_F24=0: the F(2x2) engine as in rounds 1-4)
        use_f24 = (not self.train) and int(os.environ.get("SSAD_TEACHER_F24", "1")) >= 1
        # SSAD_STUDENT_F24 bit 8 (default on): the TRAINED network's 3x3 layers of >= 128 channels too, forward and data
        # gradient (the filter gradient keeps its F(3x3, 2x2) engine); DESIGN 3.10e has the error and step-time A/B
        use_f24_train = self.train and (int(os.environ.get("SSAD_STUDENT_F24", "15")) & 8) != 0
        # SSAD_SPLIT_CONV bit 16: the >= 256-wide stride-1 3x3 layers (res4, res5, FPN outputs) on the split-operand engine
        # (default on: step -0.1 ... -1.3 ms in four same-box A/B pairs, profiles/r06_experiments.md)
        use_split = (int(os.environ.get("SSAD_SPLIT_CONV", "511")) & 16) != 0 and (use_f24 or use_f24_train)
        split_frozen, split_train = [], []
        self._split_ops, self._split_need = [], 0
        self._gemm_split_ops = []
        self._gemm_split = (int(os.environ.get("SSAD_SPLIT_CONV", "511")) & 128) != 0
        self.amax = torch.zeros(self.AMAX_WORDS, dtype=torch.int32, device=self.device)
        self._amax_groups, self._amax_next = {}, 0
        self._gemm_packs, self._gpack_train, self._gpack_frozen, self._a_train = {}, [], [], {}
        tr_frozen, tr_train = [], []          # (w, wt, M, K, ldm): every transposed filter of a program in one launch
        P.mark("pack")
        for l in L.values():
            tgt = P if l.train else prep
            trs = tr_train if l.train else tr_frozen
            if l.k == 1:
                ldm = (l.cout + 3) // 4 * 4
                l.wt = self._t(l.cin, ldm)
                trs.append((l.w, l.wt, l.cout, l.cin, ldm))
                self._a_train[l.wt.data_ptr()] = self._a_train[l.w.data_ptr()] = l.train
            elif l.k == 3 and l.group > 1:                       # ResNeXt: MFMA operand order, packed once
                l.pf = self._t(lib.ssad_grouped_conv3x3_filter_floats(l.cout, l.group))
                tgt.add(PR.GROUPED_PACK, 54, i=(l.cout, l.group), p=(l.w, l.pf), work=4.0 * (l.w.numel() + l.pf.numel()))
            elif l.k == 3 and l.stride == 2 and self._strided_own:
                # P6 / P7 at their own size: the implicit GEMM's [Cin * 9][Cout] operand (the data and filter
                # gradients read the filter in its natural layout)
                l.wt = self._t(l.cin * 9, l.cout)
                trs.append((l.w, l.wt, l.cout, l.cin * 9, l.cout))
            elif l.k == 3 and use_split and l.cout >= 256 and l.cin >= 256:
                l.pf = self._t(lib.ssad_conv_split_filter_floats(l.cout, l.cin))
                l.pd = self._t(lib.ssad_conv_split_filter_floats(l.cin, l.cout)) if l.train else None
                l.f24 = 3
                (split_train if l.train else split_frozen).append(l)
            elif l.k == 3 and use_f24 and not l.train and l.cout >= 128:
                l.pf = self._t(lib.ssad_conv_wino24_filter_floats(l.cout, l.cin))
                l.f24 = True
                wino24_frozen.append(l)
            elif l.k == 3 and use_f24_train and l.train and l.cout >= 128 and l.cin >= 128:
                l.pf = self._t(lib.ssad_conv_wino24_filter_floats(l.cout, l.cin))
                l.pd = self._t(lib.ssad_conv_wino24_filter_floats(l.cin, l.cout))
                l.f24 = True
                wino24_train.append(l)
            elif l.k == 3:
                l.pf = self._t(lib.ssad_conv_wino_filter_floats(l.cout, l.cin))
                need_pd = l.train                  # every trainable 3x3 sends a gradient further down
                l.pd = self._t(lib.ssad_conv_wino_filter_floats(l.cin, l.cout)) if need_pd else None
                (wino_train if l.train else wino_frozen).append(l)
            else:                                                    # stem: [147][64]
                l.wt = self._t(l.cin * l.k * l.k, l.cout)
                trs.append((l.w, l.wt, l.cout, l.cin * l.k * l.k, l.cout))
        # one launch per program for all transposes (ssad_transpose_filters): as 34 launches of ~5 us they were a
        # serial chain at the start of every step, each waiting for a CU slot behind the other stream's persistent
        # kernels (tools/step_timeline.py: the main stream sat idle for ~7 ms there)
        for tgt, trs in ((prep, tr_frozen), (P, tr_train)):
            if trs:
                tab = (K.TransposeEntry * len(trs))()
                for i, (w, wt, M, Kc, ldm) in enumerate(trs):
                    tab[i] = K.TransposeEntry(w.data_ptr(), wt.data_ptr(), M, Kc, ldm, 0)
                tgt.add(PR.TRANSPOSE_FILTERS, 54, i=(len(trs),), p=(tab,),
                        work=8.0 * sum(M * Kc for (_, _, M, Kc, _) in trs), keep=[t for e in trs for t in e[:2]])
        for tgt, ls in ((prep, wino_frozen), (P, wino_train)):
            if ls:
                tab = (K.PackEntry * len(ls))()
                for i, l in enumerate(ls):
                    tab[i] = K.PackEntry(l.w.data_ptr(), l.cout, l.cin, l.pf.data_ptr(),
                                         l.pd.data_ptr() if l.pd is not None else 0)
                tgt.add(PR.WINO_PACK_FILTERS, 54, i=(len(ls),), p=(tab,),
                        work=4.0 * sum(l.w.numel() + l.pf.numel() + (l.pd.numel() if l.pd is not None else 0)
                                       for l in ls))
        for tgt, ls in ((prep, wino24_frozen), (P, wino24_train)):
            if ls:
                tab = (K.PackEntry * len(ls))()
                for i, l in enumerate(ls):
                    tab[i] = K.PackEntry(l.w.data_ptr(), l.cout, l.cin, l.pf.data_ptr(),
                                         l.pd.data_ptr() if l.pd is not None else 0)
                tgt.add(PR.WINO_PACK_FILTERS, 54, i=(len(ls), 2), p=(tab,),
                        work=4.0 * sum(l.w.numel() + l.pf.numel() + (l.pd.numel() if l.pd is not None else 0)
                                       for l in ls))
        for tgt, ls in ((prep, split_frozen), (P, split_train)):
            if ls:
                tab = (K.PackEntry * len(ls))()
                for i, l in enumerate(ls):
                    tab[i] = K.PackEntry(l.w.data_ptr(), l.cout, l.cin, l.pf.data_ptr(),
                                         l.pd.data_ptr() if l.pd is not None else 0)
                tgt.add(PR.WINO_PACK_FILTERS, 54, i=(len(ls), 3), p=(tab,),
                        work=4.0 * sum(l.w.numel() + l.pf.numel() + (l.pd.numel() if l.pd is not None else 0)
                                       for l in ls))
        # the pointwise filters' split copies (which layers need one is known once the passes are emitted: the table is
        # filled in below)
        gp_max = 2 * sum(1 for l in L.values() if l.k == 1)
        gtab = {True: (K.GemmPackEntry * max(gp_max, 1))(), False: (K.GemmPackEntry * max(gp_max, 1))()}
        gidx = {True: P.add(PR.GEMM_SPLIT_PACK, 74, i=(0,), p=(gtab[True],)),
                False: prep.add(PR.GEMM_SPLIT_PACK, 74, i=(0,), p=(gtab[False],))}
        self._packed_frozen = False
        P.mark("forward")
        P.add(PR.FILL, 51, p=(self.amax,), f=(0.0,), l=(self.AMAX_WORDS,), work=4.0 * self.AMAX_WORDS)
        self._fwd = True
        self._emit_forward(P)
        self._fwd = False
        P.mark("backward")
        if self.train:
            self._emit_backward(P)
            P.mark("sgd")
            tab = (K.SgdSegment * len(self.segments))()
            for i, (off, n, isb, row_len, s2) in enumerate(self.segments):
                tab[i] = K.SgdSegment(off, n, isb, row_len if s2 is not None else 0,
                                      s2.data_ptr() if s2 is not None else None)
            P.add(PR.SGD_FLAT, 55, i=(len(self.segments),), f=(self.momentum, self.weight_decay),
                  p=(self.params_flat, self.grads_flat, self.moms_flat, self.lr, tab, self.skip_flag),
                  work=4.0 * 6 * self.params_flat.numel(),
                  keep=[s2 for (_, _, _, _, s2) in self.segments if s2 is not None])
        P.mark("end")
        for train, entries, prog in ((True, self._gpack_train, P), (False, self._gpack_frozen, prep)):
            for k, (a, lda, Kc, M, dst) in enumerate(entries):
                gtab[train][k] = K.GemmPackEntry(a.data_ptr(), dst.data_ptr(), lda, Kc, M)
            op = prog.ops[gidx[train]]
            op.i[0] = len(entries)
            op.work = 8.0 * sum(Kc * M for (_, _, Kc, M, _) in entries)
        prep.build()
        self.wss = {k: torch.empty(max(self._ws_need, 16), dtype=torch.uint8, device=dev) for k in self._wstreams}
        self.ws = self.wss[self._wstreams[0]]
        for idx, slot, k in self._ws_ops:
            P.set_ptr(idx, slot, self.wss[k])
        # the split-engine 3x3 launches of this network run one after the other on its stream: one workspace
        self.split_ws = torch.empty(max(self._split_need, 16), dtype=torch.uint8, device=dev)
        for idx in self._split_ops:
            P.set_ptr(idx, 3, self.split_ws)
        for idx in self._gemm_split_ops:
            P.set_ptr(idx, 1, self.split_ws)
        P.build()

    # -- forward ----------------------------------------------------------------------------------------
    def _emit_forward(self, P):
        L, N = self._layers, self.N
        H, W = self.hw
        self.image = self._t(N, 3, H, W)
        # stem: im2col + GEMM (K = 147), then bias + ReLU + 3x3/2 max pool in one pass.  The column
        # buffer of the whole batch is ~1.35 GB at 640x896 x 16: processed in image groups below 2 GiB.
        st = L["stem.0"]
        oh, ow = H // 2, W // 2
        kk = 3 * 49
        self.stem_z = self._t(N, 64, oh, ow)
        # stem: 7x7/2 as an implicit GEMM (K = 147 gathered by the DMA: no column buffer -- it was
        # 1.35 GB per 16 images), then bias + ReLU + 3x3/2 max pool in one pass.  Image groups keep
        # every buffer offset below 2 GiB.
        grp = max(1, min(N, int((1 << 31) - 1) // (st.cout * oh * ow * 4)))
        for n0 in range(0, N, grp):
            n1 = min(N, n0 + grp)
            d = K.gemm_conv_desc(st.wt, st.cout, self.image[n0:n1], self.stem_z[n0:n1], kk, st.cout)
            d.P = oh * ow
            P.add(PR.CONV_IMPLICIT, 53, i=(3, H, W, 7, 2, 3), p=(d,), work=2.0 * (n1 - n0) * oh * ow * kk * st.cout,
                  keep=[st.wt, self.image, self.stem_z])
        c1 = self._t(N, 64, oh // 2, ow // 2)
        self._ew(P, PR.STEM_POOL, i=(N, 64, oh, ow, 1), p=(self.stem_z, st.b, c1), nbytes=4.0 * 1.25 * self.stem_z.numel())
        x = c1
        self.saved = {}
        stage_out = {}
        for (stage, j, cin, cmid, cout, stride, proj, tr) in self.blocks:
            pre = "res%d.%d" % (stage, j)
            l1, l2, l3 = L[pre + ".c1"], L[pre + ".c2"], L[pre + ".c3"]
            h, w = x.shape[2] // stride, x.shape[3] // stride
            xs = x
            if stride != 1:
                xs = self._t(N, cin, h, w)
                self._ew(P, PR.SUBSAMPLE, i=(N, cin, x.shape[2], x.shape[3], stride), p=(x, xs), nbytes=8.0 * xs.numel())
            a = xs if l1.stride == stride else x              # ResNeXt strides on the 3x3: c1 sees the full map
            y1 = self._t(N, cmid, a.shape[2], a.shape[3])
            y2, y = self._t(N, cmid, h, w), self._t(N, cout, h, w)
            self._gemm(P, l1.wt, l1.wt.shape[1], a, y1, cin, cmid, bias=l1.b, relu=True)
            if l2.group > 1:
                P.add(PR.GROUPED_CONV3X3, 56, i=(N, cmid, y1.shape[2], y1.shape[3], l2.group, l2.stride, 1),
                      p=(y1, l2.pf, l2.b, y2), work=2.0 * 9 * cmid * l2.wcin * y2.shape[0] * h * w)
            else:
                self._conv3(P, [(y1, y2, None, l2.pf, l2.b)], cmid, cmid, K.CONV_RELU, f24=l2.f24)
            sc = xs
            if proj:
                lp = L[pre + ".proj"]
                sc = self._t(N, cout, h, w)
                self._gemm(P, lp.wt, lp.wt.shape[1], xs, sc, cin, cout, bias=lp.b)
            self._gemm(P, l3.wt, l3.wt.shape[1], y2, y, cmid, cout, bias=l3.b, res=sc, relu=True)
            self.saved[pre] = dict(x=x, xs=xs, y1=y1, y2=y2, y=y)
            x = y
            stage_out[stage] = y
        c3, c4, c5 = stage_out[3], stage_out[4], stage_out[5]
        self.c345 = (c3, c4, c5)
        D = self.D
        # FPN: laterals (GEMM + bias), top-down nearest upsampling + Sum in place
        t5 = self._t(N, D, c5.shape[2], c5.shape[3])
        t4 = self._t(N, D, c4.shape[2], c4.shape[3])
        t3 = self._t(N, D, c3.shape[2], c3.shape[3])
        for t, c, name in ((t5, c5, "lat.0"), (t4, c4, "lat.1"), (t3, c3, "lat.2")):
            l = L[name]
            self._gemm(P, l.wt, l.wt.shape[1], c, t, l.cin, D, bias=l.b, final=t is t5)
            if t is not t5:
                src = t5 if t is t4 else t4
                self._ew(P, PR.UPSAMPLE, i=(N, D, src.shape[2], src.shape[3], 2), p=(src, t, t), nbytes=9.0 * t.numel())
        p5, p4, p3 = (self._like(t) for t in (t5, t4, t3))
        self._conv3(P, [(t, p, None, L[name].pf, L[name].b)                # three filters, one launch
                        for t, p, name in ((t5, p5, "out.0"), (t4, p4, "out.1"), (t3, p3, "out.2"))], D, D, 0,
                   f24=L["out.0"].f24)
        l6, l7 = L["p6"], L["p7"]
        if self._strided_own:
            # P6 / P7 (3x3, stride 2) at their own size
            p6 = self._t(N, D, (c5.shape[2] - 1) // 2 + 1, (c5.shape[3] - 1) // 2 + 1)
            self._conv_s2(P, c5, p6, l6)
            r6 = self._like(p6)
            self._ew(P, PR.RELU, p=(p6, r6), l=(p6.numel(),), nbytes=8.0 * p6.numel())
            p7 = self._t(N, D, (p6.shape[2] - 1) // 2 + 1, (p6.shape[3] - 1) // 2 + 1)
            self._conv_s2(P, r6, p7, l7)
            p6f = p7f = None
        else:
            # SSAD_STRIDED_3X3=winograd: stride-1 convolution, then the even positions
            p6f = self._t(N, D, c5.shape[2], c5.shape[3])
            self._conv3(P, [(c5, p6f, None, l6.pf, l6.b)], D, l6.cin, 0, f24=l6.f24)
            p6 = self._t(N, D, c5.shape[2] // 2, c5.shape[3] // 2)
            self._ew(P, PR.SUBSAMPLE, i=(N, D, c5.shape[2], c5.shape[3], 2), p=(p6f, p6), nbytes=8.0 * p6.numel())
            r6 = self._like(p6)
            self._ew(P, PR.RELU, p=(p6, r6), l=(p6.numel(),), nbytes=8.0 * p6.numel())
            p7f = self._like(p6)
            self._conv3(P, [(r6, p7f, None, l7.pf, l7.b)], D, D, 0, f24=l7.f24)
            p7 = self._t(N, D, (p6.shape[2] + 1) // 2, (p6.shape[3] + 1) // 2)
            self._ew(P, PR.SUBSAMPLE, i=(N, D, p6.shape[2], p6.shape[3], 2), p=(p7f, p7), nbytes=8.0 * p7.numel())
        self.fpn = [p3, p4, p5, p6, p7]                   # finest first (synth.LEVEL_SHAPES_*)
        self._fpn_saved = dict(t3=t3, t4=t4, t5=t5, r6=r6, p6f=p6f, p7f=p7f)

    # -- backward ------------------------------------------------------------------------------------------
    def _emit_backward(self, P):
        L, N, D = self._layers, self.N, self.D
        c3, c4, c5 = self.c345
        S = self._fpn_saved
        t3, t4, t5, r6 = S["t3"], S["t4"], S["t5"], S["r6"]
        # gradient w.r.t. every FPN level, written by the caller (subnet gradients, summed)
        self.d_fpn = [self._like(p) for p in self.fpn]
        d3, d4, d5, d6, d7 = self.d_fpn
        l6, l7 = L["p6"], L["p7"]
        dc5 = self._like(c5)
        if self._strided_own:
            # P7 = conv_s2(relu(p6)): filter gradient, then the data gradient masked by p6 > 0, added to P6's own
            self._wgrad_s2(P, r6, d7, l7)
            dr6 = self._like(r6)
            self._dgrad_s2(P, l7, d7, dr6, mask=r6)
            ptrs = (C.c_void_p * 2)(d6.data_ptr(), dr6.data_ptr())
            P.add(PR.SUM_N, 51, i=(2,), l=(d6.numel(),), p=(ptrs, d6), work=12.0 * d6.numel(), keep=[d6, dr6])
            # P6 = conv_s2(c5)
            self._wgrad_s2(P, c5, d6, l6)
            self._dgrad_s2(P, l6, d6, dc5)
        else:
            # P7 = sub(conv(relu(p6)))
            d7f = self._like(S["p7f"])
            self._ew(P, PR.SUBSAMPLE_GRAD, i=(N, D, d7f.shape[2], d7f.shape[3], 2, 0), p=(d7, d7f), nbytes=4.0 * d7f.numel())
            self._wgrad3(P, r6, d7f, l7)
            dr6 = self._like(r6)
            self._conv3(P, [(d7f, dr6, r6, l7.pd, None)], D, D, K.CONV_MASK_AUX, f24=l7.f24)   # masked by p6 > 0
            ptrs = (C.c_void_p * 2)(d6.data_ptr(), dr6.data_ptr())
            P.add(PR.SUM_N, 51, i=(2,), l=(d6.numel(),), p=(ptrs, d6), work=12.0 * d6.numel(), keep=[d6, dr6])
            # P6 = sub(conv(c5))
            d6f = self._like(S["p6f"])
            self._ew(P, PR.SUBSAMPLE_GRAD, i=(N, D, d6f.shape[2], d6f.shape[3], 2, 0), p=(d6, d6f), nbytes=4.0 * d6f.numel())
            self._wgrad3(P, c5, d6f, l6)
            self._conv3(P, [(d6f, dc5, None, l6.pd, None)], l6.cin, D, 0, f24=l6.f24)
        # output convs: filter gradients and the three data gradients in one launch
        dt5, dt4, dt3 = self._like(t5), self._like(t4), self._like(t3)
        if L["out.0"].f24 == 3:
            self._measure(P, [d5, d4, d3], D)          # one pass for the three filter gradients and the data gradient
        for t, d, name in ((t5, d5, "out.0"), (t4, d4, "out.1"), (t3, d3, "out.2")):
            self._wgrad3(P, t, d, L[name])
        self._conv3(P, [(d, dt, None, L[name].pd, None)
                        for d, dt, name in ((d5, dt5, "out.0"), (d4, dt4, "out.1"), (d3, dt3, "out.2"))], D, D, 0,
                   f24=L["out.0"].f24)
        # top-down path: t3 = lat2(c3) + up(t4), t4 = lat1(c4) + up(t5)
        up4, up5 = self._like(t4), self._like(t5)
        self._ew(P, PR.UPSAMPLE_GRAD, i=(N, D, t4.shape[2], t4.shape[3], 2), p=(dt3, up4), nbytes=5.0 * dt3.numel())
        ptrs4 = (C.c_void_p * 2)(dt4.data_ptr(), up4.data_ptr())
        P.add(PR.SUM_N, 51, i=(2,), l=(dt4.numel(),), p=(ptrs4, dt4), work=12.0 * dt4.numel(), keep=[dt4, up4])
        self._ew(P, PR.UPSAMPLE_GRAD, i=(N, D, t5.shape[2], t5.shape[3], 2), p=(dt4, up5), nbytes=5.0 * dt4.numel())
        ptrs5 = (C.c_void_p * 2)(dt5.data_ptr(), up5.data_ptr())
        P.add(PR.SUM_N, 51, i=(2,), l=(dt5.numel(),), p=(ptrs5, dt5), work=12.0 * dt5.numel(), keep=[dt5, up5])
        # laterals: filter / bias gradients; data gradients meet the stage outputs' other consumers
        dc4, dc3 = self._like(c4), self._like(c3)
        for c, dt, dc, name, acc in ((c5, dt5, dc5, "lat.0", True), (c4, dt4, dc4, "lat.1", False),
                                     (c3, dt3, dc3, "lat.2", False)):
            l = L[name]
            self._wgrad1(P, c, dt, l)
            self._bias_grad(P, dt, l)
            self._gemm(P, l.w.view(l.cout, l.cin), l.cin, dt, dc, l.cout, l.cin, acc=acc)
        P.mark("bwd_fpn_done")
        grads_into = {5: dc5, 4: dc4, 3: dc3}
        dy = None
        dy_is_dz = False          # dy already went through this block's ReluGradient (previous GEMM's epilogue)
        for (stage, j, cin, cmid, cout, stride, proj, tr) in reversed(self.blocks):
            if not tr:
                break
            pre = "res%d.%d" % (stage, j)
            l1, l2, l3 = L[pre + ".c1"], L[pre + ".c2"], L[pre + ".c3"]
            sv = self.saved[pre]
            x, xs, y1, y2, y = sv["x"], sv["xs"], sv["y1"], sv["y2"], sv["y"]
            last_of_stage = j == ARCHS[self.arch][stage - 2] - 1
            if last_of_stage:
                # the stage's output feeds the lateral (already in grads_into) and, for res3 / res4,
                # the next stage's first block, whose input gradient was accumulated onto it
                dy = grads_into[stage]
                dy_is_dz = False
            h, w = y.shape[2], y.shape[3]
            if dy_is_dz:
                # the block below (later in the forward order) produced dy with this block's
                # ReluGradient mask already applied in its GEMM epilogue: no elementwise pass
                dz = dy
            else:
                # dz = ReluGradient(y, dy).  (No bias gradients in the body: the folded AffineChannel
                # biases are frozen values, affine_channel_op.cc.)
                dz = self._like(y)
                self._ew(P, PR.RELU_GRAD, p=(y, dy, dz), l=(y.numel(),), nbytes=12.0 * y.numel())
            self._wgrad1(P, y2, dz, l3)
            dz2 = self._like(y2)
            self._gemm(P, l3.w.view(cout, cmid), cmid, dz, dz2, cout, cmid, mask=y2, fold=True)
            self._wgrad3(P, y1, dz2, l2)                                   # + bias gradient of c2
            dz1 = self._like(y1)
            self._conv3(P, [(dz2, dz1, y1, l2.pd, None)], cmid, cmid, K.CONV_MASK_AUX, f24=l2.f24, fold=True)
            self._wgrad1(P, xs, dz1, l1)
            first_trainable = (stage == 3 and j == 0)
            if proj:
                lp = L[pre + ".proj"]
                self._wgrad1(P, xs, dz, lp)
                if not first_trainable:
                    dxs = self._like(xs)
                    self._gemm(P, l1.w.view(cmid, cin), cin, dz1, dxs, cmid, cin)
                    self._gemm(P, lp.w.view(cout, cin), cin, dz, dxs, cout, cin, acc=True)
                    # into the previous stage's output gradient (which already holds the lateral's part)
                    tgt = grads_into[stage - 1]
                    self._ew(P, PR.SUBSAMPLE_GRAD, i=(N, cin, x.shape[2], x.shape[3], stride, 1), p=(dxs, tgt),
                             nbytes=12.0 * tgt.numel())
                dy = None
                dy_is_dz = False
            else:
                # identity shortcut: dx = dz + W1^T dz1 (dz through the residual operand into a fresh
                # buffer: the auxiliary stream may still be reading dz)
                # x is the previous block's output y: its ReluGradient mask goes into this epilogue
                dx = self._like(dz)
                self._gemm(P, l1.w.view(cmid, cin), cin, dz1, dx, cmid, cin, res=dz, mask=x, fold=True)
                dy = dx
                dy_is_dz = True
            if j == 0:
                P.mark("bwd_res%d_done" % stage)

    # -- running ---------------------------------------------------------------------------------------------
    def prepare(self):
        """Pack the frozen filters (once, and again after load_from)."""
        if not self._packed_frozen:
            self.prep.run(timing=None)
            self._packed_frozen = True

    def pack(self):
        self.prepare()
        self.prog.run("pack", "forward", timing=self.timing)

    def forward(self, images):
        """images: [N, 3, H, W] float32.  Returns the five FPN levels (finest first)."""
        if images.data_ptr() != self.image.data_ptr():
            self.image.copy_(images)
        self.prepare()
        self.prog.run("forward", "backward", timing=self.timing)
        return self.fpn

    def backward(self, d_fpn=None):
        """d_fpn: gradients w.r.t. the five levels (or already written into self.d_fpn).  Each
        stage's gradient bucket is all-reduced as soon as its last filter gradient is enqueued."""
        if d_fpn is not None:
            for dst, src in zip(self.d_fpn, d_fpn):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        marks = ["backward", "bwd_fpn_done", "bwd_res5_done", "bwd_res4_done", "bwd_res3_done"]
        names = ["fpn", "res5", "res4", "res3"]
        for k in range(4):
            self.prog.run(marks[k], marks[k + 1], timing=self.timing)
            self.dp.issue(self.bucket[names[k]])
        # (the program has nothing between bwd_res3_done and sgd)

    def sgd_step(self):
        self.dp.wait()
        self.prog.run("sgd", "end", timing=self.timing)

    def broadcast_params(self, src=0):
        """detectron/lib/utils/net.py:185-208 broadcasts every blob of model.params from GPU 0: here the
        trained parameters and their history, the FROZEN values (conv1 / res2 filters, every folded
        AffineChannel bias) and the s^2 row scales of the folded trainable filters -- a replica that
        did not load the weights file itself must not keep its own initialisation in any of them.
        The un-folded scales kept for saving (`affine_scale_values`, host side) follow as an object."""
        if not self.dp.active:
            return
        slots = [l.s2 for l in self._layers.values() if l.s2 is not None]
        self.dp.broadcast([self.params_flat, self.moms_flat, self.frozen_flat] + slots, src=src)
        import torch.distributed as dist
        box = [self.affine_scale_values]
        dist.broadcast_object_list(box, src=src, group=self.dp.pg)
        self.affine_scale_values = box[0]
        self._packed_frozen = False

    SCALE_MOMENTUM = True             # cfg.SOLVER.SCALE_MOMENTUM (config.py:634)
    SCALE_MOMENTUM_THRESHOLD = 1.1    # config.py:638

    def update_lr(self, new_lr):
        """UpdateWorkspaceLr + _CorrectMomentum (detector.py:594-648) for the backbone's flat
        buffers: same rule as DistillHeads.update_lr."""
        cur_lr = float(self.lr.item())
        new_lr = float(np.float32(new_lr))
        if cur_lr == new_lr or not self.train:
            return new_lr
        eps = 1e-10
        ratio = max(new_lr / max(cur_lr, eps), cur_lr / max(new_lr, eps))
        self.lr.fill_(new_lr)
        if self.SCALE_MOMENTUM and cur_lr > 1e-7 and ratio > self.SCALE_MOMENTUM_THRESHOLD:
            K.scale_(self.moms_flat, new_lr / cur_lr)
        return new_lr


class NativeDistillModel(object):
    """One iteration of the whole detector on one GPU with native backbones: teacher forward
    (its own stream), student forward, subnets + losses (head_pipeline.DistillHeads), student
    backward, gradient all-reduce (when data parallel) and both SGD updates -- every launch one
    of this repo's kernels, enqueued by ssad_program_run."""

    def __init__(self, heads, student_arch="r50", teacher_arch="r101", N=16, image_hw=(640, 896), device="cuda",
                 process_group=None, world_size=1, lr=None, momentum=0.9, weight_decay=1e-4, two_streams=None,
                 overlap_wgrad=None, student_src=None, teacher_src=None, student_scales=None):
        """lr: None = the subnets' learning rate (one schedule for the whole detector,
        optimizer.py:95-130).  two_streams / overlap_wgrad: None = environment
        (SSAD_NATIVE_TWO_STREAMS / SSAD_OVERLAP_WGRAD, default on).  *_src: initial weights
        (NativeResNetFPN._alloc_params), student_scales: the folded AffineChannel scales."""
        import os
        self.heads = heads
        self.has_teacher = teacher_arch not in (None, "none")
        assert self.has_teacher == bool(getattr(heads, "distill", True))
        lr = float(heads.lr.item()) if lr is None else lr
        self.f16 = bool(getattr(heads, "F16", False))
        # fp16 subnets that exchange blocked fp16 tensors with the backbone: the backbones run in the
        # same precision (config 5: every convolution of the net, conv_op_cudnn.cc:631-636)
        self.backbone_f16 = self.f16 and bool(getattr(heads, "blocked_io", False))
        kw = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, process_group=process_group,
                  world_size=world_size, affine_scales=student_scales,
                  skip_flag=heads.ls_counters if self.f16 else None, overlap_wgrad=overlap_wgrad)
        if self.backbone_f16:
            from .backbone_f16 import NativeResNetFPNF16
            self.student = NativeResNetFPNF16(
                student_arch, N, image_hw, device, train=True, src=student_src,
                heads_io=dict(fpn_out=heads.in_blk["student"], inv_scale=heads.ls_state[1:2],
                              d_fpn_in=(heads.dbuf["cls"][0], heads.dbuf["bbox"][0])), **kw)
            self.teacher = NativeResNetFPNF16(
                teacher_arch, N, image_hw, device, train=False, src=teacher_src, process_group=process_group,
                world_size=world_size, heads_io=dict(fpn_out=heads.in_blk["teacher"])) if self.has_teacher else None
        else:
            self.student = NativeResNetFPN(student_arch, N, image_hw, device, train=True, src=student_src, **kw)
            self.teacher = NativeResNetFPN(teacher_arch, N, image_hw, device, train=False, src=teacher_src,
                                           process_group=process_group,
                                           world_size=world_size) if self.has_teacher else None
        # every replica starts from rank 0's values: trained AND frozen (utils/net.py:185-208)
        self.student.broadcast_params()
        if self.teacher is not None:
            self.teacher.broadcast_params()
        if two_streams is None:
            two_streams = os.environ.get("SSAD_NATIVE_TWO_STREAMS", "1") == "1"
        self.side = torch.cuda.Stream() if (two_streams and self.has_teacher) else None      # normal priority: high measured worse
        # the teacher's forward pass is enqueued BEFORE the student's filter packs (attribute kept for A/B in tests)
        self._teacher_first = True
        # The frozen teacher's forward pass depends on the images only -- not on the student's update.  Ordered after
        # the previous step's last READER of the teacher's output buffers (the subnets' forward launches) instead of
        # after everything the previous step enqueued, it may start while the previous step's backward pass is still
        # running whenever the launching thread is ahead of the GPU (it is: ~5 steps in steady state).  Taken only
        # when `images` cannot have a pending writer on the current stream (step(): `images_event`, or the same
        # unmodified tensor as the step before); otherwise the teacher waits for the current stream as before.
        # SSAD_TEACHER_AHEAD=0: the teacher waits for the whole previous step (rounds 2-3).
        # Measured (same-call A/B, 20 steps): config 3 (fp32) 95.7 -> 93.4 ms/step; config 5 (fp16 backbones, small
        # HBM-bound launches) 26.07 -> 26.8 ms -- there the teacher beside the backward pass costs more than it
        # fills, so the fp16 backbones keep the old order by default.
        self._teacher_ahead = os.environ.get("SSAD_TEACHER_AHEAD", "0" if self.backbone_f16 else "1") == "1"
        self._t_fpn_read = None               # event: the subnets have consumed the teacher's FPN levels
        self._images_ref, self._images_version = None, -1
        # gradient w.r.t. an FPN level = cls-subnet part + bbox-subnet part (the fp16 backbone's own
        # program starts with that sum, on the blocked tensors)
        Q = self.sum_prog = PR.Program()
        self._ptrs = []
        if not self.backbone_f16:
            for a, b, d in zip(heads.d_fpn["cls"], heads.d_fpn["bbox"], self.student.d_fpn):
                ptrs = (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())
                Q.add(PR.SUM_N, 51, i=(2,), l=(d.numel(),), p=(ptrs, d), work=12.0 * d.numel(), keep=[a, b, d])
        Q.build()
        if self.f16:
            # mixed precision: the backbone's reduced gradients join the subnets' finiteness check
            # (an overflowing step must drop BOTH updates; the flag is cleared after both)
            Q = self.check_prog = PR.Program()
            n = self.student.grads_flat.numel()
            Q.add(PR.CHECK_FINITE, 38, l=(n,), p=(self.student.grads_flat, heads.ls_counters), work=4.0 * n)
            Q.build()
        self._timing = None

    def update_lr(self, new_lr):
        """One learning-rate schedule for subnets and backbone (detector.py:594-648)."""
        self.heads.update_lr(new_lr)
        return self.student.update_lr(new_lr)

    @property
    def timing(self):
        return self._timing

    @timing.setter
    def timing(self, t):
        self._timing = t
        self.student.timing = t
        if self.teacher is not None:
            self.teacher.timing = t if self.side is None else None   # one timing object per stream

    def describe(self):
        if self.backbone_f16:
            return ("backbones = native programs of this repo's kernels in fp16 storage / fp32 accumulation "
                    "(pointwise convs: pw_f16_kernel with fused bias / shortcut / ReLU / FPN upsample-add; 3x3: "
                    "conv3x3_f16_kernel; ResNeXt grouped 3x3: grouped_f16_kernel; stem: fp32 implicit GEMM + fp16 "
                    "pool; no torch operator in the step)")
        return ("backbones = native programs of this repo's kernels (pointwise convs: fp32-MFMA GEMM with fused "
                "bias / shortcut / ReLU; 3x3: Winograd engine; stem: im2col + GEMM + fused bias/ReLU/pool; "
                "no torch operator in the step)")

    def step(self, images, labels, bbox_targets, fg_num, update=True, images_event=None):
        """One iteration.  update=False stops after the gradient exchange (every parameter
        gradient is then readable: the SGD launch overwrites the gradient buffers with the
        applied update, as MomentumSGDUpdate does, momentum_sgd_op_gpu.cu:22-38)."""
        h, st, te = self.heads, self.student, self.teacher
        # The frozen teacher needs nothing of this step but the images: its forward pass goes to the side stream
        # FIRST, so that the ~35 (fp32) / ~110 (fp16) small filter-pack launches of the student -- a serial chain
        # of 5-10 us kernels that leaves the chip empty -- run beside it instead of in front of it.
        t_fpn = None
        early = self._teacher_first and te is not None and self.side is not None
        if early:
            cur = torch.cuda.current_stream()
            # Running ahead is only safe when nothing enqueued on the current stream can still be WRITING `images`:
            # the caller says so with `images_event` (recorded by the input pipeline after its copy), or the tensor
            # is the very OBJECT the previous step already read, unmodified since (same version counter).  The
            # model keeps a reference to that tensor: a fresh batch tensor can then never be mistaken for it (its
            # storage cannot be handed out again by the caching allocator while the reference lives, and a fresh
            # tensor's version counter starts at 0 just like the old one's -- address + version alone would match
            # a new batch whose host-to-device copy is still queued on the current stream, which the side stream
            # does not wait for).  Anything else -- new tensor, view, in-place update -- takes the full wait.
            known = images_event is not None or (images is self._images_ref and
                                                 images._version == self._images_version)
            self._images_ref, self._images_version = images, images._version
            if self._teacher_ahead and self._t_fpn_read is not None and known:
                self.side.wait_event(self._t_fpn_read)      # the previous step's readers of the teacher's FPN buffers
                if images_event is not None:
                    self.side.wait_event(images_event)
            else:
                self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                t_fpn = te.forward(images)
        h.pack_student()
        st.pack()
        if te is not None and not early:
            if self.side is not None:
                self.side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.side):
                    t_fpn = te.forward(images)
            else:
                t_fpn = te.forward(images)
        s_fpn = st.forward(images)
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        if self.backbone_f16:
            t_fpn = s_fpn = None              # already in the subnets' blocked input buffers
        h.forward_all(t_fpn, s_fpn)
        h.cls_losses(labels, fg_num)
        h.bbox_losses_fwd_bwd(bbox_targets, fg_num)
        if self.side is not None and self._teacher_ahead:
            # the teacher's FPN levels have been read (forward_all); recorded behind the loss launches, which are
            # HBM-bound and short (0.45 ms): the next step's teacher starts beside the backward pass, not beside them
            if self._t_fpn_read is None:
                self._t_fpn_read = torch.cuda.Event()
            self._t_fpn_read.record(torch.cuda.current_stream())
        h.backward()
        self.sum_prog.run(timing=self._timing)
        st.backward()
        if not update:
            h.wait_gradients()
            st.dp.wait()
            return h.losses
        if self.f16:
            h.wait_gradients()
            st.dp.wait()
            h.prog.run("sgd", "sgd_update", timing=self._timing)          # subnet gradients finite?
            self.check_prog.run(timing=self._timing)                       # backbone gradients finite?
            h.prog.run("sgd_update", "ls_update", timing=self._timing)    # both updates honour the flag
            st.sgd_step()
            h.prog.run("ls_update", "end", timing=self._timing)           # scale moves, flag cleared
        else:
            h.sgd_step()
            st.sgd_step()
        return h.losses
