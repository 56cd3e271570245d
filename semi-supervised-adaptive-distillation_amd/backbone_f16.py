"""ResNet / ResNeXt-FPN backbones with fp16 storage and fp32 accumulation -- BASELINE config 5's
precision -- as native programs of this repo's kernels: every convolution of the network in the
arrangement of the reference's only fp16 route, CudnnConvOp<float16> with fp32 math
(caffe2/caffe2/operators/conv_op_cudnn.cc:631-636, :1115-1124), for the networks of
detectron/lib/modeling/ResNet.py:85-130,221-283 and FPN.py:116-250.

Mixed precision in the usual arrangement (head_pipeline.DistillHeadsF16): fp32 master
parameters, momentum and parameter gradients (NativeResNetFPN's flat buffers, SGD and all-reduce
unchanged); filters re-rounded to fp16 every step; activations and their gradients
channel-blocked fp16, Xb[n][c/8][y][x][8]; gradients carry the subnets' dynamic loss scale S and
every filter / bias gradient is multiplied by 1/S on its way into the fp32 buffers.

Kernels:
  * pointwise layers (bottleneck 1x1s, projection shortcuts, laterals): gemm_f16.hip -- bias,
    shortcut Sum, ReLU, FPN's top-down upsample + Sum (forward) and the ReluGradient mask / gradient
    Sum (data gradient) in the epilogue; filter gradient = conv3x3_wgrad_f16_kernel<true>;
  * 3x3 / stride 1: conv3x3_f16.hip (forward, masked data gradient, filter + bias gradient);
  * ResNeXt's grouped 3x3 (the X-101-64x4d teacher): grouped_f16.hip, forward;
  * stride-2 3x3 layers (P6, P7, ResNeXt's first blocks): the stride-1 layer + the even positions;
  * the 7x7 / 2 stem: the fp32 implicit GEMM on the fp32 image (Cin = 3), then bias + ReLU +
    3x3 / 2 max pool written blocked fp16 (ssad_stem_pool_f16);
  * everything else: one-slot-per-thread passes on blocked tensors (ssad_f16_elementwise).
"""
import ctypes as C

import torch

from . import kernels as K
from . import program as PR
from .backbone_pipeline import NativeResNetFPN, ARCHS, GROUPED

KL_PW, KL_C3, KL_W3, KL_W1, KL_GR, KL_EW, KL_PACK = 57, 58, 59, 60, 61, 62, 63


class NativeResNetFPNF16(NativeResNetFPN):
    """heads_io (optional): dict(fpn_out=[5 blocked fp16 tensors the FPN levels are written into],
    d_fpn_in=([5], [5]) the two blocked fp16 gradient sets to be summed (cls / bbox subnet),
    inv_scale=1-element float32 device tensor holding 1 / loss scale).  Without it the network owns
    its FPN outputs (self.fpn, blocked fp16) and gradient inputs (self.d_fpn)."""

    F16 = True

    def __init__(self, arch="r50", N=2, image_hw=(640, 896), device="cuda", train=True, heads_io=None, **kw):
        self._io = heads_io or {}
        self._act = []
        super().__init__(arch, N, image_hw, device, train=train, **kw)

    # -- buffers ---------------------------------------------------------------------------
    def _b(self, N, Cc, H, W):
        """Blocked fp16 activation [N][C/8][H][W][8]."""
        t = torch.empty((N, (Cc + 7) // 8, H, W, 8), dtype=torch.float16, device=self.device)
        self._bufs.append(t)
        return t

    def _blike(self, t):
        u = torch.empty_like(t)
        self._bufs.append(u)
        return u

    def poison(self, value=float("nan")):
        for t in self._bufs:
            t.fill_(value)
        self._packed_frozen = False

    # -- emit helpers ------------------------------------------------------------------------
    def _pw(self, P, x, w, y, Cc, M, bias=None, res=None, mask=None, relu=False, res_up=False, stride=1):
        d = K.PwF16()
        d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.residual = res.data_ptr() if res is not None else None
        d.mask = mask.data_ptr() if mask is not None else None
        d.N, d.C, d.M = y.shape[0], Cc, M
        d.Ho, d.Wo, d.Hi, d.Wi, d.stride = y.shape[2], y.shape[3], x.shape[2], x.shape[3], stride
        d.flags = (K.CONV_RELU if relu else 0) | (K.PW_F16_RES_UPSAMPLE2 if res_up else 0)
        px = y.shape[0] * y.shape[2] * y.shape[3]
        P.add(PR.PW_F16, KL_PW, p=(d,), work=2.0 * px * Cc * M,
              keep=[t for t in (x, w, y, bias, res, mask) if t is not None])

    def _c3(self, P, probs, Cin, Cout, flags):
        """probs: [(x, y, mask or None, packed, bias or None)]: independent 3x3 convolutions of one
        (Cin, Cout) in one launch (conv3x3_f16_kernel)."""
        arr = (K.F16Level * len(probs))()
        for i, (x, y, mask, packed, bias) in enumerate(probs):
            arr[i].x, arr[i].y = x.data_ptr(), y.data_ptr()
            arr[i].aux = mask.data_ptr() if mask is not None else None
            arr[i].N, arr[i].H, arr[i].W = x.shape[0], x.shape[2], x.shape[3]
            arr[i].packed = packed.data_ptr()
            arr[i].bias = bias.data_ptr() if bias is not None else None
        px = sum(p[0].shape[0] * p[0].shape[2] * p[0].shape[3] for p in probs)
        P.add(PR.F16_CONV3X3, KL_C3, i=(len(probs), Cin, Cout, flags), p=(arr, None, None),
              work=2.0 * 9 * Cout * Cin * px, keep=[t for p in probs for t in p if t is not None])

    def _ew16(self, P, mode, a, b, y, Cc, stride=1, acc=0):
        P.add(PR.F16_EW, KL_EW, i=(mode, y.shape[0], Cc, y.shape[2], y.shape[3], stride, acc), p=(a, b, y),
              work=2.0 * (y.numel() + a.numel() + (b.numel() if b is not None else 0)))

    def _wg3(self, P, x, dy, layer):
        arr = (K.F16WgradLevel * 1)()
        arr[0].x, arr[0].dy = x.data_ptr(), dy.data_ptr()
        arr[0].N, arr[0].H, arr[0].W = x.shape[0], x.shape[2], x.shape[3]
        nb = K.lib().ssad_conv3x3_wgrad_f16_levels_workspace_bytes(arr, 1, layer.cin, layer.cout)
        self._ws_need = max(self._ws_need, nb)
        self._aux(P)
        idx = P.add(PR.F16_WGRAD, KL_W3, i=(1, layer.cin, layer.cout, 0), f=(1.0,), l=(nb,),
                    p=(arr, self.inv_scale, layer.gw, layer.gb, None),
                    work=2.0 * 9 * layer.cout * layer.cin * x.shape[0] * x.shape[2] * x.shape[3], keep=[x, dy],
                    stream=self._wstream)
        self._ws_ops.append((idx, 4, self._wstream))

    def _wg1(self, P, x, dy, layer):
        N, H, W = x.shape[0], x.shape[2], x.shape[3]
        nb = K.lib().ssad_conv1x1_wgrad_f16_workspace_bytes(N, layer.cin, H, W, layer.cout)
        self._ws_need = max(self._ws_need, nb)
        self._aux(P)
        idx = P.add(PR.PW_F16_WGRAD, KL_W1, i=(N, layer.cin, H, W, layer.cout, 0), f=(1.0,), l=(nb,),
                    p=(x, dy, self.inv_scale, layer.gw, layer.gb, None),
                    work=2.0 * N * H * W * layer.cin * layer.cout, keep=[x, dy], stream=self._wstream)
        self._ws_ops.append((idx, 5, self._wstream))

    # -- program construction --------------------------------------------------------------------
    def _build(self):
        import os
        self._ws_need, self._ws_ops = 0, []
        ov = self._overlap_wgrad
        on = os.environ.get("SSAD_OVERLAP_WGRAD", "1") == "1" if ov is None else ov
        # filter / bias gradients go round-robin over this many auxiliary streams (each with its own
        # workspace): the small reduce launch that ends one filter gradient then runs beside the next
        # one's main kernel instead of in front of it
        nws = 2      # 1 was indistinguishable, 3 worse
        self._wstreams = list(range(1, nws + 1)) if on else [0]
        self._wstream, self._wnext = self._wstreams[0], 0
        L, dev, lib = self._layers, self.device, K.lib()
        inv = self._io.get("inv_scale")
        self.inv_scale = inv if inv is not None else torch.ones(1, dtype=torch.float32, device=dev)
        prep, P = PR.Program(), PR.Program()
        self.prep, self.prog = prep, P
        P.mark("pack")
        h16 = dict(dtype=torch.float16, device=dev)
        # every fp16 filter of a program in ONE launch (ssad_f16_pack_filters): as 111 launches of 5-13 us (R-101
        # student) they were a serial chain at the head of every step
        packs = {True: [], False: []}              # trainable (every step) / frozen (once): (w, wf, wd, M, C, taps)
        for l in L.values():
            tgt = P if l.train else prep
            if l.k == 1 or (l.k == 3 and l.group == 1):
                n = (lib.ssad_pw_f16_filter_halves if l.k == 1 else lib.ssad_f16_filter_halves)(l.cout, l.cin)
                l.pf = torch.empty(n, **h16)
                l.pd = torch.empty(n, **h16) if l.train else None
                packs[bool(l.train)].append((l.w, l.pf, l.pd, l.cout, l.cin, 1 if l.k == 1 else 9))
            elif l.k == 3:
                n = lib.ssad_grouped_conv3x3_f16_filter_halves(l.cout, l.group)
                l.pf = torch.empty(n, **h16)
                tgt.add(PR.GROUPED_F16_PACK, KL_PACK, i=(l.cout, l.group), p=(l.w, l.pf),
                        work=4.0 * l.w.numel() + 2.0 * n)
            else:                                                    # stem: fp32 [147][64] (Cin = 3)
                l.wt = self._t(l.cin * l.k * l.k, l.cout)
                tgt.add(PR.TRANSPOSE_FILTER, 54, i=(l.cout, l.cin * l.k * l.k, l.cout), p=(l.w, l.wt),
                        work=8.0 * l.w.numel())
        for train, tgt in ((True, P), (False, prep)):
            ents = packs[train]
            if not ents:
                continue
            tab = (K.F16PackEntry * len(ents))()
            for i, (w, wf, wd, M, Cc, taps) in enumerate(ents):
                tab[i] = K.F16PackEntry(w.data_ptr(), wf.data_ptr(), wd.data_ptr() if wd is not None else None,
                                        M, Cc, taps, 0)
            tgt.add(PR.F16_PACK_FILTERS, KL_PACK, i=(len(ents),), p=(tab,),
                    work=sum(4.0 * w.numel() + 2.0 * (wf.numel() + (wd.numel() if wd is not None else 0))
                             for (w, wf, wd, M, Cc, taps) in ents),
                    keep=[t for e in ents for t in e[:3] if t is not None])
        prep.build()
        self._packed_frozen = False
        P.mark("forward")
        self._emit_forward(P)
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
        self.wss = {k: torch.empty(max(self._ws_need, 16), dtype=torch.uint8, device=dev) for k in self._wstreams}
        self.ws = self.wss[self._wstreams[0]]
        for idx, slot, k in self._ws_ops:
            P.set_ptr(idx, slot, self.wss[k])
        P.build()

    # -- forward -----------------------------------------------------------------------------------
    def _emit_forward(self, P):
        L, N, D = self._layers, self.N, self.D
        H, W = self.hw
        self.image = self._t(N, 3, H, W)
        st = L["stem.0"]
        oh, ow = H // 2, W // 2
        kk = 3 * 49
        self.stem_z = self._t(N, 64, oh, ow)
        grp = max(1, min(N, int((1 << 31) - 1) // (st.cout * oh * ow * 4)))
        for n0 in range(0, N, grp):
            n1 = min(N, n0 + grp)
            d = K.gemm_conv_desc(st.wt, st.cout, self.image[n0:n1], self.stem_z[n0:n1], kk, st.cout)
            d.P = oh * ow
            P.add(PR.CONV_IMPLICIT, 53, i=(3, H, W, 7, 2, 3), p=(d,), work=2.0 * (n1 - n0) * oh * ow * kk * st.cout,
                  keep=[st.wt, self.image, self.stem_z])
        c1 = self._b(N, 64, oh // 2, ow // 2)
        P.add(PR.STEM_POOL_F16, KL_EW, i=(N, 64, oh, ow), p=(self.stem_z, st.b, c1),
              work=4.0 * self.stem_z.numel() + 2.0 * c1.numel())
        x = c1
        self.saved = {}
        stage_out = {}
        for (stage, j, cin, cmid, cout, stride, proj, tr) in self.blocks:
            pre = "res%d.%d" % (stage, j)
            l1, l2, l3 = L[pre + ".c1"], L[pre + ".c2"], L[pre + ".c3"]
            h, w = x.shape[2] // stride, x.shape[3] // stride
            xs = x
            if stride != 1:                                    # shared by c1 (ResNet) and the projection
                xs = self._b(N, cin, h, w)
                self._ew16(P, K.EW_SUBSAMPLE, x, None, xs, cin, stride=stride)
            a = xs if l1.stride == stride else x               # ResNeXt strides on the 3x3: c1 sees the full map
            y1 = self._b(N, cmid, a.shape[2], a.shape[3])
            y2, y = self._b(N, cmid, h, w), self._b(N, cout, h, w)
            self._pw(P, a, l1.pf, y1, cin, cmid, bias=l1.b, relu=True)
            if l2.group > 1:
                y2f = y2 if l2.stride == 1 else self._b(N, cmid, y1.shape[2], y1.shape[3])
                P.add(PR.GROUPED_F16, KL_GR, i=(N, cmid, y1.shape[2], y1.shape[3], l2.group, 1),
                      p=(y1, l2.pf, l2.b, y2f), work=2.0 * 9 * cmid * l2.wcin * N * y1.shape[2] * y1.shape[3],
                      keep=[y1, y2f])
                if l2.stride != 1:                             # ReLU commutes with taking the even positions
                    self._ew16(P, K.EW_SUBSAMPLE, y2f, None, y2, cmid, stride=l2.stride)
            else:
                self._c3(P, [(y1, y2, None, l2.pf, l2.b)], cmid, cmid, K.CONV_RELU)
            sc = xs
            if proj:
                lp = L[pre + ".proj"]
                sc = self._b(N, cout, h, w)
                self._pw(P, xs, lp.pf, sc, cin, cout, bias=lp.b)
            self._pw(P, y2, l3.pf, y, cmid, cout, bias=l3.b, res=sc, relu=True)
            self.saved[pre] = dict(x=x, xs=xs, y1=y1, y2=y2, y=y)
            x = y
            stage_out[stage] = y
        c3, c4, c5 = stage_out[3], stage_out[4], stage_out[5]
        self.c345 = (c3, c4, c5)
        # FPN: laterals with the top-down upsample + Sum in the GEMM's epilogue (FPN.py:283-306)
        t5 = self._b(N, D, c5.shape[2], c5.shape[3])
        t4 = self._b(N, D, c4.shape[2], c4.shape[3])
        t3 = self._b(N, D, c3.shape[2], c3.shape[3])
        self._pw(P, c5, L["lat.0"].pf, t5, 2048, D, bias=L["lat.0"].b)
        self._pw(P, c4, L["lat.1"].pf, t4, 1024, D, bias=L["lat.1"].b, res=t5, res_up=True)
        self._pw(P, c3, L["lat.2"].pf, t3, 512, D, bias=L["lat.2"].b, res=t4, res_up=True)
        outs = self._io.get("fpn_out")
        if outs is not None:
            p3, p4, p5, p6, p7 = outs
        else:
            p5, p4, p3 = (self._blike(t) for t in (t5, t4, t3))
            p6 = self._b(N, D, c5.shape[2] // 2, c5.shape[3] // 2)
            p7 = self._b(N, D, p6.shape[2] // 2, p6.shape[3] // 2)
        for big in (c5, p6):
            if big.shape[2] % 2 or big.shape[3] % 2:
                raise K.KernelError("fp16 backbone: P6 / P7 need even res5 / P6 maps")
        self._c3(P, [(t, p, None, L[name].pf, L[name].b)
                     for t, p, name in ((t5, p5, "out.0"), (t4, p4, "out.1"), (t3, p3, "out.2"))], D, D, 0)
        l6, l7 = L["p6"], L["p7"]
        p6f = self._b(N, D, c5.shape[2], c5.shape[3])
        self._c3(P, [(c5, p6f, None, l6.pf, l6.b)], l6.cin, D, 0)
        self._ew16(P, K.EW_SUBSAMPLE, p6f, None, p6, D, stride=2)
        r6 = self._blike(p6)
        self._ew16(P, K.EW_RELU, p6, None, r6, D)
        p7f = self._blike(p6)
        self._c3(P, [(r6, p7f, None, l7.pf, l7.b)], D, D, 0)
        self._ew16(P, K.EW_SUBSAMPLE, p7f, None, p7, D, stride=2)
        self.fpn = [p3, p4, p5, p6, p7]
        self._fpn_saved = dict(t3=t3, t4=t4, t5=t5, r6=r6, p6f=p6f, p7f=p7f)

    # -- backward ------------------------------------------------------------------------------------
    def _emit_backward(self, P):
        L, N, D = self._layers, self.N, self.D
        c3, c4, c5 = self.c345
        S = self._fpn_saved
        t3, t4, t5, r6 = S["t3"], S["t4"], S["t5"], S["r6"]
        # gradient w.r.t. every FPN level (times the loss scale): the sum of the two subnets' parts
        self.d_fpn = [self._blike(p) for p in self.fpn]
        din = self._io.get("d_fpn_in")
        if din is not None:
            for a, b, d in zip(din[0], din[1], self.d_fpn):
                self._ew16(P, K.EW_SUM2, a, b, d, D)
        d3, d4, d5, d6, d7 = self.d_fpn
        l6, l7 = L["p6"], L["p7"]
        # P7 = sub(conv(relu(p6)))
        d7f = self._blike(S["p7f"])
        self._ew16(P, K.EW_SUBSAMPLE_GRAD, d7, None, d7f, D, stride=2)
        self._wg3(P, r6, d7f, l7)
        dr6 = self._blike(r6)
        self._c3(P, [(d7f, dr6, r6, l7.pd, None)], D, D, K.CONV_MASK_AUX)          # masked by p6 > 0
        d6s = self._blike(d6)
        self._ew16(P, K.EW_SUM2, d6, dr6, d6s, D)
        # P6 = sub(conv(c5))
        d6f = self._blike(S["p6f"])
        self._ew16(P, K.EW_SUBSAMPLE_GRAD, d6s, None, d6f, D, stride=2)
        self._wg3(P, c5, d6f, l6)
        dc5a = self._blike(c5)
        self._c3(P, [(d6f, dc5a, None, l6.pd, None)], D, l6.cin, 0)
        # output convs
        dt5, dt4, dt3 = self._blike(t5), self._blike(t4), self._blike(t3)
        for t, d, name in ((t5, d5, "out.0"), (t4, d4, "out.1"), (t3, d3, "out.2")):
            self._wg3(P, t, d, L[name])
        self._c3(P, [(d, dt, None, L[name].pd, None)
                     for d, dt, name in ((d5, dt5, "out.0"), (d4, dt4, "out.1"), (d3, dt3, "out.2"))], D, D, 0)
        # top-down path: t3 = lat2(c3) + up(t4), t4 = lat1(c4) + up(t5)
        dt4s, dt5s = self._blike(t4), self._blike(t5)
        self._ew16(P, K.EW_UPSAMPLE_GRAD, dt3, dt4, dt4s, D)
        self._ew16(P, K.EW_UPSAMPLE_GRAD, dt4s, dt5, dt5s, D)
        # laterals: filter / bias gradients; data gradients meet the stage outputs' other consumers
        dc5, dc4, dc3 = self._blike(c5), self._blike(c4), self._blike(c3)
        for c, dt, dc, name, add in ((c5, dt5s, dc5, "lat.0", dc5a), (c4, dt4s, dc4, "lat.1", None),
                                     (c3, dt3, dc3, "lat.2", None)):
            l = L[name]
            self._wg1(P, c, dt, l)
            self._pw(P, dt, l.pd, dc, l.cout, l.cin, res=add)
        P.mark("bwd_fpn_done")
        grads_into = {5: dc5, 4: dc4, 3: dc3}
        dy, dy_is_dz = None, False
        for (stage, j, cin, cmid, cout, stride, proj, tr) in reversed(self.blocks):
            if not tr:
                break
            pre = "res%d.%d" % (stage, j)
            l1, l2, l3 = L[pre + ".c1"], L[pre + ".c2"], L[pre + ".c3"]
            sv = self.saved[pre]
            x, xs, y1, y2, y = sv["x"], sv["xs"], sv["y1"], sv["y2"], sv["y"]
            if j == ARCHS[self.arch][stage - 2] - 1:
                dy, dy_is_dz = grads_into[stage], False
            if dy_is_dz:
                dz = dy                       # the ReluGradient mask was applied by the GEMM that produced it
            else:
                dz = self._blike(y)
                self._ew16(P, K.EW_RELU_GRAD, y, dy, dz, cout)
            self._wg1(P, y2, dz, l3)
            dz2 = self._blike(y2)
            self._pw(P, dz, l3.pd, dz2, cout, cmid, mask=y2)
            self._wg3(P, y1, dz2, l2)
            dz1 = self._blike(y1)
            self._c3(P, [(dz2, dz1, y1, l2.pd, None)], cmid, cmid, K.CONV_MASK_AUX)
            self._wg1(P, xs, dz1, l1)
            first_trainable = (stage == 3 and j == 0)
            if proj:
                lp = L[pre + ".proj"]
                self._wg1(P, xs, dz, lp)
                if not first_trainable:
                    dxa, dxs = self._blike(xs), self._blike(xs)
                    self._pw(P, dz1, l1.pd, dxa, cmid, cin)
                    self._pw(P, dz, lp.pd, dxs, cout, cin, res=dxa)
                    # into the previous stage's output gradient (which already holds the lateral's part)
                    tgt = grads_into[stage - 1]
                    self._ew16(P, K.EW_SUBSAMPLE_GRAD, dxs, None, tgt, cin, stride=stride, acc=1)
                dy, dy_is_dz = None, False
            else:
                # identity shortcut: dx = dz + W1^T dz1, masked by the previous block's output
                dx = self._blike(dz)
                self._pw(P, dz1, l1.pd, dx, cmid, cin, res=dz, mask=x)
                dy, dy_is_dz = dx, True
            if j == 0:
                P.mark("bwd_res%d_done" % stage)

    # -- running (fp32 NCHW views for tests and callers outside the fp16 domain) ----------------------
    def fpn_f32(self):
        """The five FPN levels as float32 NCHW tensors (a conversion, not part of the step)."""
        return [K.f16_unpack_activations(p, self.D) for p in self.fpn]

    def backward(self, d_fpn=None, scale=1.0):
        """d_fpn: optional float32 NCHW gradients (they are rounded to blocked fp16 times `scale`);
        with heads_io the program sums the subnets' blocked gradients itself."""
        if d_fpn is not None:
            for dst, src in zip(self.d_fpn, d_fpn):
                K.f16_pack_activations(src, scale, out=dst)
        marks = ["backward", "bwd_fpn_done", "bwd_res5_done", "bwd_res4_done", "bwd_res3_done"]
        names = ["fpn", "res5", "res4", "res3"]
        for k in range(4):
            self.prog.run(marks[k], marks[k + 1], timing=self.timing)
            self.dp.issue(self.bucket[names[k]])
