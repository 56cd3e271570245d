"""Independent float64 statement of the ResNet-FPN backbone for the GPU parity tests: plain
torch.nn.functional convolutions, no kernel of this repo.

Network: detectron/lib/modeling/ResNet.py:85-130,221-283 (bottleneck stages 3-4-{6,23}-3; frozen
BN = AffineChannel folded into the preceding convolution, W' = s W, bias = b; the stride on the
first 1x1, or on the grouped 3x3 for ResNeXt) and FPN.py:116-250 (laterals, nearest top-down Sum,
3x3 output convolutions, P6 = conv3x3/2(res5), P7 = conv3x3/2(relu(P6))).  Layer names are those
of ssad_amd.backbone_pipeline.NativeResNetFPN (`state_dict()` feeds its `src=`).

`calibrate()` picks the folded biases so that no ReLU mask of the trainable part can flip
between two correct implementations: per channel the pre-activation is shifted to be either
everywhere >= margin or everywhere <= -margin (alternating channels), the arrangement
tests/test_gpu_operators.py:make_mask_safe uses for the subnets.
"""
import numpy as np
import torch
import torch.nn.functional as F

ARCHS = {   # name -> (block counts, groups, width per group, stride on the 1x1)
    "r50": ((3, 4, 6, 3), 1, 64, True),
    "r101": ((3, 4, 23, 3), 1, 64, True),
    "x101-64x4d": ((3, 4, 23, 3), 64, 4, False),
}


class RefResNetFPN(object):
    def __init__(self, arch="r50", fpn_dim=256, seed=11, device="cuda", bias_std=0.05, c3_scale=0.25):
        self.arch, self.D, self.device = arch, fpn_dim, device
        blocks, groups, width, s1x1 = ARCHS[arch]
        self.blocks_per_stage, self.groups, self.s1x1 = blocks, groups, s1x1
        gen = torch.Generator().manual_seed(seed)
        self.p = {}              # name.weight / name.bias -> float64 tensor (requires_grad on trainable)
        self.scales = {}         # folded affine scale per layer (the c3 damping)
        self.spec = []           # (prefix, cin, cmid, cout, stride, has_proj, trainable)

        def add(name, cout, cin_g, k, train, he=True, scale=1.0):
            fan_in = cin_g * k * k
            if he:
                w = torch.randn((cout, cin_g, k, k), generator=gen, dtype=torch.float64) * np.sqrt(2.0 / fan_in) * scale
            else:
                bound = np.sqrt(6.0 / (fan_in + cout * k * k))
                w = (torch.rand((cout, cin_g, k, k), generator=gen, dtype=torch.float64) * 2 - 1) * bound
            b = torch.randn((cout,), generator=gen, dtype=torch.float64) * bias_std
            self.p[name + ".weight"] = w.to(device).requires_grad_(train)
            # the body's folded biases are frozen values; the FPN's own biases are parameters
            self.p[name + ".bias"] = b.to(device).requires_grad_(train and not name.startswith(("stem", "res")))
            if scale != 1.0:
                self.scales[name] = scale

        add("stem.0", 64, 3, 7, False)
        cin = 64
        for si, n in enumerate(blocks):
            stage = si + 2
            cmid, cout = groups * width * 2 ** si, 256 * 2 ** si
            tr = stage > 2
            for j in range(n):
                stride = 2 if (j == 0 and si > 0) else 1
                pre = "res%d.%d" % (stage, j)
                add(pre + ".c1", cmid, cin, 1, tr)
                add(pre + ".c2", cmid, cmid // groups, 3, tr)
                add(pre + ".c3", cout, cmid, 1, tr, scale=c3_scale)
                proj = cin != cout or stride != 1
                if proj:
                    add(pre + ".proj", cout, cin, 1, tr)
                self.spec.append((pre, cin, cmid, cout, stride, proj, tr))
                cin = cout
        for i, c in enumerate((2048, 1024, 512)):
            add("lat.%d" % i, fpn_dim, c, 1, True, he=False)
        for i in range(3):
            add("out.%d" % i, fpn_dim, fpn_dim, 3, True, he=False)
        add("p6", fpn_dim, 2048, 3, True, he=False)
        add("p7", fpn_dim, fpn_dim, 3, True, he=False)

    def state_dict(self):
        return {k: v.detach() for k, v in self.p.items()}

    def named_parameters(self):
        return [(k, v) for k, v in self.p.items() if v.requires_grad]

    def zero_grad(self):
        for v in self.p.values():
            v.grad = None

    # -- forward --------------------------------------------------------------------------
    def _act(self, z, name, calibrate, margin):
        """relu(z + bias); with calibrate, first choose the bias so that every channel of z + bias
        is >= margin * range everywhere (even channels) or <= -margin * range (odd channels)."""
        b = self.p[name + ".bias"]
        if calibrate:
            with torch.no_grad():
                zz = z.detach()
                lo = zz.amin(dim=(0, 2, 3))
                hi = zz.amax(dim=(0, 2, 3))
                d = margin * (hi - lo).clamp_min(1e-6)
                ch = torch.arange(zz.shape[1], device=zz.device)
                b.copy_(torch.where(ch % 2 == 0, -lo + d, -hi - d))
        return F.relu(z + b.view(1, -1, 1, 1))

    def forward(self, x, calibrate=False, margin=0.03):
        p = self.p
        x = x.to(torch.float64)
        z = F.conv2d(x, p["stem.0.weight"], p["stem.0.bias"], 2, 3)
        y = F.max_pool2d(F.relu(z), 3, 2, 1)
        outs = {}
        k = 0
        for si, n in enumerate(self.blocks_per_stage):
            for j in range(n):
                pre, cin, cmid, cout, stride, proj, tr = self.spec[k]
                k += 1
                s1, s3 = (stride, 1) if self.s1x1 else (1, stride)
                cal = calibrate and tr
                y1 = self._act(F.conv2d(y, p[pre + ".c1.weight"], None, s1), pre + ".c1", cal, margin)
                y2 = self._act(F.conv2d(y1, p[pre + ".c2.weight"], None, s3, 1, 1, self.groups), pre + ".c2", cal,
                               margin)
                sc = y
                if proj:
                    sc = F.conv2d(y, p[pre + ".proj.weight"], p[pre + ".proj.bias"], stride)
                y = self._act(F.conv2d(y2, p[pre + ".c3.weight"], None) + sc, pre + ".c3", cal, margin)
            outs[si + 2] = y
        c3, c4, c5 = outs[3], outs[4], outs[5]
        t5 = F.conv2d(c5, p["lat.0.weight"], p["lat.0.bias"])
        t4 = F.conv2d(c4, p["lat.1.weight"], p["lat.1.bias"]) + F.interpolate(t5, scale_factor=2, mode="nearest")
        t3 = F.conv2d(c3, p["lat.2.weight"], p["lat.2.bias"]) + F.interpolate(t4, scale_factor=2, mode="nearest")
        p5 = F.conv2d(t5, p["out.0.weight"], p["out.0.bias"], 1, 1)
        p4 = F.conv2d(t4, p["out.1.weight"], p["out.1.bias"], 1, 1)
        p3 = F.conv2d(t3, p["out.2.weight"], p["out.2.bias"], 1, 1)
        z6 = F.conv2d(c5, p["p6.weight"], None, 2, 1)
        if calibrate:
            r6 = self._act(z6, "p6", True, margin)          # P7 sees relu(P6): its mask must be safe too
        else:
            r6 = F.relu(z6 + p["p6.bias"].view(1, -1, 1, 1))
        p6 = z6 + p["p6.bias"].view(1, -1, 1, 1)
        p7 = F.conv2d(r6, p["p7.weight"], p["p7.bias"], 2, 1)
        return [p3, p4, p5, p6, p7]

    __call__ = forward

    def calibrate(self, images, margin=0.03):
        with torch.no_grad():
            self.forward(images, calibrate=True, margin=margin)
        return self
