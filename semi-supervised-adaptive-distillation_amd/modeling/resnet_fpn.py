"""ResNet-FPN body graph builder for RetinaNet (row f1), op for op what the reference emits.

Mirrors detectron/lib/modeling/ResNet.py:47-260 (stem, stages of bottleneck blocks,
stride on the first 1x1 = RESNETS.STRIDE_1X1, frozen-BN AffineChannel after every
conv, in-place Relu / Sum, StopGradient after res2 = TRAIN.FREEZE_AT 2) and
FPN.py:116-250 (lateral 1x1 on res5/res4/res3, UpsampleNearest + Sum top-down path,
3x3 output convs, extra stride-2 levels P6 from res5 and P7 from relu(P6)).
Checked against a capture of the reference builder (tests/golden/
backbone_graph_r50_fpn.json).  Every op runs on this repo's HIP operator surface:
3x3/s1 convolutions on the matrix-core engine, the other geometries on the default
im2col + GEMM engine."""
from dataclasses import dataclass, field

from ..caffe2_hip import core

XAVIER = ("XavierFill", {})


def const_fill(v):
    return ("ConstantFill", {"value": v})


@dataclass
class BodyConfig:
    block_counts: tuple = (3, 4, 6, 3)        # ResNet-50; ResNet-101: (3, 4, 23, 3)
    freeze_at: int = 2                        # TRAIN.FREEZE_AT
    stride_1x1: bool = True                   # RESNETS.STRIDE_1X1
    num_groups: int = 1                       # RESNETS.NUM_GROUPS (ResNeXt: 64)
    width_per_group: int = 64                 # RESNETS.WIDTH_PER_GROUP (ResNeXt-101-64x4d: 4)
    fpn_dim: int = 256                        # FPN.DIM
    k_min: int = 3                            # FPN.RPN_MIN_LEVEL
    k_max: int = 7                            # FPN.RPN_MAX_LEVEL
    use_cudnn_engine: bool = True             # keep the reference's engine="CUDNN" argument


@dataclass
class BodyModel:
    cfg: BodyConfig
    name: str = "resnet_fpn"
    net: core.Net = None
    params: list = field(default_factory=list)      # (name, shape, (filler, kwargs))

    def __post_init__(self):
        if self.net is None:
            self.net = core.Net(self.name)

    def _engine(self, kw):
        if self.cfg.use_cudnn_engine:
            kw["engine"] = "CUDNN"
        return kw

    def Conv(self, blob_in, blob_out, dim_in, dim_out, kernel, weight_init=None, bias_init=None,
             no_bias=0, group=1, **kw):
        """CNNModelHelper.Conv (caffe2/python/helpers/conv.py:28-149): the filter of a grouped
        convolution is [dim_out, dim_in / group, k, k]; `group` is an argument only when != 1."""
        kw = self._engine(kw)
        if self.cfg.use_cudnn_engine:
            kw["exhaustive_search"] = False
        if group != 1:
            kw["group"] = group
        w = blob_out + "_w"
        self.params.append((w, [dim_out, dim_in // group, kernel, kernel], weight_init or XAVIER))
        ins = [blob_in, w]
        if not no_bias:
            b = blob_out + "_b"
            self.params.append((b, [dim_out], bias_init or ("ConstantFill", {})))
            ins.append(b)
        return self.net.Conv(ins, blob_out, kernel=kernel, order="NCHW", **kw)

    def AffineChannel(self, blob_in, blob_out, dim, inplace=False):
        """DetectionModelHelper.AffineChannel (detector.py:83-107)."""
        s, b = blob_out + "_s", blob_out + "_b"
        self.params.append((s, [dim], const_fill(1.0)))
        self.params.append((b, [dim], const_fill(0.0)))
        return self.net.AffineChannel([blob_in, s, b], blob_in if inplace else blob_out)

    def ConvAffine(self, blob_in, prefix, dim_in, dim_out, kernel, stride, pad, group=1, dilation=1,
                   suffix="_bn", inplace=False):
        """detector.py:559-587."""
        c = self.Conv(blob_in, prefix, dim_in, dim_out, kernel, stride=stride, pad=pad, group=group,
                      dilation=dilation, no_bias=1)
        return self.AffineChannel(c, prefix + suffix, dim=dim_out, inplace=inplace)

    def Relu(self, blob_in, blob_out):
        return self.net.Relu(blob_in, blob_out)

    def MaxPool(self, blob_in, blob_out, **kw):
        return self.net.MaxPool(blob_in, blob_out, order="NCHW", **self._engine(kw))

    def StopGradient(self, blob_in, blob_out):
        return self.net.StopGradient(blob_in, blob_out)


# ---- ResNet.py ---------------------------------------------------------------------

def bottleneck_transformation(model, blob_in, dim_in, dim_out, stride, prefix, dim_inner,
                              dilation=1, group=1):
    """ResNet.py:223-283."""
    str1x1, str3x3 = (stride, 1) if model.cfg.stride_1x1 else (1, stride)
    cur = model.ConvAffine(blob_in, prefix + "_branch2a", dim_in, dim_inner, kernel=1,
                           stride=str1x1, pad=0, inplace=True)
    cur = model.Relu(cur, cur)
    cur = model.ConvAffine(cur, prefix + "_branch2b", dim_inner, dim_inner, kernel=3,
                           stride=str3x3, pad=1 * dilation, dilation=dilation, group=group,
                           inplace=True)
    cur = model.Relu(cur, cur)
    return model.ConvAffine(cur, prefix + "_branch2c", dim_inner, dim_out, kernel=1, stride=1,
                            pad=0, inplace=False)


def add_shortcut(model, prefix, blob_in, dim_in, dim_out, stride):
    """ResNet.py:199-213."""
    if dim_in == dim_out:
        return blob_in
    c = model.Conv(blob_in, prefix + "_branch1", dim_in, dim_out, kernel=1, stride=stride,
                   no_bias=1)
    return model.AffineChannel(c, prefix + "_branch1_bn", dim=dim_out)


def add_residual_block(model, prefix, blob_in, dim_in, dim_out, dim_inner, dilation,
                       stride_init=2, inplace_sum=False):
    """ResNet.py:158-197."""
    stride = stride_init if (dim_in != dim_out and dim_in != 64 and dilation == 1) else 1
    tr = bottleneck_transformation(model, blob_in, dim_in, dim_out, stride, prefix, dim_inner,
                                   group=model.cfg.num_groups, dilation=dilation)
    sc = add_shortcut(model, prefix, blob_in, dim_in, dim_out, stride)
    s = model.net.Sum([tr, sc], tr if inplace_sum else prefix + "_sum")
    return model.Relu(s, s)


def add_stage(model, prefix, blob_in, n, dim_in, dim_out, dim_inner, dilation, stride_init=2):
    """ResNet.py:57-86."""
    for i in range(n):
        blob_in = add_residual_block(model, "%s_%d" % (prefix, i), blob_in, dim_in, dim_out,
                                     dim_inner, dilation, stride_init, inplace_sum=i < n - 1)
        dim_in = dim_out
    return blob_in, dim_in


def add_resnet_conv5_body(model):
    """add_ResNet_convX_body (ResNet.py:88-131) with four stages."""
    cfg = model.cfg
    p = model.Conv("data", "conv1", 3, 64, 7, pad=3, stride=2, no_bias=1)
    p = model.AffineChannel(p, "res_conv1_bn", dim=64, inplace=True)
    p = model.Relu(p, p)
    p = model.MaxPool(p, "pool1", kernel=3, pad=1, stride=2)
    dim_in, dim_bottleneck = 64, cfg.num_groups * cfg.width_per_group      # ResNet.py:99
    n1, n2, n3, n4 = cfg.block_counts
    stage_out = {}
    s, dim_in = add_stage(model, "res2", p, n1, dim_in, 256, dim_bottleneck, 1)
    if cfg.freeze_at == 2:
        model.StopGradient(s, s)
    stage_out[2] = (s, dim_in)
    s, dim_in = add_stage(model, "res3", s, n2, dim_in, 512, dim_bottleneck * 2, 1)
    if cfg.freeze_at == 3:
        model.StopGradient(s, s)
    stage_out[3] = (s, dim_in)
    s, dim_in = add_stage(model, "res4", s, n3, dim_in, 1024, dim_bottleneck * 4, 1)
    if cfg.freeze_at == 4:
        model.StopGradient(s, s)
    stage_out[4] = (s, dim_in)
    s, dim_in = add_stage(model, "res5", s, n4, dim_in, 2048, dim_bottleneck * 8, 1)
    if cfg.freeze_at == 5:
        model.StopGradient(s, s)
    stage_out[5] = (s, dim_in)
    return stage_out


# ---- FPN.py -------------------------------------------------------------------------

def add_topdown_lateral_module(model, fpn_top, fpn_lateral, fpn_bottom, dim_top, dim_lateral):
    """FPN.py:262-288."""
    lat = model.Conv(fpn_lateral, fpn_bottom + "_lateral", dim_in=dim_lateral, dim_out=dim_top,
                     kernel=1, pad=0, stride=1, weight_init=XAVIER, bias_init=const_fill(0.0))
    td = model.net.UpsampleNearest(fpn_top, fpn_bottom + "_topdown", scale=2)
    model.net.Sum([lat, td], fpn_bottom)


def add_fpn_resnet_conv5_body(model):
    """add_fpn_ResNet50/101_conv5_body for RetinaNet (FPN.py:60-113 + add_fpn :116-250).
    Returns (blobs coarsest first, dim, spatial scales) like the reference."""
    cfg = model.cfg
    stages = add_resnet_conv5_body(model)
    HIGHEST_BACKBONE_LVL = 5
    levels = list(range(HIGHEST_BACKBONE_LVL, cfg.k_min - 1, -1))         # 5, 4, 3
    lateral_in = [stages[l][0] for l in levels]
    dims = [stages[l][1] for l in levels]
    out_blobs = ["fpn_inner_%s" % b for b in lateral_in]
    fpn_dim = cfg.fpn_dim
    model.Conv(lateral_in[0], out_blobs[0], dim_in=dims[0], dim_out=fpn_dim, kernel=1, pad=0,
               stride=1, weight_init=XAVIER, bias_init=const_fill(0.0))
    for i in range(len(levels) - 1):
        add_topdown_lateral_module(model, out_blobs[i], lateral_in[i + 1], out_blobs[i + 1],
                                   fpn_dim, dims[i + 1])
    blobs_fpn, scales = [], []
    for i, l in enumerate(levels):
        blobs_fpn.append(model.Conv(out_blobs[i], "fpn_%s" % lateral_in[i], dim_in=fpn_dim,
                                    dim_out=fpn_dim, kernel=3, pad=1, stride=1, weight_init=XAVIER,
                                    bias_init=const_fill(0.0)))
        scales.append(1.0 / 2 ** l)
    fpn_blob, dim_in = lateral_in[0], dims[0]
    for i in range(HIGHEST_BACKBONE_LVL + 1, cfg.k_max + 1):               # FPN.EXTRA_CONV_LEVELS
        fpn_blob_in = fpn_blob
        if i > HIGHEST_BACKBONE_LVL + 1:
            fpn_blob_in = model.Relu(fpn_blob, fpn_blob + "_relu")
        fpn_blob = model.Conv(fpn_blob_in, "fpn_%d" % i, dim_in=dim_in, dim_out=fpn_dim, kernel=3,
                              pad=1, stride=2, weight_init=XAVIER, bias_init=const_fill(0.0))
        dim_in = fpn_dim
        blobs_fpn.insert(0, fpn_blob)
        scales.insert(0, scales[0] * 0.5)
    return blobs_fpn, fpn_dim, scales
