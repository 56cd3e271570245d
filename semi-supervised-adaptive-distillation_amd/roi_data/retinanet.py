"""RetinaNet training blobs computed on the GPU (row f4): the device replacement of
`add_retinanet_blobs` (detectron/lib/roi_data/retinanet.py:97-196)."""
import ctypes as C

import numpy as np
import torch

from .. import kernels as K
from ..modeling.generate_anchors import AnchorConfig, cell_anchors, field_sizes


class RetinanetLabeler(object):
    """Pre-allocates everything once; `__call__` labels one minibatch.

    gt_boxes  float32 [N][Gmax][4] device (x1, y1, x2, y2 in blob pixels, already scaled)
    gt_classes int32  [N][Gmax] device, gt_counts int32 [N] device
    Returns a dict with the reference's blob names (labels / locs / targets per
    level, `retnet_fg_num`, `retnet_bg_num`); the list tensors are views of
    length M (one host sync to read the five counts)."""

    def __init__(self, N, Gmax, im_height, im_width, cfg=AnchorConfig, capacity=1 << 16,
                 device="cuda"):
        self.cfg, self.N, self.Gmax, self.capacity = cfg, N, Gmax, capacity
        self.levels = cfg.k_max - cfg.k_min + 1
        self.A = cfg.scales_per_octave * len(cfg.aspect_ratios)
        self.fs = field_sizes(cfg)
        self.h = [int(im_height / float(2 ** l)) for l in range(cfg.k_min, cfg.k_max + 1)]
        self.w = [int(im_width / float(2 ** l)) for l in range(cfg.k_min, cfg.k_max + 1)]
        self.cells = torch.as_tensor(cell_anchors(cfg), dtype=torch.float64, device=device)
        self.labels = [torch.empty((N, self.A, h, w), dtype=torch.int32, device=device)
                       for h, w in zip(self.h, self.w)]
        self.locs = [torch.empty((capacity, 4), dtype=torch.float32, device=device)
                     for _ in range(self.levels)]
        self.targets = [torch.empty((capacity, 4), dtype=torch.float32, device=device)
                        for _ in range(self.levels)]
        self.counts = torch.zeros(self.levels, dtype=torch.int32, device=device)
        self.fg_bg = torch.zeros(2, dtype=torch.float32, device=device)
        L = K.lib()
        IntArr = C.c_int * self.levels
        self._fs, self._h, self._w = IntArr(*self.fs), IntArr(*self.h), IntArr(*self.w)
        L.ssad_retinanet_anchor_labels_workspace_bytes.restype = C.c_size_t
        nb = L.ssad_retinanet_anchor_labels_workspace_bytes(self.levels, self.A, cfg.k_min,
                                                            self._fs, N, Gmax)
        self.ws = torch.empty(max(int(nb), 1), dtype=torch.uint8, device=device)
        PtrArr = C.c_void_p * self.levels
        self._labels = PtrArr(*[t.data_ptr() for t in self.labels])
        self._locs = PtrArr(*[t.data_ptr() for t in self.locs])
        self._targets = PtrArr(*[t.data_ptr() for t in self.targets])

    def __call__(self, gt_boxes, gt_classes, gt_counts):
        cfg = self.cfg
        for t, dt in ((gt_boxes, torch.float32), (gt_classes, torch.int32), (gt_counts, torch.int32)):
            if t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
                raise K.KernelError("ground truth must be contiguous device tensors (f32, i32, i32)")
        if tuple(gt_boxes.shape) != (self.N, self.Gmax, 4):
            raise K.KernelError("gt_boxes must be N x Gmax x 4")
        rc = K.lib().ssad_retinanet_anchor_labels(
            C.c_void_p(self.cells.data_ptr()), self.levels, self.A, cfg.k_min, self._fs, self._h,
            self._w, C.c_void_p(gt_boxes.data_ptr()), C.c_void_p(gt_classes.data_ptr()),
            C.c_void_p(gt_counts.data_ptr()), self.N, self.Gmax, cfg.num_classes,
            C.c_float(cfg.positive_overlap), C.c_float(cfg.negative_overlap), self._labels,
            self._locs, self._targets, self.capacity, C.c_void_p(self.counts.data_ptr()),
            C.c_void_p(self.fg_bg.data_ptr()), C.c_void_p(self.ws.data_ptr()),
            C.c_size_t(self.ws.numel()), K._stream())
        if rc:
            raise K.KernelError("retinanet_anchor_labels failed (%d)" % rc)
        counts = self.counts.cpu().numpy()
        if int(counts.max(initial=0)) > self.capacity:
            raise K.KernelError("foreground list capacity %d exceeded (%d)" %
                                (self.capacity, int(counts.max())))
        blobs = {"retnet_fg_num": self.fg_bg[0:1], "retnet_bg_num": self.fg_bg[1:2]}
        for i, lvl in enumerate(range(cfg.k_min, cfg.k_max + 1)):
            m = int(counts[i])
            blobs["retnet_cls_labels_fpn%d" % lvl] = self.labels[i]
            blobs["retnet_roi_fg_bbox_locs_fpn%d" % lvl] = self.locs[i][:m]
            blobs["retnet_roi_bbox_targets_fpn%d" % lvl] = self.targets[i][:m]
        return blobs


class RetinanetDetector(object):
    """Device replacement of `im_detect_bbox`'s post-processing
    (detectron/lib/core/test_retinanet.py:108-206) for one image.

    `__call__(cls_probs, box_preds, im_height, im_width, im_scale)` takes the per-level
    sigmoid scores [1][A*C][H][W] and box deltas [1][A*4][H][W] (device tensors) and
    returns float32 [n <= dets_per_im][6] = x1, y1, x2, y2, score, class."""

    def __init__(self, level_shapes, cfg=AnchorConfig, inference_th=0.05, pre_nms_topn=1000,
                 nms_thresh=0.5, dets_per_im=100, device="cuda"):
        self.cfg = cfg
        self.levels = len(level_shapes)
        self.A = cfg.scales_per_octave * len(cfg.aspect_ratios)
        self.C = cfg.num_classes - 1
        self.th, self.topn, self.nms, self.keep = inference_th, pre_nms_topn, nms_thresh, dets_per_im
        self.cells = torch.as_tensor(cell_anchors(cfg)[:self.levels], dtype=torch.float64,
                                     device=device).contiguous()
        IntArr = C.c_int * self.levels
        self._H = IntArr(*[h for h, _ in level_shapes])
        self._W = IntArr(*[w for _, w in level_shapes])
        self.shapes = list(level_shapes)
        L = K.lib()
        L.ssad_retinanet_detect_workspace_bytes.restype = C.c_size_t
        nb = L.ssad_retinanet_detect_workspace_bytes(self.levels, self.A, self.C, self._H, self._W,
                                                     self.topn)
        if nb == 0:
            raise K.KernelError("retinanet_detect: unsupported geometry")
        self.ws = torch.empty(int(nb), dtype=torch.uint8, device=device)
        self.out = torch.zeros((dets_per_im, 6), dtype=torch.float32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)

    def __call__(self, cls_probs, box_preds, im_height, im_width, im_scale):
        for t, (h, w) in zip(cls_probs, self.shapes):
            if tuple(t.shape) != (1, self.A * self.C, h, w) or t.dtype != torch.float32 or \
                    not t.is_contiguous() or not t.is_cuda:
                raise K.KernelError("cls_prob must be 1 x A*C x H x W contiguous float32 device tensors")
        for t, (h, w) in zip(box_preds, self.shapes):
            if tuple(t.shape) != (1, self.A * 4, h, w) or t.dtype != torch.float32 or \
                    not t.is_contiguous() or not t.is_cuda:
                raise K.KernelError("box_pred must be 1 x A*4 x H x W contiguous float32 device tensors")
        PtrArr = C.c_void_p * self.levels
        rc = K.lib().ssad_retinanet_detect(
            PtrArr(*[t.data_ptr() for t in cls_probs]), PtrArr(*[t.data_ptr() for t in box_preds]),
            C.c_void_p(self.cells.data_ptr()), self.levels, self.A, self.C, self.cfg.k_min, self._H,
            self._W, C.c_float(self.th), self.topn, C.c_float(self.nms), self.keep,
            C.c_float(im_scale), int(im_height), int(im_width),
            C.c_float(float(np.log(1000. / 16.))), C.c_void_p(self.out.data_ptr()),
            C.c_void_p(self.count.data_ptr()), C.c_void_p(self.ws.data_ptr()),
            C.c_size_t(self.ws.numel()), K._stream())
        if rc:
            raise K.KernelError("retinanet_detect failed (%d)" % rc)
        return self.out[:int(self.count.item())]
