"""Data-parallel exchange for the hot path: one process per GPU, replicated
parameters, images sharded across ranks, ONE exchange step per iteration --
the sum all-reduce of parameter gradients -- as in the reference
(detectron/lib/modeling/optimizer.py:33-130: per-GPU loss scaled by
1/NUM_GPUS, per-parameter NCCLAllreduce or muji tree, identical local SGD).

MI355X-first differences:
  * process-per-GPU with torch.distributed (backend "nccl" = RCCL over xGMI)
    instead of the reference's single-process ncclCommInitAll;
  * gradients live in flat buckets, so the exchange is a few large
    all-reduces (xGMI is 7 point-to-point links per GPU: large messages, few
    launches) instead of one small collective per parameter blob;
  * each bucket's all-reduce is issued asynchronously as soon as its last
    weight gradient has been enqueued and overlaps the rest of backward; the
    reference syncs the stream after every operator and cannot overlap.

Backend-agnostic: the same code runs under "gloo" on CPU tensors (tests).
"""
import torch


class BucketedAllReduce(object):
    issued_total = 0        # collectives started by every instance of this process (bench.py reports the rate)

    def __init__(self, process_group=None, world_size=1):
        self.pg = process_group
        self.world_size = world_size
        self._pending = []

    @property
    def active(self):
        # SSAD_DP_FORCE=1 runs the collectives even on a 1-rank group (used to
        # exercise the RCCL path on a single-GPU box)
        import os
        return self.pg is not None and (self.world_size > 1 or os.environ.get("SSAD_DP_FORCE") == "1")

    def issue(self, bucket):
        """Start the sum all-reduce of one flat gradient bucket (in place)."""
        if not self.active:
            return
        import torch.distributed as dist
        # On GPU the collective runs on RCCL's own stream; torch orders it after
        # the kernels already enqueued on the current stream (the HIP kernels
        # of this package launch on torch's current stream) and work.wait()
        # orders later kernels after it.
        self._pending.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg,
                                             async_op=True))
        BucketedAllReduce.issued_total += 1

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def broadcast(self, tensors, src=0):
        """Initial parameter sync from rank `src`
        (detectron/lib/utils/net.py:185-208 broadcasts gpu0 -> all)."""
        if not self.active:
            return
        import torch.distributed as dist
        for t in tensors:
            if t.numel():
                dist.broadcast(t, src=src, group=self.pg)


class GradBuckets(object):
    """Flat gradient buckets for autograd-owned parameters (the backbone of the full
    model): `groups` lists the parameters bucket by bucket in the order the backward pass
    finishes them (FPN, res5, res4, res3 -- cf. SURVEY 8e "backbone 4-6 buckets").  A
    post-accumulate hook counts a bucket down; the moment its last gradient exists the
    bucket is gathered into its slice of one flat buffer (one multi-tensor copy), its
    all-reduce is started -- so the exchange of the late stages overlaps the backward pass
    of the early ones -- and every .grad of the bucket is pointed at its view of the
    reduced buffer for the optimizer.  Autograd itself writes fresh gradients (no
    per-parameter accumulate kernels, no buffer clear)."""

    def __init__(self, groups, dp):
        self.dp = dp
        self.groups = [list(g) for g in groups if len(g)]
        first = self.groups[0][0]
        n = sum(p.numel() for g in self.groups for p in g)
        self.flat = torch.zeros(n, device=first.device, dtype=first.dtype)
        self.slices, self.views, off = [], [], 0
        for gi, g in enumerate(self.groups):
            start, vs = off, []
            for p in g:
                vs.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                p.register_post_accumulate_grad_hook(self._hook(gi))
            self.views.append(vs)
            self.slices.append(self.flat[start:off])
        self.sizes = [len(g) for g in self.groups]
        self.begin()

    def _hook(self, gi):
        def ready(_param):
            self._left[gi] -= 1
            if self._left[gi] == 0:
                self._exchange(gi)
        return ready

    def _exchange(self, gi):
        if self._issued[gi]:
            return
        self._issued[gi] = True
        with torch.no_grad():
            have = [(v, p.grad) for v, p in zip(self.views[gi], self.groups[gi]) if p.grad is not None]
            if len(have) < len(self.groups[gi]):
                self.slices[gi].zero_()            # a parameter without gradient contributes 0
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            for v, p in zip(self.views[gi], self.groups[gi]):
                p.grad = v
        self.dp.issue(self.slices[gi])

    def begin(self):
        """Before backward: drop last step's gradients so autograd assigns, not accumulates."""
        for g in self.groups:
            for p in g:
                p.grad = None
        self._left = list(self.sizes)
        self._issued = [False] * len(self.sizes)

    def finish(self):
        """After backward: exchange whatever a hook did not (a bucket with a parameter that
        received no gradient this step), then wait for all of them."""
        for gi in range(len(self.groups)):
            self._exchange(gi)
        self.dp.wait()


def shard_images(global_batch, rank, world_size):
    """Contiguous image shard of a global batch: independent units = images."""
    per = global_batch // world_size
    assert per * world_size == global_batch, "global batch must divide evenly"
    return rank * per, (rank + 1) * per
