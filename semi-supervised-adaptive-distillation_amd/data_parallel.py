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
            dist.broadcast(t, src=src, group=self.pg)


def shard_images(global_batch, rank, world_size):
    """Contiguous image shard of a global batch: independent units = images."""
    per = global_batch // world_size
    assert per * world_size == global_batch, "global batch must divide evenly"
    return rank * per, (rank + 1) * per
