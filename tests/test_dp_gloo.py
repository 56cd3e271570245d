"""The N>1 path on CPU: world_size-2 gloo processes run the same bucketed
gradient exchange the GPU path uses (ssad_amd.data_parallel), on flat
parameter buckets laid out by head_pipeline.FlatParams, followed by the SGD
update -- and must end bitwise identical on both ranks and equal to the
single-process sum (cf. caffe2/contrib/nccl/nccl_ops_test.py:56-80)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ssad_amd  # noqa: F401
from oracle import oracle
from ssad_amd.data_parallel import BucketedAllReduce, shard_images
from ssad_amd.modeling.retinanet_heads import HeadConfig


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssad_amd.head_pipeline import FlatParams
    cfg = HeadConfig(num_convs=2, fpn_dim=8, aspect_ratios=(1.0,), scales_per_octave=1,
                     num_classes=4, num_gpus=world)
    params, grads, moms = (FlatParams(cfg, "cpu") for _ in range(3))
    g0 = torch.Generator().manual_seed(100)            # same initial params on rank 0 only
    if rank == 0:
        params.flat.copy_(torch.randn(params.flat.shape, generator=g0))
    dp = BucketedAllReduce(dist.group.WORLD, world)
    dp.broadcast([params.flat, moms.flat], src=0)
    gr = torch.Generator().manual_seed(7 + rank)       # rank-local gradients
    grads.flat.copy_(torch.randn(grads.flat.shape, generator=gr))
    local = grads.flat.clone()
    assert set(grads.bucket) <= {"late", "early"} and grads.bucket
    for name in grads.bucket:                          # one all-reduce per bucket
        dp.issue(grads.bucket[name])
    dp.wait()
    reduced = grads.flat.clone()                       # before the SGD op rewrites the buffer
    # identical SGD on every rank (optimizer.py:95-130)
    for name, _, is_bias, _ in params.specs:
        w, g, m = oracle.sgd_update(params[name].numpy(), grads[name].numpy(), moms[name].numpy(),
                                    0.01, 0.9, 1e-4, is_bias)
        params[name].copy_(torch.from_numpy(w))
        moms[name].copy_(torch.from_numpy(m))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), params=params.flat.numpy(),
             reduced=reduced.numpy(), local=local.numpy(), moms=moms.flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_allreduce_and_update():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [np.load(os.path.join(d, "rank%d.npz" % i)) for i in range(world)]
    # bitwise identical across ranks after the exchange and after the update
    assert np.array_equal(r[0]["params"], r[1]["params"])
    assert np.array_equal(r[0]["moms"], r[1]["moms"])
    # the reduced gradient is the sum of the rank-local gradients, on BOTH ranks, bit for bit
    # (two fp32 addends: the sum does not depend on the order; nccl_ops_test.py:77-79)
    total = r[0]["local"] + r[1]["local"]
    assert not np.array_equal(r[0]["local"], r[1]["local"]) and np.any(total != 0)
    for k in range(world):
        assert np.array_equal(r[k]["reduced"], total)
    # and the update is the oracle's SGD of that sum on the broadcast parameters
    assert np.all(np.isfinite(r[0]["params"]))


def test_image_sharding():
    assert shard_images(32, 0, 2) == (0, 16) and shard_images(32, 1, 2) == (16, 32)
    assert [shard_images(128, r, 8) for r in (0, 7)] == [(0, 16), (112, 128)]
    with pytest.raises(AssertionError):
        shard_images(10, 0, 4)


def _bucket_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssad_amd.data_parallel import GradBuckets
    torch.manual_seed(3)                                   # same replica on every rank
    stages = [torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.Tanh()) for _ in range(4)]
    unused = torch.nn.Linear(3, 3)                         # a bucket no gradient ever reaches
    order = []

    class Spy(BucketedAllReduce):
        def issue(self, bucket):
            order.append(int(bucket.numel()) * 1000 + int(bucket.data_ptr() % 997))
            return super().issue(bucket)

    dp = Spy(dist.group.WORLD, world)
    # backward finishes the LAST stage first
    groups = [list(s.parameters()) for s in reversed(stages)] + [list(unused.parameters())]
    gb = GradBuckets(groups, dp)
    out = {}
    for step in range(2):                                  # hooks must re-arm every step
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * step + rank))
        gb.begin()
        issued_before = len(order)
        y = x
        for s in stages:
            y = s(y)
        y.square().sum().backward()
        issued_in_backward = len(order) - issued_before
        gb.finish()
        assert issued_in_backward == 4                     # one per stage, from the hooks
        assert len(order) - issued_before == 5             # + the unused bucket, in finish()
        out["flat%d" % step] = gb.flat.clone().numpy()
        # the rank-local gradient of the same step, for the sum check
        loc = []
        for g in groups[:-1]:
            for p in g:
                loc.append(torch.autograd.grad(
                    _replay(stages, x), p, retain_graph=False)[0].reshape(-1))
        out["local%d" % step] = torch.cat(loc).numpy()
    for s in stages:                                       # .grad stayed a view of the flat buffer
        for p in s.parameters():
            assert p.grad.data_ptr() >= gb.flat.data_ptr()
            assert p.grad.data_ptr() < gb.flat.data_ptr() + gb.flat.numel() * 4
    np.savez(os.path.join(outdir, "b%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def _replay(stages, x):
    y = x
    for s in stages:
        y = s(y)
    return y.square().sum()


def test_two_rank_hook_driven_gradient_buckets():
    """GradBuckets (the backbone's exchange in the full model): every bucket's all-reduce is
    started by the hook of its last gradient during backward, buckets re-arm each step, a
    bucket without gradients is still exchanged, and both ranks end with the bitwise
    identical sum of the rank-local gradients."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_bucket_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [np.load(os.path.join(d, "b%d.npz" % i)) for i in range(world)]
    for step in range(2):
        k = "flat%d" % step
        assert np.array_equal(r[0][k], r[1][k])
        n = r[0]["local%d" % step].size
        want = r[0]["local%d" % step] + r[1]["local%d" % step]
        assert np.allclose(r[0][k][:n], want, rtol=1e-6, atol=1e-7)
        assert np.all(r[0][k][n:] == 0) and np.any(want != 0)
    assert not np.array_equal(r[0]["flat0"], r[0]["flat1"])


# ---------------------------------------------------------------------------
# the backbone's four gradient buckets (backbone_pipeline.NativeResNetFPN) under gloo, world 2
# ---------------------------------------------------------------------------

def _backbone_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    torch.manual_seed(1234 + rank)                    # only the broadcast may make the ranks agree
    bb = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True, lr=0.01, process_group=dist.group.WORLD,
                         world_size=world)
    if rank == 1:
        bb.params_flat.add_(1.0)                      # diverge on purpose
        bb.moms_flat.fill_(3.0)
    bb.broadcast_params()
    gr = torch.Generator().manual_seed(70 + rank)
    bb.grads_flat.copy_(torch.randn(bb.grads_flat.shape, generator=gr))
    local = bb.grads_flat.clone()
    # as NativeResNetFPN.backward() does: one asynchronous all-reduce per stage bucket, in the order the
    # backward pass completes them, then the wait in front of the SGD launch
    names = list(bb.bucket)
    for n in names:
        bb.dp.issue(bb.bucket[n])
    bb.dp.wait()
    np.savez(os.path.join(outdir, "bb%d.npz" % rank), params=bb.params_flat.numpy(), moms=bb.moms_flat.numpy(),
             local=local.numpy(), reduced=bb.grads_flat.numpy(),
             sizes=np.array([bb.bucket[n].numel() for n in names]), names=np.array(names))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_backbone_buckets_cover_every_gradient_exactly_once():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_backbone_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [np.load(os.path.join(d, "bb%d.npz" % i)) for i in range(world)]
    assert list(r[0]["names"]) == ["fpn", "res5", "res4", "res3"]
    assert int(r[0]["sizes"].sum()) == r[0]["local"].size           # the buckets tile the flat buffer
    assert np.array_equal(r[0]["params"], r[1]["params"]) and np.array_equal(r[0]["moms"], r[1]["moms"])
    assert float(np.abs(r[1]["moms"]).max()) == 0.0                 # rank 0's state won the broadcast
    total = r[0]["local"] + r[1]["local"]
    assert not np.array_equal(r[0]["local"], r[1]["local"])
    for k in range(world):
        assert np.array_equal(r[k]["reduced"], total)               # every element reduced once, bit for bit


# ---------------------------------------------------------------------------
# world 8 (the size SCALE_rNN is measured at), without hardware: the subnets' flat buckets at their REAL
# size (6.46 M parameters, HeadConfig defaults) and the R-50-FPN backbone's four buckets, 8 gloo ranks
# ---------------------------------------------------------------------------

def _world8_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    from ssad_amd.head_pipeline import FlatParams
    cfg = HeadConfig(num_gpus=world)
    params, grads, moms = (FlatParams(cfg, "cpu") for _ in range(3))
    torch.manual_seed(500 + rank)                     # every rank starts somewhere else
    params.flat.copy_(torch.randn(params.flat.shape))
    bb = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True, lr=0.01, process_group=dist.group.WORLD,
                         world_size=world)
    bb.params_flat.add_(float(rank))
    dp = BucketedAllReduce(dist.group.WORLD, world)
    dp.broadcast([params.flat, moms.flat], src=0)
    bb.broadcast_params()
    gr = torch.Generator().manual_seed(900 + rank)
    grads.flat.copy_(torch.randn(grads.flat.shape, generator=gr))
    bb.grads_flat.copy_(torch.randn(bb.grads_flat.shape, generator=gr))
    # the step's order: the subnets' "late" bucket, then "early", then the backbone's four as backward reaches them
    for name in ("late", "early"):
        dp.issue(grads.bucket[name])
    for n in bb.bucket:
        bb.dp.issue(bb.bucket[n])
    dp.wait()
    bb.dp.wait()
    # identical SGD on every rank (subnets: the oracle's update of optimizer.py:95-130)
    for name, _, is_bias, _ in params.specs[:4]:
        w, g, m = oracle.sgd_update(params[name].numpy(), grads[name].numpy(), moms[name].numpy(),
                                    0.01 / world, 0.9, 1e-4, is_bias)
        params[name].copy_(torch.from_numpy(w))
    import hashlib
    h = lambda t: hashlib.sha256(t.numpy().tobytes()).hexdigest()
    lo, hi = shard_images(16 * world, rank, world)
    with open(os.path.join(outdir, "w8_%d.txt" % rank), "w") as f:
        f.write("\n".join([h(grads.flat), h(bb.grads_flat), h(params.flat), h(bb.params_flat), h(moms.flat),
                           "%d %d" % (lo, hi), "%d %d" % (grads.flat.numel(), bb.grads_flat.numel()),
                           repr(float(grads.flat[:1000].double().sum())), repr(sorted(grads.bucket)),
                           repr(list(bb.bucket))]))
    if rank == 0:
        np.save(os.path.join(outdir, "w8_head.npy"), grads.flat[:4096].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_exchange_of_the_real_bucket_sizes():
    """no 8-GPU node was ever available, so the world-8 exchange is exercised here -- 8 gloo
    ranks (this box has 8 cores), the subnets' buckets at full size (25.9 MB) + the R-50-FPN backbone's four:
    reduced buckets, broadcast parameters and the updated parameters bitwise identical on ALL ranks
    (nccl_ops_test.py:77-79), the reduced values = the sum of the eight rank-local gradients, and shard_images tiling a
    128-image global batch."""
    world = 8
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_world8_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = [open(os.path.join(d, "w8_%d.txt" % i)).read().split("\n") for i in range(world)]
        head = np.load(os.path.join(d, "w8_head.npy"))
    for i in range(1, world):
        for k in (0, 1, 2, 3, 4, 6, 7, 8, 9):
            assert r[i][k] == r[0][k], (i, k)
    assert r[0][6].split()[0] == "6463220"                       # head parameters
    assert r[0][8] == "['early', 'late']" and r[0][9] == "['fpn', 'res5', 'res4', 'res3']"
    # the shards tile the global batch in rank order
    spans = [tuple(int(v) for v in r[i][5].split()) for i in range(world)]
    assert spans == [(16 * i, 16 * i + 16) for i in range(world)]
    # the reduced gradient = sum over ranks of the rank-local gradients (8 fp32 addends: order-dependent in the last
    # bit, so a tolerance here -- and bit equality ACROSS ranks above)
    want = np.zeros(4096, np.float64)
    for i in range(world):
        g = torch.Generator().manual_seed(900 + i)
        want += torch.randn(6463220, generator=g)[:4096].double().numpy()
    assert np.allclose(head, want, rtol=0, atol=1e-5)


def test_bench_launcher_refuses_a_short_node():
    """`python bench.py --gpus 8` on a box with fewer GPUs must refuse (exit 2) BEFORE starting any rank: the launcher
    path of bench.py up to its device check (a SCALE run can never silently shrink)."""
    import subprocess
    import sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("this node has 8 GPUs: the refusal cannot be exercised")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2, (p.returncode, p.stderr[-400:])
    assert "--gpus 8 asked for" in p.stderr and "refusing" in p.stderr
    assert p.stdout.strip() == ""                                # no JSON line from a refused run
