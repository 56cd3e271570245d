"""The data-parallel exchange on real GPUs over RCCL (backend "nccl"): N ranks = N
processes = N GPUs run one DistillHeads iteration on rank-local images, all-reduce the two
flat gradient buckets while backward is still running, and apply the SGD update.

Checked (cf. caffe2/caffe2/contrib/nccl/nccl_ops_test.py:56-80: every GPU's output equals
the sum of the inputs): the reduced buckets are bitwise identical on every rank and equal the
sum of the rank-local gradients; the parameters after the update are bitwise identical on
every rank; `bench.py --gpus N` refuses to run when the node has fewer than N GPUs.

world = 1 forces the collectives onto a 1-rank RCCL communicator (SSAD_DP_FORCE=1) so the
whole code path runs on a one-GPU box; world >= 2 skips unless that many GPUs are visible.
"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:
        os.environ["SSAD_DP_FORCE"] = "1"
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import ssad_amd  # noqa: F401
    from ssad_amd import kernels as K, synth
    from ssad_amd.head_pipeline import DistillHeads
    from ssad_amd.modeling.retinanet_heads import HeadConfig
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    K.lib()
    shapes = [(10, 14), (5, 7)]
    N = 2
    cfg = HeadConfig(num_gpus=world)
    # rank 0 owns the initial parameters; the others start from different ones and must
    # receive rank 0's through the broadcast (utils/net.py:185-208)
    S = synth.head_params(np.random.default_rng(10 + rank))
    T = synth.head_params(np.random.default_rng(5))
    rng = np.random.default_rng(1234 + rank)               # rank-local images
    f = [torch.from_numpy(a).to(dev) for a in synth.fpn_features(rng, N, shapes)]
    labs = [synth.distill_inputs(rng, N, 9, 80, h, w)[2] for h, w in shapes]
    tg = [synth.bbox_targets(rng, l) for l in labs]
    fg = torch.tensor([float(max(1, sum(t[0].shape[0] for t in tg)))], device=dev)
    labs_d = [torch.from_numpy(a).to(dev) for a in labs]
    tg_d = [(torch.from_numpy(y).to(dev), torch.from_numpy(l).to(dev)) for y, l in tg]

    # (1) rank-local gradients: the same iteration with the exchange switched off
    local = DistillHeads(cfg, N=N, shapes=shapes, device=dev, student_init=S, teacher_init=T)
    dist.broadcast(local.params.flat, src=0)
    local.step(f, f, labs_d, update=False, bbox_targets=tg_d, fg_num=fg)
    local_g = local.grads.flat.clone()

    # (2) the data-parallel iteration
    heads = DistillHeads(cfg, N=N, shapes=shapes, device=dev, student_init=S, teacher_init=T,
                         process_group=dist.group.WORLD, world_size=world, lr=0.01)
    heads.broadcast_params()
    p0 = heads.params.flat.clone()
    heads.step(f, f, labs_d, update=False, bbox_targets=tg_d, fg_num=fg)
    reduced = heads.grads.flat.clone()
    heads.sgd_step()
    torch.cuda.synchronize()

    # every rank gathers everyone's tensors and compares on the device
    def gather(t):
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return out
    all_local, all_red = gather(local_g), gather(reduced)
    all_p0, all_p1 = gather(p0), gather(heads.params.flat)
    res = {
        "params_broadcast_equal": all(torch.equal(all_p0[0], t) for t in all_p0),
        "reduced_equal": all(torch.equal(all_red[0], t) for t in all_red),
        "params_after_equal": all(torch.equal(all_p1[0], t) for t in all_p1),
        "params_changed": not torch.equal(all_p0[0], all_p1[0]),
        "local_differs": world == 1 or not torch.equal(all_local[0], all_local[1]),
    }
    want = torch.stack([t.double() for t in all_local]).sum(0)
    err = (reduced.double() - want).abs().max().item()
    res["sum_err"] = err
    res["sum_scale"] = want.abs().max().item()
    # two fp32 addends (or one): the sum is order independent, so it must match bit for bit
    res["sum_exact"] = (world > 2) or torch.equal(reduced, sum(all_local[1:], all_local[0]))
    res["finite"] = bool(torch.isfinite(heads.params.flat).all())
    # the native backbone's four gradient buckets (FPN, res5, res4, res3) through the same exchange
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    hw = (128, 128)
    bb = NativeResNetFPN("r50", 1, hw, dev, train=True, lr=0.01, process_group=dist.group.WORLD, world_size=world)
    if rank != 0:
        bb.params_flat.mul_(1.5)                       # must be overwritten by the broadcast
    bb.broadcast_params()
    img = torch.randn((1, 3) + hw, device=dev, generator=torch.Generator(device=dev).manual_seed(50 + rank))
    bb.pack()
    out = bb.forward(img)
    d_out = [torch.randn(t.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(70 + rank))
             for t in out]
    solo = NativeResNetFPN("r50", 1, hw, dev, train=True, lr=0.01)          # no exchange: rank-local gradients
    solo.params_flat.copy_(bb.params_flat); solo.frozen_flat.copy_(bb.frozen_flat)
    solo.pack(); solo.forward(img); solo.backward(d_out); solo.dp.wait()
    bb.backward(d_out)
    bb.dp.wait()
    torch.cuda.synchronize()
    bl, br = gather(solo.grads_flat), gather(bb.grads_flat)
    bwant = torch.stack([t.double() for t in bl]).sum(0)
    res["bb_reduced_equal"] = all(torch.equal(br[0], t) for t in br)
    res["bb_sum_err"] = (bb.grads_flat.double() - bwant).abs().max().item()
    res["bb_sum_scale"] = bwant.abs().max().item()
    bb.sgd_step()
    torch.cuda.synchronize()
    bp = gather(bb.params_flat)
    res["bb_params_equal"] = all(torch.equal(bp[0], t) for t in bp)
    # the object bench.py measures -- NativeDistillModel: both backbones, subnets, losses, both updates -- under
    # bench.py's schedule (step on a high-priority stream, teacher on a side stream running ahead, filter gradients
    # on the auxiliary streams) WITH the collectives in flight: three iterations enqueued back to back.  One rank:
    # same bits as the model that issues no collective at all (an all-reduce over one rank is the identity, so any
    # difference is an ordering bug between RCCL's stream and the step's).  More ranks: rank-local images,
    # identical parameters everywhere afterwards.
    from ssad_amd.backbone_pipeline import NativeDistillModel
    mhw, mshapes, mN = (128, 256), [(16, 32), (8, 16), (4, 8), (2, 4), (1, 2)], 2
    mrng = np.random.default_rng(900 + rank)
    mlabs = [torch.from_numpy(synth.distill_inputs(mrng, mN, 9, 80, h, w)[2]).to(dev) for h, w in mshapes]
    mtg = [synth.bbox_targets(mrng, l.cpu().numpy()) for l in mlabs]
    mfg = torch.tensor([float(max(1, sum(t[0].shape[0] for t in mtg)))], device=dev)
    mtg = [(torch.from_numpy(y).to(dev), torch.from_numpy(l).to(dev)) for y, l in mtg]
    mimg = torch.randn((mN, 3) + mhw, device=dev, generator=torch.Generator(device=dev).manual_seed(300 + rank))
    S0, T0 = synth.head_params(np.random.default_rng(1)), synth.head_params(np.random.default_rng(2))

    def run_model(pg_):
        hd = DistillHeads(cfg, N=mN, shapes=mshapes, device=dev, student_init=S0, teacher_init=T0, lr=1e-3,
                          process_group=pg_, world_size=world)
        hd.broadcast_params()
        m = NativeDistillModel(hd, "r50", "r50", mN, mhw, dev, process_group=pg_, world_size=world, lr=1e-3)
        assert m.side is not None and m._teacher_ahead
        main = torch.cuda.Stream(priority=-1)
        main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main):
            for _ in range(3):
                m.step(mimg, mlabs, mtg, mfg)
        torch.cuda.synchronize()
        return m
    dpm = run_model(dist.group.WORLD)
    assert dpm.student.dp.active and dpm.heads.dp.active
    res["model_finite"] = bool(torch.isfinite(dpm.student.params_flat).all() and
                               torch.isfinite(dpm.heads.params.flat).all() and torch.isfinite(dpm.heads.losses).all())
    mp_b, mp_h = gather(dpm.student.params_flat), gather(dpm.heads.params.flat)
    res["model_params_equal"] = all(torch.equal(mp_b[0], t) for t in mp_b) and all(torch.equal(mp_h[0], t) for t in mp_h)
    if world == 1:
        solo_m = run_model(None)
        assert not solo_m.student.dp.active
        res["model_matches_no_collective_run"] = bool(
            torch.equal(solo_m.student.params_flat, dpm.student.params_flat) and
            torch.equal(solo_m.student.moms_flat, dpm.student.moms_flat) and
            torch.equal(solo_m.heads.params.flat, dpm.heads.params.flat) and
            torch.equal(solo_m.heads.losses, dpm.heads.losses))
    else:
        res["model_matches_no_collective_run"] = True
    # the exchange as OPERATORS of the C-ABI surface (optimizer.py:72-92 `model.net.NCCLAllreduce(grads, grads)`,
    # cuda_nccl_op_gpu.cc:68-88): rank 0 creates the id, every rank joins, each rank's net lists its own blob
    from ssad_amd.caffe2_hip import caffe2_pb2, core, workspace as c2ws, dyndep
    dyndep.InitOpsLibrary()
    c2ws.ResetWorkspace()
    ids = [c2ws.CommUniqueId() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    assert c2ws.CommWorld() == 0
    gpu_opt = core.DeviceOption(caffe2_pb2.HIP, rank)
    mine = (np.arange(1000, dtype=np.float32) * 0.25 + 10.0 * rank).reshape(10, 100)
    c2ws.FeedBlob("g", mine, gpu_opt)
    with core.DeviceScope(gpu_opt):
        ar = core.CreateOperator("NCCLAllreduce", ["g"], ["g"])
        bc = core.CreateOperator("NCCLBroadcast", ["w"], ["w"], root=0)
    c2ws.RunOperatorOnce(ar)                                   # no communicator yet: the single-GPU no-op
    res["op_noop_without_comm"] = bool(np.array_equal(c2ws.FetchBlob("g"), mine))
    c2ws.CommInit(ids[0], world, rank, gpu_id=rank)
    res["op_comm_world"] = c2ws.CommWorld()
    c2ws.RunOperatorOnce(ar)
    want_g = sum((np.arange(1000, dtype=np.float32) * 0.25 + 10.0 * r).reshape(10, 100) for r in range(world))
    res["op_allreduce_ok"] = bool(np.array_equal(c2ws.FetchBlob("g"), want_g.astype(np.float32)))
    c2ws.FeedBlob("w", np.full((7,), float(rank + 3), np.float32), gpu_opt)
    c2ws.RunOperatorOnce(bc)
    res["op_broadcast_ok"] = bool(np.array_equal(c2ws.FetchBlob("w"), np.full((7,), 3.0, np.float32)))
    c2ws.FeedBlob("h", (np.ones(64) * (rank + 1)).astype(np.float16), gpu_opt)
    with core.DeviceScope(gpu_opt):
        c2ws.RunOperatorOnce(core.CreateOperator("NCCLAllreduce", ["h"], ["h2"]))
    res["op_allreduce_f16_ok"] = bool(np.array_equal(c2ws.FetchBlob("h2"),
                                                     np.full(64, world * (world + 1) / 2, np.float16)))
    c2ws.CommDestroy()
    if rank == 0:
        np.savez(os.path.join(outdir, "res.npz"), **{k: np.asarray(v) for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_rccl_allreduce_buckets_and_update(world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, node has %d" % (world, torch.cuda.device_count()))
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r = dict(np.load(os.path.join(d, "res.npz")))
    assert bool(r["params_broadcast_equal"]) and bool(r["reduced_equal"])
    assert bool(r["params_after_equal"]) and bool(r["params_changed"]) and bool(r["finite"])
    assert bool(r["local_differs"]) and bool(r["sum_exact"])
    assert float(r["sum_err"]) <= 1e-6 * float(r["sum_scale"])
    assert bool(r["bb_reduced_equal"]) and bool(r["bb_params_equal"])
    assert float(r["bb_sum_err"]) <= 1e-6 * float(r["bb_sum_scale"])
    assert bool(r["model_finite"]) and bool(r["model_params_equal"]) and bool(r["model_matches_no_collective_run"])
    assert bool(r["op_noop_without_comm"]) and int(r["op_comm_world"]) == world
    assert bool(r["op_allreduce_ok"]) and bool(r["op_broadcast_ok"]) and bool(r["op_allreduce_f16_ok"])


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts N ranks itself; with fewer than N GPUs it must exit
    non-zero instead of quietly benchmarking fewer."""
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "refusing" in p.stderr and '"metric"' not in p.stdout
    # a launcher-provided world that disagrees with --gpus is refused as well
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1",
                        "--warmup", "0"], env=env2, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and '"metric"' not in p.stdout
