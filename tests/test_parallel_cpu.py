"""CPU, gloo, world_size 2 / 4 / 8: the flat gradient bucket + all-reduce reproduce the global-mean gradient."""
import pytest
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from pointnet12_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv1d(5, 8, 1), nn.BatchNorm1d(8), nn.ReLU(), nn.Conv1d(8, 3, 1))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _make_model()
    if rank != 0:                      # perturb: broadcast must restore rank 0's values
        for p in model.parameters():
            p.data.add_(1.0)
    parallel.broadcast_module(model)
    bucket = parallel.FlatGradBucket(model)
    x = torch.randn(8, 5, 16, generator=torch.Generator().manual_seed(100))
    lo, hi = parallel.shard_range(8, rank, world)
    assert bucket.use_comm_stream() is bucket and bucket.comm is None     # host tensors: there is no stream to move to
    for it in range(2):                # second pass checks zero() really clears the aliased grads
        bucket.wait_reduced()          # (the call sites of the GPU step; no-ops here)
        bucket.zero()
        model(x[lo:hi]).square().mean().backward()
        # the timed form is what bench.py calls: on the CPU it performs the collective and has no events to hand back
        if it == 0:
            assert bucket.all_reduce() is bucket.flat
        else:
            assert bucket.all_reduce_timed() is None
    out[rank] = bucket.flat.clone()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_ranks_bucket_matches_per_replica_mean(world):
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        flats = [out[r] for r in range(world)]
    assert all(torch.equal(flats[0], f) for f in flats[1:])
    # single process, batch split into per-replica BN groups (the DataParallel semantics)
    x = torch.randn(8, 5, 16, generator=torch.Generator().manual_seed(100))
    per = 8 // world
    ref = None
    for r in range(world):
        model = _make_model()
        model(x[per * r:per * r + per]).square().mean().backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        ref = g if ref is None else ref + g
    ref /= world
    assert torch.allclose(flats[0], ref, rtol=1e-6, atol=1e-7)


def _two_bucket_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for two in (False, True):
        model = _make_model()
        bucket = parallel.FlatGradBucket(model)
        if two:
            bucket.use_two_buckets(list(model[0].parameters()))          # the first stage's parameters: the late bucket
            assert bucket.n_late == 5 * 8 + 8 and bucket.early_ready is None   # host tensors: no event, no comm stream
            h = bucket.arm(model[0])
        x = torch.randn(4, 5, 16, generator=torch.Generator().manual_seed(100))
        lo, hi = parallel.shard_range(4, rank, world)
        bucket.zero()
        model(x[lo:hi]).square().mean().backward()
        bucket.all_reduce()
        res.append(bucket.flat.clone())
        if two:
            h.remove()
            try:
                bucket.use_two_buckets(list(model[3].parameters()))      # not a prefix of the bucket
                res.append("no error")
            except ValueError:
                pass
    out[rank] = res
    dist.destroy_process_group()


def test_two_bucket_all_reduce_equals_one_bucket_over_two_gloo_ranks():
    """The two-bucket protocol (early = everything but the first stage, late = the first stage) gives the one-bucket result bit
    for bit -- the same elementwise reduction in two pieces -- on both ranks; a late set that is not a prefix is refused."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_two_bucket_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    for r in range(world):
        assert len(res[r]) == 2
        assert torch.equal(res[r][0], res[r][1])
    assert torch.equal(res[0][1], res[1][1])


def test_bucket_aliases_grads():
    model = _make_model()
    bucket = parallel.FlatGradBucket(model)
    assert bucket.nbytes == 4 * sum(p.numel() for p in model.parameters())
    model(torch.randn(2, 5, 7)).sum().backward()
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in model.parameters())
    assert float(bucket.flat.abs().sum()) > 0
    bucket.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in model.parameters())
    assert bucket.all_reduce() is None          # no process group: no-op
    assert bucket.all_reduce_timed() is None
    bucket.wait_reduced()                       # nothing to wait for


def test_external_event_binds_to_the_loaded_hip_runtime():
    """parallel._ExternalEvent must talk to the HIP runtime torch itself is running on (torch ships its own copy; a second copy
    opened by bare name would not know torch's streams)."""
    path = parallel._loaded_hip_runtime()
    assert "libamdhip64.so" in os.path.basename(path)
    if os.path.isabs(path):
        assert os.path.exists(path)


# ------------------------------------------------------------------ the path's own network over gloo ranks
def _oracle_net():
    from oracle import torch_ref as T
    torch.manual_seed(3)
    return T.RefSSGSemSeg(13, 6, dropout=0.0).train()


def _oracle_batch():
    from pointnet12_amd import synthetic as syn
    pts, labels = syn.kitti_batch(50, 4, 1024)
    return torch.from_numpy(pts), torch.from_numpy(labels)


def _oracle_worker(rank, world, port, out):
    from oracle import torch_ref as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _oracle_net()
    if rank != 0:
        for p in net.parameters():
            p.data.mul_(0.5)               # broadcast_module must restore rank 0's parameters
    parallel.broadcast_module(net)
    bucket = parallel.FlatGradBucket(net)
    pts, labels = _oracle_batch()
    lo, hi = parallel.shard_range(pts.shape[0], rank, world)
    bucket.zero()
    torch.manual_seed(100 + rank)          # per-rank FPS start draws, reproduced by the single-process run below
    T.seg_loss(net(pts[lo:hi]), labels[lo:hi]).backward()
    bucket.all_reduce()
    out[rank] = (bucket.flat.clone(), net.sa1.mlp_bns[0].running_mean.clone())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_run_of_the_oracle_network_equals_per_rank_bn_groups(world):
    """World gloo ranks, each running the CPU restatement of SSG-SemSeg on its own clouds (4 / world of them; per-replica BatchNorm,
    as the reference's nn.DataParallel, semseg.py:91) and averaging the flat bucket, against ONE process evaluating the
    shards as separate BatchNorm groups and averaging the gradients: <= 1e-6 relative."""
    from oracle import torch_ref as T
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_oracle_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert all(torch.equal(res[0][0], res[r][0]) for r in range(1, world))        # every rank holds the same averaged gradient
    torch.set_num_threads(2)
    pts, labels = _oracle_batch()
    per = pts.shape[0] // world
    ref = None
    for r in range(world):
        net = _oracle_net()
        torch.manual_seed(100 + r)
        T.seg_loss(net(pts[per * r:per * r + per]), labels[per * r:per * r + per]).backward()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        ref = g if ref is None else ref + g
        if r == 0:
            assert torch.allclose(res[0][1], net.sa1.mlp_bns[0].running_mean, rtol=1e-6, atol=1e-7)   # per-replica BN
    ref /= world
    assert float((res[0][0] - ref).norm()) <= 1e-6 * float(ref.norm())


def test_graph_capture_error_mode_follows_the_process_group():
    """Round 6: the RCCL watchdog thread polls its work events (hipEventQuery); a poll inside a `global`-mode capture window raised
    hipErrorStreamCaptureUnsupported in that thread and aborted the rank (1 of 12 one-rank torchrun launches of bench.py).  Captures
    restrict their own thread only while a process group is alive, and keep torch's default otherwise."""
    from pointnet12_amd import graph
    assert not dist.is_initialized()
    assert graph.capture_error_mode() == "global"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        assert graph.capture_error_mode() == "thread_local"
    finally:
        dist.destroy_process_group()
    assert graph.capture_error_mode() == "global"
