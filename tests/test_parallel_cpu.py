"""CPU, world_size 2, gloo: the flat gradient bucket + all-reduce reproduce the global-mean gradient."""
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
    x = torch.randn(4, 5, 16, generator=torch.Generator().manual_seed(100))
    lo, hi = parallel.shard_range(4, rank, world)
    for _ in range(2):                 # second pass checks zero() really clears the aliased grads
        bucket.zero()
        model(x[lo:hi]).square().mean().backward()
        bucket.all_reduce()
    out[rank] = bucket.flat.clone()
    dist.destroy_process_group()


def test_two_rank_bucket_matches_per_replica_mean():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        flats = [out[r] for r in range(world)]
    assert torch.equal(flats[0], flats[1])
    # single process, batch split into per-replica BN groups (the DataParallel semantics)
    x = torch.randn(4, 5, 16, generator=torch.Generator().manual_seed(100))
    ref = None
    for r in range(world):
        model = _make_model()
        model(x[2 * r:2 * r + 2]).square().mean().backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        ref = g if ref is None else ref + g
    ref /= world
    assert torch.allclose(flats[0], ref, rtol=1e-6, atol=1e-7)


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
