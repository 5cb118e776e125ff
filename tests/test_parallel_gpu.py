"""GPU: the RCCL leg of the data-parallel path on the one GPU a test box has.

PN2_FORCE_COLLECTIVES=1 makes a single rank go through everything N ranks go through: init_process_group("nccl")
(= RCCL), broadcast_module, a forward + backward that accumulates straight into FlatGradBucket(direct=True), and the
flat all-reduce (AVG over one rank = identity, so the gradients must come back unchanged and finite).  Runs in a child
process: a process group belongs to its process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from pointnet12_amd import parallel, pointnet2 as M, synthetic as syn
from pointnet12_amd.loss import nll_loss
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
net = M.PointNet2SemSeg(13, 6).to(dev).train()
before = [p.detach().clone() for p in net.parameters()]
parallel.broadcast_module(net)                       # forced: really broadcasts (from rank 0 to rank 0)
assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
bucket = parallel.FlatGradBucket(net, direct=True)
pts, lab = syn.kitti_batch(0, 2, 1024)
pts, lab = torch.from_numpy(pts).to(dev), torch.from_numpy(lab).to(dev)
bucket.zero()
torch.manual_seed(1)
lp = net(pts)
nll_loss(lp.reshape(-1, 13), lab.reshape(-1)).backward()
g0 = bucket.flat.clone()
out = bucket.all_reduce()
torch.cuda.synchronize()
assert out is bucket.flat, "the collective was skipped"
assert bool(torch.isfinite(bucket.flat).all()) and float(g0.abs().max()) > 0
assert float((bucket.flat - g0).abs().max()) <= 1e-7 * float(g0.abs().max())
dist.barrier()
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


CHILD_COMM = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from pointnet12_amd import parallel, pointnet2 as M, synthetic as syn
from pointnet12_amd.graph import GraphedStep
from pointnet12_amd.loss import nll_loss
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
net = M.PointNet2SemSeg(13, 6).to(dev).train()
bucket = parallel.FlatGradBucket(net, direct=True).use_comm_stream()
assert bucket.comm is not None and bucket.reduced is not None
pts, lab = syn.kitti_batch(0, 2, 1024)
pts, lab = torch.from_numpy(pts).to(dev), torch.from_numpy(lab).to(dev)

def step():
    bucket.wait_reduced()
    bucket.zero()
    lp = net(pts)
    loss = nll_loss(lp.reshape(-1, 13), lab.reshape(-1))
    loss.backward()
    return loss

torch.manual_seed(1)
graphed = GraphedStep(step, dev, warmup=2, geometry_fn=lambda: net.features(pts))
sums = []
for it in range(6):
    graphed()
    marks = bucket.all_reduce_timed()             # on the comm stream, behind the replay
    assert marks is not None, "the collective was skipped"
    # NO synchronisation here: the next replay's zero() must itself wait for this all-reduce (the captured event wait);
    # a replay that zeroed early would hand the collective a half-cleared bucket and the sum below would shrink
    if it >= 2:
        bucket.wait_reduced()
        sums.append(bucket.flat.double().abs().sum())
torch.cuda.synchronize()
vals = [float(v) for v in sums]
assert all(v > 0 and v == v for v in vals), vals
# same input, BatchNorm in train mode, FPS starts redrawn per replay: the gradient mass stays within a few percent
assert max(vals) / min(vals) < 1.5, vals
assert marks[0].elapsed_time(marks[1]) >= 0.0
dist.barrier()
dist.destroy_process_group()
print("RCCL-COMM-STREAM-OK")
"""


CHILD_TWO = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from pointnet12_amd import parallel, pointnet2 as M, synthetic as syn
from pointnet12_amd.graph import GraphedStep
from pointnet12_amd.loss import nll_loss
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ok, seen = parallel.verify_in_graph_record(dev)
if not ok:
    # The HIP runtime torch 2.10 + rocm7.0 ships (7.0.51831) REFUSES hipEventRecordWithFlags(hipEventRecordExternal) under
    # stream capture (hipErrorInvalidValue, whatever the event's creation flags: profiles/r04_ext_record_probe.txt), so the
    # protocol cannot be armed here: what must then hold is that the check says so and nothing half-armed is left behind.
    assert isinstance(seen, str) and "hipEventRecordWithFlags" in seen, seen
    print("RCCL-TWO-BUCKET-UNSUPPORTED-BY-RUNTIME", seen)
    sys.exit(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pts, lab = syn.kitti_batch(0, 2, 1024)
pts, lab = torch.from_numpy(pts).to(dev), torch.from_numpy(lab).to(dev)
res = []
for two in (False, True):
    torch.manual_seed(0)
    net = M.PointNet2SemSegMsg(13, 6).to(dev).train()
    net.drop1.p = 0.0
    bucket = parallel.FlatGradBucket(net, direct=True).use_comm_stream()
    if two:
        bucket.use_two_buckets(list(net.sa1.parameters()))
        bucket.arm(net.sa1)
        assert 0 < bucket.n_late < bucket.flat.numel() // 8

    def step():
        bucket.wait_reduced()
        bucket.zero()
        lp = net(pts)
        loss = nll_loss(lp.reshape(-1, 13), lab.reshape(-1))
        loss.backward()
        return loss

    torch.manual_seed(1)
    graphed = GraphedStep(step, dev, warmup=2, geometry_fn=lambda: net.features(pts))
    torch.manual_seed(7)
    for it in range(4):                             # back to back, no host synchronisation between replay and collective
        graphed()
        assert bucket.all_reduce_timed() is not None
    bucket.wait_reduced()
    torch.cuda.synchronize()
    res.append(bucket.flat.clone())
a, b = res
assert bool(torch.isfinite(b).all()) and float(b.abs().max()) > 0
# one rank: AVG is the identity, so both protocols must hand back the step's own gradients (same seeds, same replays)
assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max()), float((a - b).abs().max()) / float(a.abs().max())
assert abs(float(a.norm()) - float(b.norm())) <= 5e-3 * float(a.norm())
dist.barrier()
dist.destroy_process_group()
print("RCCL-TWO-BUCKET-OK")
"""


def test_one_rank_two_bucket_all_reduce_under_graph_replay(dev):
    """The two-bucket protocol with one forced rank: the in-graph event record is checked first (parallel.verify_in_graph_record),
    then four replayed MSG steps with the early all-reduce waiting for that event and the late one for the step must leave
    the gradients of the one-bucket run (up to the atomics-order noise of two separate runs)."""
    env = dict(os.environ, PN2_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29536",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD_TWO % ROOT], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0 and ("RCCL-TWO-BUCKET-OK" in p.stdout or "RCCL-TWO-BUCKET-UNSUPPORTED-BY-RUNTIME" in p.stdout), \
        (p.stdout[-1000:], p.stderr[-3000:])


def test_one_rank_rccl_all_reduce_on_comm_stream_under_graph_replay(dev):
    """FlatGradBucket.use_comm_stream(): the collective on its own stream, the captured step waiting for it through an
    (external) event in front of bucket.zero() -- replayed back to back with no host synchronisation in between."""
    env = dict(os.environ, PN2_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29535",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD_COMM % ROOT], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0 and "RCCL-COMM-STREAM-OK" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


def test_one_rank_rccl_bucket_all_reduce(dev):
    env = dict(os.environ, PN2_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0 and "RCCL-ONE-RANK-OK" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


def test_bench_self_launch_one_rank_through_torchrun(dev):
    """bench.py started by torch.distributed.run with one rank: the same code path the driver's N-rank launch takes
    (RCCL init, broadcast, barrier-bracketed timing, all-reduce after every step)."""
    env = dict(os.environ, PN2_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    import socket
    with socket.socket() as sk:                   # a free port of this box (a fixed one collided with a lingering rendezvous twice
        sk.bind(("127.0.0.1", 0))                 # in the round-5 evidence runs, where the suite follows two dozen bench launches)
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "3", "--warmup", "2", "--workload", "ssg", "--no-cpu-baseline", "--no-roofline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    doc = json.loads(line[0])
    assert doc["n_gpus"] == 1 and doc["value"] > 1e6 and doc["config"]["parallelism"] == "dp1"


def test_bench_two_ranks_on_one_device_or_records_the_refusal(dev):
    """the N > 1 path of bench.py (RCCL communicator over two ranks, disjoint-shard check, comm-stream
    verification, RCCL log parsing, all-reduce after every step, MAX over ranks) with a REAL peer -- two ranks on the one GPU of
    this box (`--same-device`).  RCCL may refuse two ranks on one device ("Duplicate GPU detected"): then the refusal is what is
    recorded (gpurun_out/rccl_same_device.txt) and the launch must have failed cleanly -- non-zero exit of the parent, no JSON
    line, no hang (the parent starts its ranks as child processes and never re-execs)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--steps", "3", "--warmup", "2",
                        "--workload", "ssg", "--batch", "4", "--no-cpu-baseline", "--no-roofline", "--no-other-configs"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_same_device.txt"), "w") as f:
        if p.returncode == 0:
            assert len(lines) == 1, p.stdout[-2000:]
            doc = json.loads(lines[0])
            assert doc["n_gpus"] == 2 and doc["config"]["parallelism"] == "dp2" and doc["config"]["global_batch"] == 8
            assert doc["rccl"]["first_cloud_per_rank"] == [0, 4] and doc["rccl"]["ranks"] == 2
            assert doc["allreduce_ms"] > 0.0 and doc["rank_ms_per_step_max"] >= doc["rank_ms_per_step_min"] > 0.0
            f.write("two ranks on one device: RCCL accepted\n" + json.dumps({k: doc[k] for k in ("ms_per_step", "allreduce_ms", "rccl", "config")}, indent=1) + "\n")
        else:
            refused = [ln for ln in p.stderr.splitlines() if "Duplicate GPU" in ln or "invalid usage" in ln.lower() or "ncclInvalidUsage" in ln]
            f.write("two ranks on one device: launch failed with exit code %d\n" % p.returncode + "\n".join(refused[:6]) + "\n---- stderr tail\n" + p.stderr[-3000:])
            assert not lines, "a failed launch must not print a result line"
            assert refused, p.stderr[-3000:]            # the only accepted failure is RCCL's refusal of a duplicate device
