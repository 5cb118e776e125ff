"""CPU: `python bench.py --gpus N` launches its own ranks (it used to exit unless torchrun wrapped it).

--dry-run runs the rank protocol of the real benchmark (rendezvous on 127.0.0.1, barrier, K timed steps, MAX over ranks,
ONE JSON line from rank 0 relayed by the parent) with gloo ranks: a toy conv + BatchNorm network accumulating into the real
parallel.FlatGradBucket, averaged by its all-reduce, no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "4", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    return p


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_self_launches_its_ranks_and_prints_one_json_line(world):
    p = _run("--gpus", str(world))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == world and doc["steps"] == 4 and doc["dry_run"] is True
    # the dry run drives a toy network into the real FlatGradBucket and its gloo all-reduce through the SAME timed
    # protocol as the GPU run: the ranks (different seeds before the broadcast) must end with identical parameters and bucket
    n_params = (9 * 32 + 32) + 2 * 32 + (32 * 64 + 64) + 2 * 64 + (64 * 13 + 13)
    assert doc["ranks_agree"] is True and doc["grad_bucket_bytes"] == n_params * 4
    assert doc["allreduce_ms"] > 0.0
    # rank 1 sleeps 2 ms per step more than rank 0: the line carries the MAX over ranks, and says how far apart they are
    assert doc["ms_per_step"] == doc["rank_ms_per_step_max"] >= doc["rank_ms_per_step_min"]


def test_bench_single_rank_needs_no_launcher():
    p = _run()
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip())["n_gpus"] == 1


def test_bench_propagates_a_failing_rank():
    env = dict(os.environ)
    env.pop("RANK", None)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "2", "--no-such-flag"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert q.returncode != 0 and not q.stdout.strip()


def test_make_step_flag_matrix_without_a_geometry_branch():
    """`--workload ssg --no-prefetch` raised in GraphedStep because make_step marked the step as forking a
    geometry branch that does not exist.  Without prefetch the step is the plain sequence, whatever the fork default says."""
    import types
    import torch
    sys.path.insert(0, ROOT)
    import bench
    hooks = []

    class Mod(torch.nn.Module):
        def register_forward_hook(self, fn):
            hooks.append(fn)

    net = types.SimpleNamespace(sa1=Mod(), sa2=Mod())
    pts = torch.zeros(1, 9, 4096)
    for workload in ("msg", "ssg", "sa"):
        for fork in (None, "top", "sa1", "sa2", "loss", "sa2_bwd"):
            env = dict(os.environ)
            try:
                if fork is None:
                    os.environ.pop("PN2_BENCH_FORK", None)
                else:
                    os.environ["PN2_BENCH_FORK"] = fork
                del hooks[:]
                assert bench.make_step(workload, net, pts, None, None, prefetch=False).fork_in_step is False
                assert not hooks
                with_branch = bench.make_step(workload, net, pts, None, None, prefetch=True).fork_in_step
                assert with_branch == (workload in ("msg", "ssg") and (fork or {"msg": "sa2", "ssg": "sa2"}[workload]) != "top")
            finally:
                os.environ.clear()
                os.environ.update(env)
