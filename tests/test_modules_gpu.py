"""GPU: the three drop-in modules and the two benchmark networks against the reference's own numbers.

Golden vectors (tests/golden/g5_modules.npz, g6_nets.npz) hold inputs, state_dicts and the outputs /
gradients / BN buffers the REFERENCE produced on CPU.  Tolerances: forward 1e-5 absolute (north-star),
gradients 5e-5 of the tensor's max (reference self-noise is 1.2e-5), conv biases that feed a
training-mode BatchNorm have a mathematically zero gradient and are compared against 0.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from test_oracle_golden import MODULE_INPUTS, module_case
from pointnet12_amd import pointnet2 as M
from pointnet12_amd import pointnet_util as U

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-5
GRAD_TOL = 5e-5

CASES = {
    "sa": lambda: U.PointNetSetAbstraction(256, 0.2, 32, 9, [32, 32, 64], False),
    "sa_nofeat": lambda: U.PointNetSetAbstraction(128, 0.4, 16, 3, [16, 32], False),
    "sa_all": lambda: U.PointNetSetAbstraction(None, None, None, 9, [32, 64], True),
    "msg": lambda: U.PointNetSetAbstractionMsg(128, [0.1, 0.2, 0.4], [16, 32, 64], 6, [[16, 32], [32, 48], [32, 196]]),
    "fp": lambda: U.PointNetFeaturePropagation(30, [32, 16]),
    "fp_noskip": lambda: U.PointNetFeaturePropagation(24, [32, 32, 16]),
    "fp_s1": lambda: U.PointNetFeaturePropagation(30, [16]),
}


def relmax(a, ref):
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-12)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_module_matches_reference(dev, tag):
    g = golden("g5_modules.npz")
    state, ins, seed = module_case(g, tag)
    mod = CASES[tag]()
    mod.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})   # reference keys load as is
    mod.to(dev).train()
    names = MODULE_INPUTS[tag]
    tens = []
    for n, a in zip(names, ins):
        t = None if a is None else torch.from_numpy(a).to(dev)
        if t is not None and ("%s/gin/%s" % (tag, n)) in g.files:
            t.requires_grad_(True)
        tens.append(t)
    torch.manual_seed(seed)                      # same CPU-generator draw for the FPS start as the reference
    y = mod(*tens)
    ys = y if isinstance(y, tuple) else (y,)
    for i, t in enumerate(ys):
        ref = g["%s/out/%d" % (tag, i)]
        assert tuple(t.shape) == ref.shape
        assert np.abs(t.detach().cpu().numpy() - ref).max() <= FWD_TOL, (tag, i)
    (ys[-1] * torch.from_numpy(g[tag + "/gw"]).to(dev)).sum().backward()
    for n, t in zip(names, tens):
        key = "%s/gin/%s" % (tag, n)
        if key in g.files:
            assert relmax(t.grad.cpu().numpy(), g[key]) <= GRAD_TOL, (tag, n)
    for k, p in mod.named_parameters():
        ref = g["%s/gpar/%s" % (tag, k)]
        if "conv" in k and k.endswith("bias"):
            scale = np.abs(g["%s/gpar/%s" % (tag, k.replace("bias", "weight"))]).max()
            assert np.abs(p.grad.cpu().numpy()).max() <= 1e-4 * scale, k      # exact answer is 0
            continue
        assert relmax(p.grad.cpu().numpy(), ref) <= GRAD_TOL, (tag, k)
    for k, v in mod.state_dict().items():
        key = "%s/state1/%s" % (tag, k)
        if key in g.files:
            assert np.allclose(v.cpu().numpy(), g[key], rtol=1e-5, atol=1e-6), k


def test_eval_mode_uses_running_stats(dev):
    g = golden("g5_modules.npz")
    state, ins, seed = module_case(g, "sa")
    mod = CASES["sa"]()
    mod.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    mod.to(dev).eval()
    from oracle import torch_ref as T
    orc = T.RefSetAbstraction(256, 0.2, 32, 9, [32, 32, 64], False)
    T.load_numpy_state(orc, state)
    orc.eval()
    with torch.no_grad():
        torch.manual_seed(3)
        _, a = mod(torch.from_numpy(ins[0]).to(dev), torch.from_numpy(ins[1]).to(dev))
        torch.manual_seed(3)
        _, b = orc(torch.from_numpy(ins[0]), torch.from_numpy(ins[1]))
    assert np.abs(a.cpu().numpy() - b.numpy()).max() <= FWD_TOL
    assert int(mod.mlp_bns[0].num_batches_tracked) == int(state["mlp_bns.0.num_batches_tracked"])


@pytest.mark.parametrize("tag,make", [("ssg", lambda: M.PointNet2SemSeg(13, 6)), ("msg", lambda: M.PointNet2SemSegMsg(13, 6))])
def test_network_matches_reference(dev, tag, make):
    g = golden("g6_nets.npz")
    torch.manual_seed(int(g["init_seed"]))
    net = make()
    net.drop1.p = 0.0
    net.to(dev).train()
    pts = torch.from_numpy(g["points"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    torch.manual_seed(int(g["fwd_seed"]))
    lp = net(pts)
    loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1))
    loss.backward()
    # nine BN-coupled stages deep and B*N = 2048 only: the bound is twice what the REFERENCE moves against itself on
    # this very input when only its thread count changes (8 vs 1), measured with the reference by
    # tools/make_golden.py g6n -> g6_noise.npz (ssg 9.1e-5, msg 4.9e-5); the fp64 yardstick for the same case is in
    # test_parity_fullsize_gpu.py
    noise = float(golden("g6_noise.npz")[tag + "/log_probs_absdiff"])
    assert np.abs(lp.detach().cpu().numpy() - g[tag + "/log_probs"]).max() <= 2 * noise
    assert abs(float(loss) - float(g[tag + "/loss"])) <= 2e-6
    grads = dict(net.named_parameters())
    for n, l2, amax in zip(g[tag + "/grad_names"], g[tag + "/grad_l2"], g[tag + "/grad_absmax"]):
        n = str(n)
        if ("conv" in n and n.endswith("bias") and n != "conv2.bias"):
            continue
        mine = np.linalg.norm(grads[n].grad.double().cpu().numpy())
        # whole-network gradients at B*N = 2048 are dominated by the handful of argmax / ReLU decisions that
        # flip under any fp32 re-ordering (the reference's own thread-count noise does the same); the tight
        # gradient check is the per-module one above, this one guards the composition.  Measured: the torch
        # restatement against ITSELF (1 vs 8 threads, same seeds) moves these L2 norms by up to 4.5e-3
        # (sa1.bn_blocks.0.1.bias, then 3.1e-3, 3.0e-3, 2.7e-3 ...); the bound is 3x that self-noise.
        assert abs(mine - l2) <= 1.5e-2 * l2 + 1e-6 * float(g[tag + "/grad_l2"].max()), (n, mine, l2)
    for k in g.files:
        if k.startswith(tag + "/grad/"):
            n = k[len(tag) + 6:]
            # ELEMENTWISE maxima of whole-network gradients at B*N = 2048 are set by the single largest ReLU / arg-max decision
            # that falls the other way (one flip moves one entry by a few per cent of the tensor's maximum): the reference moves
            # 3.8e-2 against ITSELF on its worst sampled tensor when only its thread count changes (g6_noise.npz, measured with
            # the reference; the MSG entry of that file sits on a tensor whose exact gradient is zero and says nothing), this
            # path measured 1.0e-1 on fp1.mlp_convs.0.weight of MSG (round 4).  The bound is FOUR times the reference's
            # self-noise; the statistic that can be held tight -- every tensor's L2 error against an fp64 evaluation, at most 3x
            # the reference arithmetic's -- is asserted for the same case in
            # test_parity_fullsize_gpu.py::test_small_batch_network_vs_fp64_and_reference_self_noise, and the per-stage
            # gradients at 5e-5 in test_parity_stages_gpu.py.
            assert relmax(grads[n].grad.cpu().numpy(), g[k]) <= 4.0 * float(golden("g6_noise.npz")["ssg/grad_relmax_worst"]), n


def test_head_dropout_in_train_mode(dev):
    """The one float op of the timed step that no oracle comparison covers: ``F.dropout(p=0.5)`` of the head
    (model/pointnet2.py:172).  Its mask comes from torch's device Philox stream and cannot equal the reference's CPU draw,
    so the checks are the ones that do not depend on the draw: survivors are the input scaled by exactly 1/(1-p), the dropped
    fraction is p within sampling error, the mean is preserved, a different seed gives a different mask, eval mode is the
    identity -- and the train-step loss stays in the band of the p = 0 oracle step (dropout perturbs, it must not bias)."""
    from oracle import torch_ref as T
    from pointnet12_amd import synthetic as syn
    from pointnet12_amd.loss import nll_loss
    pts_np, lab_np = syn.kitti_batch(40, 2, 1024)
    torch.manual_seed(0)
    net = M.PointNet2SemSeg(13, 6)
    orc = T.RefSSGSemSeg(13, 6, dropout=0.0)
    orc.load_state_dict(net.state_dict())
    net.to(dev).train()
    seen = {}
    h = net.drop1.register_forward_hook(lambda m, i, o: seen.update(x=i[0].detach(), y=o.detach()))
    pts, labels = torch.from_numpy(pts_np).to(dev), torch.from_numpy(lab_np).to(dev)
    torch.manual_seed(1)
    torch.cuda.manual_seed(7)
    loss = nll_loss(net(pts).reshape(-1, 13), labels.reshape(-1))
    loss.backward()
    x, y = seen["x"], seen["y"]
    assert net.drop1.p == 0.5
    kept = y != 0
    live = x != 0                                        # (post-ReLU input: its own zeros say nothing about the mask)
    assert torch.equal(y[kept], x[kept] * 2.0)           # survivors: exactly x / (1 - p)
    frac = float((kept & live).sum()) / float(live.sum())
    n = int(live.sum())
    assert abs(frac - 0.5) <= 5.0 * 0.5 / n ** 0.5       # five sigma of a fair coin over n draws
    assert abs(float(y.mean()) - float(x.mean())) <= 0.02 * float(x.mean())
    torch.manual_seed(1)
    torch.cuda.manual_seed(8)
    net(pts)
    assert not torch.equal(seen["y"] != 0, kept)         # another seed, another mask
    torch.manual_seed(1)
    orc.train()
    loss_ref = float(T.seg_loss(orc(torch.from_numpy(pts_np)), torch.from_numpy(lab_np)))
    assert abs(float(loss) - loss_ref) <= 0.05 * loss_ref, (float(loss), loss_ref)
    g = net.conv2.weight.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    net.eval()
    with torch.no_grad():
        net(pts)
    assert torch.equal(seen["y"], seen["x"])             # eval mode: identity
    h.remove()


def test_reference_checkpoint_keys_load(dev):
    """state_dict compatibility incl. the DataParallel 'module.' prefix (reference model/utils.py:22-27)."""
    net = M.PointNet2SemSeg(19, 1)
    sd = {"module." + k: v.clone() for k, v in net.state_dict().items()}
    res = M.load_reference_state(M.PointNet2SemSeg(19, 1), sd)
    assert not res.missing_keys and not res.unexpected_keys
    assert len(sd) == 156


@pytest.mark.parametrize("kind,D", [("ssg", 40), ("ssg", 33), ("msg", 64), ("msg", 33), ("msg", 37)])   # msg 33 / 37: features first, D % 4 == 1 (padded copy of the feature columns)
def test_factorised_first_layer_matches_oracle(dev, kind, D):
    """D >= 32 features: layer 1 runs as Zf[idx] + W_x (xyz - centre) (csrc/grouped.hip); compare the whole module,
    forward and backward, with the oracle module on the same state and inputs."""
    from oracle import torch_ref as T
    from pointnet12_amd import synthetic as syn
    assert D >= U.FACTORISE_MIN_FEATURES
    pts = torch.from_numpy(syn.kitti_batch(500, 2, 1024)[0])
    xyz = pts[:, :3].contiguous()
    feat = torch.randn(2, D, 1024, generator=torch.Generator().manual_seed(D))
    torch.manual_seed(D)
    if kind == "ssg":
        orc = T.RefSetAbstraction(128, 0.3, 32, D + 3, [48, 64], False)
        mod = U.PointNetSetAbstraction(128, 0.3, 32, D + 3, [48, 64], False)
    else:
        orc = T.RefSetAbstractionMsg(128, [0.2, 0.4], [16, 32], D, [[32, 64], [48, 96, 128]])
        mod = U.PointNetSetAbstractionMsg(128, [0.2, 0.4], [16, 32], D, [[32, 64], [48, 96, 128]])
    mod.load_state_dict(orc.state_dict())
    mod.to(dev).train()
    orc.train()
    f_ref = feat.clone().requires_grad_(True)
    f_gpu = feat.clone().to(dev).requires_grad_(True)
    torch.manual_seed(9)
    _, a = orc(xyz, f_ref)
    torch.manual_seed(9)
    _, b = mod(xyz.to(dev), f_gpu)
    assert float((a.detach() - b.detach().cpu()).abs().max()) <= FWD_TOL
    gw = torch.randn(a.shape, generator=torch.Generator().manual_seed(1))
    (a * gw).sum().backward()
    (b * gw.to(dev)).sum().backward()
    assert relmax(f_gpu.grad.cpu().numpy(), f_ref.grad.numpy()) <= 1e-4
    for (n, p), (_, q) in zip(orc.named_parameters(), mod.named_parameters()):
        if "conv" in n and n.endswith("bias"):
            continue
        assert relmax(q.grad.cpu().numpy(), p.grad.numpy()) <= 1e-4, n
    for (n, u), (_, v) in zip(orc.named_buffers(), mod.named_buffers()):
        assert np.allclose(v.cpu().numpy(), u.numpy(), rtol=1e-5, atol=1e-6), n


def test_graphed_step_matches_eager(dev):
    """hipGraph replay of forward+loss+backward: same loss and gradients as eager launches, same FPS start draws."""
    from pointnet12_amd import parallel
    from pointnet12_amd.graph import GraphedStep
    g = golden("g6_nets.npz")
    pts = torch.from_numpy(g["points"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    results = []
    for graphed in (False, True):
        torch.manual_seed(int(g["init_seed"]))
        net = M.PointNet2SemSeg(13, 6)
        net.drop1.p = 0.0
        net.to(dev).train()
        bucket = parallel.FlatGradBucket(net)

        def compute():
            bucket.zero()
            lp = net(pts)
            loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1))
            loss.backward()
            return loss
        step = GraphedStep(compute, dev, warmup=2) if graphed else compute
        if not graphed:
            for _ in range(2):          # the graphed variant ran 2 eager warm-up steps: same BN buffer history
                compute()
        torch.manual_seed(77)
        for _ in range(3):
            loss = step()
        results.append((float(loss), bucket.flat.clone(), net.sa1.mlp_bns[0].running_mean.clone(),
                        int(net.sa1.mlp_bns[0].num_batches_tracked)))
    (l0, g0, r0, n0), (l1, g1, r1, n1) = results
    assert n0 == n1 == 5
    assert abs(l0 - l1) <= 1e-5
    assert float((r0 - r1).abs().max()) <= 1e-6
    assert float((g0 - g1).abs().max()) <= 2e-2 * float(g0.abs().max())     # flip noise of the tiny batch, see above


def test_direct_grad_accumulation_matches_autograd(dev):
    """FlatGradBucket(direct=True): the HIP backward adds into the bucket itself; same gradients as via autograd."""
    from pointnet12_amd import parallel
    g = golden("g6_nets.npz")
    pts = torch.from_numpy(g["points"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    flats = []
    try:
        for direct in (False, True):
            torch.manual_seed(int(g["init_seed"]))
            net = M.PointNet2SemSegMsg(13, 6)
            net.drop1.p = 0.0
            net.to(dev).train()
            bucket = parallel.FlatGradBucket(net, direct=direct)
            for _ in range(2):                       # second pass: accumulation starts from a zeroed bucket again
                bucket.zero()
                torch.manual_seed(5)
                lp = net(pts)
                F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1)).backward()
            flats.append(bucket.flat.clone())
    finally:
        U.set_direct_grad_accumulation(False)
    a, b = flats
    assert float(b.abs().max()) > 0
    assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max())       # atomics order + flip noise of the tiny batch
    # exactness check on a flip-free quantity: per-tensor L2 norms
    assert abs(float(a.norm()) - float(b.norm())) <= 5e-3 * float(a.norm())


def _eager_and_captured_sequences(dev):
    """(losses, bucket, recorded geometry per step) of four steps run eagerly, as a captured step with the next batch's geometry on a side
    stream, and with that branch forked between forward and backward -- same draws."""
    from pointnet12_amd import parallel
    from pointnet12_amd.graph import GraphedStep
    g = golden("g6_nets.npz")
    pts = torch.from_numpy(g["points"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    seqs = []
    from pointnet12_amd import graph as G_
    for mode in ("eager", "prefetch", "prefetch-forked"):
        torch.manual_seed(int(g["init_seed"]))
        net = M.PointNet2SemSegMsg(13, 6)
        net.drop1.p = 0.0
        net.to(dev).train()
        bucket = parallel.FlatGradBucket(net)

        def compute():
            bucket.zero()
            lp = net(pts)
            loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1))
            G_.fork_point()                       # (round 4) a no-op except under GraphedStep(fork_in_step=True)
            loss.backward()
            return loss
        torch.manual_seed(31)
        geos = []

        def eager_geometry():
            # the geometry an eager step is about to compute (same generator state: restored afterwards), as a recorded tape
            state = torch.get_rng_state()
            tape = U.GeometryTape()
            U.set_geometry_tape(tape)
            try:
                with torch.no_grad():
                    net.features(pts)
            finally:
                U.set_geometry_tape(None)
            torch.set_rng_state(state)
            return [t.detach().cpu().clone() for t in G_._flatten(tape.items)]
        if mode == "eager":
            for _ in range(2):
                compute()

            def step():
                geos.append(eager_geometry())
                return compute()
        else:
            # "prefetch-forked": the geometry branch of the captured step starts at fork_point(), between forward and backward,
            # instead of at its top -- when it runs changes, what it computes (draws, tapes) does not
            step = GraphedStep(compute, dev, warmup=2, geometry_fn=lambda: net.features(pts), fork_in_step=mode == "prefetch-forked")
        losses = []
        for r in range(4):
            losses.append(float(step()))
            if mode != "eager":
                # the tape replay r ran on (graph r % 2 reads the tape the other graph's side branch wrote one replay earlier)
                geos.append([t.detach().cpu().clone() for t in G_._flatten(step._tapes[r % 2].items)])
        seqs.append((losses, bucket.flat.clone(), geos))
    return seqs


def test_prefetched_geometry_graph_matches_eager(dev):
    """Graph with the next batch's geometry on a side stream: same loss sequence as plain eager steps (same draws)."""
    (la, ga, ta), (lb, gb, tb), (lc, gc, tc) = _eager_and_captured_sequences(dev)
    # the geometry is index-exact -- the recorded tensors (FPS indices, centres, ball-query indices, 3-NN indices and
    # weights) of every step must be bit-identical between the eager run and both captured schedules; a tape or stream-ordering
    # race cannot hide under the loss tolerance below
    for name, other in (("prefetch", tb), ("prefetch-forked", tc)):
        assert len(other) == len(ta) == 4
        for r, (x, y) in enumerate(zip(ta, other)):
            assert len(x) == len(y), (name, r, len(x), len(y))
            for i, (u, v) in enumerate(zip(x, y)):
                assert u.shape == v.shape and u.dtype == v.dtype, (name, r, i)
                if u.dtype == torch.int32:
                    # the inverse neighbour index (pn2_invert_index: members of every target point, for the segmented scatter of the
                    # backward): a counting sort whose slots are handed out by atomics -- the members of a segment come in any order;
                    # the same members must be there
                    assert torch.equal(torch.sort(u.flatten())[0], torch.sort(v.flatten())[0]), (name, r, i)
                    continue
                assert torch.equal(u, v), "%s: recorded geometry %d of step %d differs from the eager step's" % (name, i, r)
    # The forward has no order-dependent arithmetic left except the fp64 statistics atomics (1e-16 of a sum): measured, the three
    # loss sequences are BIT-IDENTICAL (tools/exp/graph_eager_loss_delta.py: 32 of 32 comparisons, round 6).  Back to 2e-5 (round 5
    # had widened it to 2e-4 after one trip in five full-suite runs that nothing reproduced); a difference beyond it must
    # reproduce in a second evaluation -- a wrong start draw or a stale tape moves the loss by 1e-2 every time, a last-bit flip
    # of a statistic does not come twice.
    def close(x, y):
        return np.allclose(x, y, rtol=0, atol=2e-5)
    if not (close(la, lb) and close(la, lc)):
        (la, ga, _), (lb, gb, _), (lc, gc, _) = _eager_and_captured_sequences(dev)
    assert close(la, lb), (la, lb)
    assert abs(float(ga.norm()) - float(gb.norm())) <= 5e-3 * float(ga.norm())
    assert close(la, lc), (la, lc)
    assert abs(float(ga.norm()) - float(gc.norm())) <= 5e-3 * float(ga.norm())


def test_channel_last_copy_is_shared_and_follows_in_place_writes(dev):
    """The [B,C,N] -> [B,N,C] copy of a caller tensor is made once per version of that tensor (sa1 and fp1 consume the same
    network input): the same object comes back while the source is unchanged, a fresh copy after an in-place write, and never a
    kept one for a tensor that requires grad (its copy belongs to one autograd graph)."""
    x = torch.randn(2, 5, 64, device=dev)
    a = U._channel_last(x, "x")
    b = U._channel_last(x, "x")
    assert a is b and torch.equal(a, x.permute(0, 2, 1))
    x.mul_(2.0)
    c = U._channel_last(x, "x")
    assert c is not a and torch.equal(c, x.permute(0, 2, 1))
    rows = torch.randn(2, 64, 5, device=dev)
    v = rows.permute(0, 2, 1)                                   # channel-first VIEW of channel-last storage: no copy at all
    assert U._channel_last(v, "v").data_ptr() == rows.data_ptr()
    g = torch.randn(2, 5, 64, device=dev, requires_grad=True)
    assert U._channel_last(g, "g") is not U._channel_last(g, "g")


def test_graph_replays_leave_the_callers_memory_alone(dev):
    """Every buffer a captured step writes must be owned by the GraphedStep: tensors the caller allocates AFTER the constructor
    (here: many small ones, which the allocator serves from whatever the constructor released) keep their contents across
    replays.  (Round 3: the eagerly recorded geometry tape was released at the end of the constructor while the second graph's
    prefetch branch still wrote into it.)"""
    import gc
    from pointnet12_amd import parallel, synthetic as syn
    from pointnet12_amd.graph import GraphedStep
    from pointnet12_amd.loss import nll_loss
    torch.manual_seed(0)
    net = M.PointNet2SemSeg(13, 6).to(dev).train()
    bucket = parallel.FlatGradBucket(net, direct=True)
    pts, lab = syn.kitti_batch(0, 2, 1024)
    pts, lab = torch.from_numpy(pts).to(dev), torch.from_numpy(lab).to(dev)

    def step():
        bucket.zero()
        lp = net(pts)
        loss = nll_loss(lp.reshape(-1, 13), lab.reshape(-1))
        loss.backward()
        return loss
    torch.manual_seed(1)
    try:
        graphed = GraphedStep(step, dev, warmup=2, geometry_fn=lambda: net.features(pts))
        gc.collect()
        # sizes from one element up to the tape's largest tensors, in both allocator pools (small blocks < 1 MiB, large above)
        sentinels = [torch.full((n,), 7.25, device=dev, dtype=torch.float64)
                     for n in [1] * 64 + [16] * 64 + [512] * 32 + [4096] * 32 + [65536] * 16 + [262144] * 8]
        torch.cuda.synchronize()
        for _ in range(6):
            graphed()
        torch.cuda.synchronize()
        assert all(bool((t == 7.25).all()) for t in sentinels)
    finally:
        U.set_direct_grad_accumulation(False)


def test_cfg2_full_size_set_abstraction_vs_oracle(dev):
    """BASELINE.json configs[1] at its real size: PointNetSetAbstraction(1024, 0.1, 32, 9, [32,32,64]), B=8 x 4096 KITTI-shaped
    clouds, forward + backward, against the oracle module on the host CPU (about 2 s there)."""
    from oracle import torch_ref as T
    from pointnet12_amd import synthetic as syn
    pts = torch.from_numpy(syn.kitti_batch(700, 8, 4096)[0])
    xyz, feat = pts[:, :3].contiguous(), pts[:, 3:].contiguous()
    torch.manual_seed(2)
    orc = T.RefSetAbstraction(1024, 0.1, 32, 9, [32, 32, 64], False)
    mod = U.PointNetSetAbstraction(1024, 0.1, 32, 9, [32, 32, 64], False)
    mod.load_state_dict(orc.state_dict())
    mod.to(dev).train()
    orc.train()
    f_ref, f_gpu = feat.clone().requires_grad_(True), feat.clone().to(dev).requires_grad_(True)
    torch.manual_seed(3)
    nx_ref, a = orc(xyz, f_ref)
    torch.manual_seed(3)
    nx, b = mod(xyz.to(dev), f_gpu)
    assert torch.equal(nx_ref, nx.cpu())                            # same FPS picks -> identical centroids
    assert float((a.detach() - b.detach().cpu()).abs().max()) <= FWD_TOL
    gw = torch.randn(a.shape, generator=torch.Generator().manual_seed(4))
    (a * gw).sum().backward()
    (b * gw.to(dev)).sum().backward()
    assert relmax(f_gpu.grad.cpu().numpy(), f_ref.grad.numpy()) <= GRAD_TOL
    for (n, p), (_, q) in zip(orc.named_parameters(), mod.named_parameters()):
        if "conv" in n and n.endswith("bias"):
            continue
        assert relmax(q.grad.cpu().numpy(), p.grad.numpy()) <= GRAD_TOL, n


@pytest.mark.parametrize("direct", [False, True])
def test_fused_bn_tails_match_standalone_launches(dev, direct, monkeypatch):
    """pn2_bn_finalize_tail / pn2_bn_coef_tail (statistics -> affine block / coefficients inside the producer kernel)
    against the stand-alone pn2_bn_finalize / pn2_bn_bwd_coef launches: same arithmetic, so the same network state
    after two training steps (running statistics, num_batches_tracked) and the same gradients up to the
    atomics-order noise of the reductions."""
    from pointnet12_amd import parallel
    g = golden("g6_nets.npz")
    pts = torch.from_numpy(g["points"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    res = []
    try:
        for fused in (False, True):
            monkeypatch.setattr(U, "FUSED_BN_TAILS", fused)
            torch.manual_seed(int(g["init_seed"]))
            net = M.PointNet2SemSegMsg(13, 6)
            net.drop1.p = 0.0
            net.to(dev).train()
            bucket = parallel.FlatGradBucket(net, direct=direct)
            losses = []
            for _ in range(2):
                bucket.zero()
                torch.manual_seed(5)
                lp = net(pts)
                loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1))
                loss.backward()
                losses.append(float(loss))
            bufs = {k: v.clone() for k, v in net.named_buffers()}
            res.append((losses, bucket.flat.clone(), bufs))
    finally:
        U.set_direct_grad_accumulation(False)
    (la, ga, ba), (lb, gb, bb) = res
    assert np.allclose(la, lb, rtol=0, atol=2e-5), (la, lb)
    assert abs(float(ga.norm()) - float(gb.norm())) <= 5e-3 * float(ga.norm())
    assert float((ga - gb).abs().max()) <= 2e-2 * float(ga.abs().max())
    for k in ba:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == int(bb[k]) == 2, k
        else:
            assert float((ba[k] - bb[k]).abs().max()) <= 1e-5 * max(1.0, float(ba[k].abs().max())), k


@pytest.mark.parametrize("kind,direct", [("msg", True), ("msg", False), ("ssg", True)])
def test_consumer_side_batchnorm_matches_standalone_launches(dev, kind, direct, monkeypatch):
    """Round 4 (ABI 8): statistics -> affine block and reductions -> coefficients as a prologue of the first kernel that reads
    the block (pn2_bn_lazy / pn2_bn_coef_lazy: every workgroup recomputes it from the producer's finished sums) against the
    stand-alone pn2_bn_finalize / pn2_bn_bwd_coef launches.  Same fp64 arithmetic per channel, so after two training steps the
    same losses, gradients (up to the atomics-order noise of the reductions), running statistics and num_batches_tracked (the
    prologue's once-per-launch writes: a double update would show here) -- and the launches are really gone."""
    from pointnet12_amd import _lib, parallel
    g = golden("g6_nets.npz")
    pts = torch.from_numpy(g["points"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    res = []
    try:
        for lazy in (False, True):
            monkeypatch.setattr(U, "LAZY_BN", lazy)
            torch.manual_seed(int(g["init_seed"]))
            net = M.PointNet2SemSegMsg(13, 6) if kind == "msg" else M.PointNet2SemSeg(13, 6)
            net.drop1.p = 0.0
            net.to(dev).train()
            bucket = parallel.FlatGradBucket(net, direct=direct)
            losses = []
            with _lib.call_profile() as calls:
                for _ in range(2):
                    bucket.zero()
                    torch.manual_seed(5)
                    lp = net(pts)
                    loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1))
                    loss.backward()
                    losses.append(float(loss))
                torch.cuda.synchronize()
                names = [c[0] for c in calls]
            bufs = {k: v.clone() for k, v in net.named_buffers()}
            res.append((losses, bucket.flat.clone(), bufs, names))
    finally:
        U.set_direct_grad_accumulation(False)
    (la, ga, ba, na), (lb, gb, bb, nb) = res
    n_bn = sum(1 for k in ba if k.endswith("num_batches_tracked"))
    assert na.count("pn2_bn_finalize") == 2 * n_bn and na.count("pn2_bn_bwd_coef") == 2 * n_bn
    assert nb.count("pn2_bn_finalize") == 0
    assert nb.count("pn2_bn_bwd_coef") == 0           # every first consumer of a coefficient block fills it itself
    assert np.allclose(la, lb, rtol=0, atol=2e-5), (la, lb)
    assert abs(float(ga.norm()) - float(gb.norm())) <= 5e-3 * float(ga.norm())
    assert float((ga - gb).abs().max()) <= 2e-2 * float(ga.abs().max())
    for k in ba:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == int(bb[k]) == 2, k
        else:
            assert float((ba[k] - bb[k]).abs().max()) <= 1e-5 * max(1.0, float(ba[k].abs().max())), k


def test_segmented_backward_matches_elementwise_atomics(dev, monkeypatch):
    """The backward scatter-adds (3-NN interpolation, factorised first layer) as segmented reductions over the
    target-sorted index (pn2_invert_index + pn2_*_bwd_seg, the default) against the element-wise atomic kernels:
    same sums in another order, on a KITTI-shaped cloud (targets with hundreds of members next to empty ones)."""
    from pointnet12_amd import synthetic as syn
    pts, _ = syn.kitti_batch(321, 3, 2048)
    pts = torch.from_numpy(pts)
    xyz, feat = pts[:, :3].contiguous().to(dev), torch.randn(3, 64, 2048, generator=torch.Generator().manual_seed(1)).to(dev)
    grads = []
    for seg in (True, False):
        monkeypatch.setattr(U, "GATHER_BACKWARD", seg)
        torch.manual_seed(11)
        sa = U.PointNetSetAbstraction(256, 0.2, 32, 64 + 3, [64, 64, 128], False).to(dev).train()
        fp = U.PointNetFeaturePropagation(128 + 64, [128, 64]).to(dev).train()
        f = feat.clone().requires_grad_(True)
        torch.manual_seed(12)
        new_xyz, new_feat = sa(xyz, f)                       # factorised first layer (D = 64 >= 32)
        out = fp(xyz, new_xyz, f, new_feat)                  # 3-NN interpolation back onto the 2048 points
        (out * torch.linspace(-1, 1, out.numel(), device=dev).view_as(out)).sum().backward()
        grads.append([f.grad.clone()] + [p.grad.clone() for p in list(sa.parameters()) + list(fp.parameters())])
    for a, b in zip(*grads):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-4 * scale


def test_graphed_step_benchmark_size_without_prefetch(dev):
    """The whole-step graph with the geometry INSIDE the step (no prefetch branch) at B=8 x 4096: every buffer the
    step clears is allocated during the capture here.  (hipMemsetAsync nodes on such allocations faulted at replay
    -- "write access to a read-only page" -- while the 2 x 1024 case above passed; the library clears with kernels.)"""
    from pointnet12_amd import parallel, synthetic as syn
    from pointnet12_amd.graph import GraphedStep
    pts_np, labels_np = syn.kitti_batch(40, 8, 4096)
    pts, labels = torch.from_numpy(pts_np).to(dev), torch.from_numpy(labels_np).to(dev)
    torch.manual_seed(3)
    net = M.PointNet2SemSeg(13, 6).to(dev).train()
    bucket = parallel.FlatGradBucket(net)

    def compute():
        bucket.zero()
        lp = net(pts)
        loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1))
        loss.backward()
        return loss
    step = GraphedStep(compute, dev, warmup=1)
    losses = [float(step()) for _ in range(4)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and 1.0 < losses[-1] < 4.0, losses
    assert bool(torch.isfinite(bucket.flat).all()) and float(bucket.flat.abs().max()) > 0


def test_full_size_batch_permutation_equivariance(dev):
    """BASELINE cfg3 size (MSG-SemSeg B=16 x 4096 x 9, train mode): clouds are independent units apart from the
    BatchNorm batch statistics, so permuting the clouds of the batch -- with their FPS start indices
    -- must permute the outputs and the input gradients and leave every parameter gradient unchanged, up to the
    summation order of the statistics and the gradient atomics.  Needs no oracle: a size-independent property of the
    whole forward + backward path."""
    from pointnet12_amd import pointnet2, synthetic as syn
    torch.manual_seed(5)
    net = pointnet2.PointNet2SemSegMsg(13, 6).to(dev).train()
    pts_np, _ = syn.kitti_batch(0, 16, 4096)
    pts = torch.from_numpy(pts_np).to(dev)
    g = torch.Generator().manual_seed(9)
    perm = torch.randperm(16, generator=g).to(dev)
    starts = [torch.randint(0, 4096, (16,), generator=g).to(dev), torch.randint(0, 512, (16,), generator=g).to(dev)]
    proj = torch.randn(16, 128, 4096, generator=g).to(dev) / 64

    class Feed:
        def __init__(self, seq):
            self.seq, self.i = seq, 0

        def take(self, B, N, device):
            self.i += 1
            return self.seq[self.i - 1]

    def run(x, st, w):
        x = x.clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        U.set_fps_start_feed(Feed(st))
        try:
            out = net.features(x)
        finally:
            U.set_fps_start_feed(None)
        (out * w).sum().backward()
        return out.detach(), x.grad.detach(), [p.grad.detach().clone() for p in net.parameters() if p.grad is not None]

    out_a, gin_a, gp_a = run(pts, starts, proj)
    out_b, gin_b, gp_b = run(pts[perm], [s[perm] for s in starts], proj[perm])
    assert out_a.shape == (16, 128, 4096)
    assert float((out_b - out_a[perm]).abs().max()) <= 2e-5 * max(1.0, float(out_a.abs().max()))
    scale = float(gin_a[:, 3:].abs().max())                 # xyz columns carry no gradient
    assert scale > 0
    d = (gin_b - gin_a[perm])[:, 3:]
    # statistics summed in another order flip a few max-pool winners / ReLU signs: whole rows of gradient move
    # (measured 3.4e-3 of the norm; a wrong pairing of clouds gives O(1))
    assert float(d.norm()) <= 1e-2 * float(gin_a[:, 3:].norm())
    assert len(gp_a) == len(gp_b) and len(gp_a) > 60
    typical = float(torch.stack([a.norm() for a in gp_a]).median())
    for a, b in zip(gp_a, gp_b):
        # conv biases in front of a BatchNorm have a mathematically zero gradient: what is there is rounding noise,
        # measured against the typical gradient norm instead of its own
        assert float((a - b).norm()) <= 2e-2 * float(a.norm()) + 1e-4 * typical, a.shape


def _randomise_bn(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)


@pytest.mark.parametrize("tag", ["sa32", "sa16", "sa64", "msg", "all128", "fp", "fp_s1"])
def test_fused_eval_module_vs_oracle(dev, tag):
    """pn2_fused_eval (csrc/eval.hip): under .eval() + no_grad a module is ONE launch -- gather, centre, concat, BatchNorm
    folded into the weights, ReLU, max over the neighbours -- against the oracle module in eval mode (<= 1e-5), and
    against this package's own layer-by-layer eval kernels."""
    from oracle import torch_ref as T
    from pointnet12_amd import synthetic as syn
    pts = torch.from_numpy(syn.kitti_batch(800, 2, 1024)[0])
    xyz, feat = pts[:, :3].contiguous(), pts[:, 3:].contiguous()
    torch.manual_seed(3)
    if tag.startswith("sa"):
        K = int(tag[2:])
        orc, mod = T.RefSetAbstraction(128, 0.3, K, 9, [32, 48, 64], False), U.PointNetSetAbstraction(128, 0.3, K, 9, [32, 48, 64], False)
        args = (xyz, feat)
    elif tag == "msg":
        cfg = (64, [0.2, 0.4, 0.8], [16, 32, 128], 6, [[16, 32], [32, 48, 64], [32, 196]])
        orc, mod = T.RefSetAbstractionMsg(*cfg), U.PointNetSetAbstractionMsg(*cfg)
        args = (xyz, feat)
    elif tag == "all128":
        orc, mod = T.RefSetAbstraction(None, None, None, 9, [64, 256, 520], True), U.PointNetSetAbstraction(None, None, None, 9, [64, 256, 520], True)
        args = (xyz[:, :, :128].contiguous(), feat[:, :, :128].contiguous())
    else:
        S = 1 if tag == "fp_s1" else 128
        orc, mod = T.RefFeaturePropagation(6 + 24, [40, 16]), U.PointNetFeaturePropagation(6 + 24, [40, 16])
        p2 = torch.randn(2, 24, S, generator=torch.Generator().manual_seed(8))
        x2 = xyz[:, :, :S].contiguous() if S > 1 else torch.zeros(2, 3, 1)
        args = (xyz, x2, feat, p2)
    _randomise_bn(orc, 5)
    mod.load_state_dict(orc.state_dict())
    mod.to(dev).eval()
    orc.eval()
    with torch.no_grad():
        torch.manual_seed(9)
        a = orc(*args)
        torch.manual_seed(9)
        b = mod(*[t.to(dev) for t in args])
        calls = []
        with U._lib.call_profile() as prof:
            torch.manual_seed(9)
            mod(*[t.to(dev) for t in args])
            calls = [c[0] for c in prof]
        U.FUSED_EVAL = False
        try:
            torch.manual_seed(9)
            c = mod(*[t.to(dev) for t in args])
        finally:
            U.FUSED_EVAL = True
    a, b, c = (x[-1] if isinstance(x, tuple) else x for x in (a, b, c))
    assert "pn2_fused_eval" in calls and "pn2_conv1x1_fwd" not in calls, calls
    assert tuple(a.shape) == tuple(b.shape)
    assert float((a - b.cpu()).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
    assert float((c - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))


def test_eval_network_fused_vs_oracle(dev):
    """PointNet2SemSeg(19, 1) in eval mode on one 8192-point cloud (the viewer-loop shape, pcdvis.py:118-136), fused eval
    path, against the oracle net in eval mode."""
    from oracle import torch_ref as T
    from pointnet12_amd import synthetic as syn
    torch.manual_seed(2)
    orc = T.RefSSGSemSeg(19, 1)
    _randomise_bn(orc, 7)
    net = M.PointNet2SemSeg(19, 1)
    net.load_state_dict(orc.state_dict())
    net.to(dev).eval()
    orc.eval()
    pts = torch.from_numpy(syn.kitti_batch(55, 1, 8192, channels=4)[0])
    with torch.no_grad():
        torch.manual_seed(4)
        a = orc(pts)
        torch.manual_seed(4)
        b = net(pts.to(dev))
    assert float((a - b.cpu()).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))


def test_fused_eval_follows_training(dev):
    """Train -> eval -> train -> eval, the reference's epoch loop (pcdseg.py:69,190): the eval-mode fold (BatchNorm into the
    conv weights) must follow writes the HIP library makes through raw pointers -- running statistics by pn2_bn_finalize,
    parameters by pn2_adam_step -- which tensor._version does not see.  Checked against the layer-by-layer eval kernels
    (PN2_FUSED_EVAL=0, which read the live parameters), and through a hipGraph that captured the fused launch BEFORE the
    training steps (its baked-in weight pointers must stay valid and show the refreshed fold)."""
    from pointnet12_amd import optim, synthetic as syn
    torch.manual_seed(11)
    mod = U.PointNetSetAbstraction(128, 0.3, 32, 9, [32, 48, 64], False).to(dev)
    pts = torch.from_numpy(syn.kitti_batch(801, 2, 1024)[0]).to(dev)
    xyz, feat = pts[:, :3].contiguous(), pts[:, 3:].contiguous()
    start = torch.zeros(2, dtype=torch.int64, device=dev)
    opt = optim.Adam(mod.parameters(), lr=1e-2)

    def evaluate(fused):
        mod.eval()
        U.FUSED_EVAL = fused
        try:
            with torch.no_grad():
                return mod(xyz, feat, fps_start=start)[1].clone()
        finally:
            U.FUSED_EVAL = True

    first = evaluate(True)
    assert float((first - evaluate(False)).abs().max()) <= 1e-5 * max(1.0, float(first.abs().max()))
    # a graph that captured the fused launch now
    mod.eval()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side), torch.no_grad():
        mod(xyz, feat, fps_start=start)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            g_out = mod(xyz, feat, fps_start=start)[1]
    torch.cuda.current_stream(dev).wait_stream(side)
    for _ in range(3):
        mod.train()
        opt.zero_grad()
        out = mod(xyz, feat, fps_start=start)[1]
        (out * out).mean().backward()
        opt.step()
    after = evaluate(True)
    ref = evaluate(False)
    assert float((after - first).abs().max()) > 1e-3, "three Adam steps at lr 1e-2 must move the eval output"
    assert float((after - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    graph.replay()
    torch.cuda.synchronize()
    assert float((g_out - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("K,co,D,xyz_first", [(16, 32, 6, False), (32, 64, 6, True), (64, 64, 3, False), (32, 64, 9, False)])
def test_gather_conv_first_layer_equals_group_then_gemm(dev, K, co, D, xyz_first):
    """pn2_group_conv_fwd (gather + first conv in one launch, the sa1 stacks) against pn2_group followed by the GEMM: same
    grouped rows bit for bit, outputs and every gradient (weights, BatchNorm, the gathered features) within the GEMM-vs-fma
    rounding of the first layer."""
    import torch.nn as nn
    from pointnet12_amd import _lib
    gen = torch.Generator().manual_seed(K + co + D)
    B, N, S = 2, 2048, 256
    xyz = torch.rand(B, N, 3, generator=gen).to(dev)
    pts = torch.randn(B, N, D, generator=gen).to(dev)
    new_xyz = xyz[:, :S].contiguous()
    idx = torch.randint(0, N, (B, S, K), generator=gen).to(dev)
    convs = nn.ModuleList([nn.Conv2d(3 + D, co, 1), nn.Conv2d(co, 64, 1)]).to(dev)
    bns = nn.ModuleList([nn.BatchNorm2d(co), nn.BatchNorm2d(64)]).to(dev)
    res = {}
    for flag in (True, False):
        U.GATHER_CONV = flag
        try:
            for p in list(convs.parameters()) + list(bns.parameters()):
                p.grad = None
            f = pts.clone().requires_grad_(True)
            with _lib.call_profile() as calls:
                out = U.grouped_mlp(xyz, f, new_xyz, idx, xyz_first, convs, bns, True)
            gw = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(dev)
            (out * gw).sum().backward()
            res[flag] = (out.detach().clone(), f.grad.clone(), [p.grad.clone() for p in list(convs.parameters()) + list(bns.parameters())],
                         [c[0] for c in calls])
        finally:
            U.GATHER_CONV = True
    assert "pn2_group_conv_fwd" in res[True][3] and "pn2_group" not in res[True][3]
    assert "pn2_group" in res[False][3] and "pn2_group_conv_fwd" not in res[False][3]
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-5 * float(res[False][0].abs().max())
    for a, b in zip([res[True][1]] + res[True][2], [res[False][1]] + res[False][2]):
        assert float((a - b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1e-6)


def test_forward_is_bit_reproducible_and_the_backward_noise_is_bounded(dev):
    """What is reproducible run to run, measured (HISTORY.md round 6, tools/exp/determinism_probe.py): the FORWARD of a training step --
    every module's output, the log-probabilities, the loss -- is bit-identical (the statistics are fp64 atomics whose rounding never
    reached an fp32 bit in any run); the BACKWARD is not (fp32 atomics: the weight-gradient flush, the 3-NN interpolation's and the
    grouping's scatter-adds), and differs by <= 3.2e-6 of a gradient's largest element.  There is no deterministic mode (VERDICT r5
    #6: about 80 atomic sites in ten files); this test holds the forward to the bit and the backward to 1e-5."""
    from pointnet12_amd import pointnet2 as M
    from pointnet12_amd import synthetic as syn
    from pointnet12_amd.loss import nll_loss
    torch.manual_seed(0)
    net = M.PointNet2SemSegMsg(13, 6).to(dev).train()
    pts_np, lab_np = syn.kitti_batch(0, 4, 4096)
    pts, labels = torch.from_numpy(pts_np).to(dev), torch.from_numpy(lab_np).to(dev)
    acts = {}
    for name, mod in net.named_children():
        mod.register_forward_hook(lambda m, i, o, name=name: acts.setdefault(name, []).append((o[1] if isinstance(o, tuple) else o).detach().clone()))
    runs = []
    for r in range(3):
        torch.manual_seed(123)                      # the FPS start draws and the dropout mask
        net.zero_grad(set_to_none=True)
        lp = net(pts)
        loss = nll_loss(lp.reshape(-1, lp.shape[-1]), labels.reshape(-1))
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), lp.detach().clone(), [p.grad.detach().clone() for p in net.parameters()]))
    for name, lst in acts.items():
        assert all(torch.equal(lst[0], t) for t in lst[1:]), "forward output of %s differs between two runs" % name
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    names = [n for n, _ in net.named_parameters()]
    for r in runs[1:]:
        for n, a, b in zip(names, runs[0][2], r[2]):
            scale = float(a.abs().max())
            if scale < 1e-7:                        # (a gradient that is zero in exact arithmetic: conv biases before a BatchNorm, ...)
                continue
            assert float((a - b).abs().max()) <= 1e-5 * scale, n
