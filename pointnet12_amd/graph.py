"""Whole-step hipGraph capture: forward + loss + backward of a PointNet++ network as ONE graph launch.

A training step of MSG-SemSeg is ~400 kernel launches (SSG ~470), most of them tens of microseconds long; issued
one by one from Python the host becomes the bottleneck as soon as the kernels are fast (SSG-SemSeg B=16: 7 ms of
device work per step, 10 ms of launching).  MI355X-first means HIP streams and graphs, not a tracing compiler:
the step is captured once (``torch.cuda.graph`` drives hipStreamBeginCapture; the C-ABI launches of
libpn2_hip.so go to torch's current stream and are recorded like any other kernel) and replayed per step.

The only per-step host input of the path is the FPS start index (reference pointnet_util.py:75: one
``torch.randint`` on the CPU generator per sampling call).  ``FpsStartFeed`` keeps that contract under replay:
during capture every draw site receives a slice of one device buffer; before each replay the host draws the
same ``randint(0, N, (B,))`` sequence, in the same order, into a pinned staging buffer and enqueues one H2D
copy ahead of the graph.
"""
import os

import torch

from . import pointnet_util as U

_GEO_FIRST = os.environ.get("PN2_GEO_FIRST", "1") == "1"      # capture order of the two branches (the executor's launch order follows it)
# GraphedStep(fork_in_step=True): the geometry branch of the captured step (the NEXT batch's FPS / ball query / 3-NN: 0.55 ms on
# 16 .. 64 CUs) starts where the step function calls graph.fork_point() -- bench.py (B = 16 x 4096) calls it between forward and
# backward, so the branch
# runs under the head / FP / sa4 / sa3 backward launches that do not fill the chip instead of under sa1's forward kernels, whose
# one-workgroup-per-CU grids lose the CUs the FPS workgroups hold (same box, round 4: MSG 5.85 / 5.84 -> 5.75 / 5.78 ms, SSG within
# noise; cfg2, whose step IS the FPS chain, and cfg5 lose with it: 0.68 -> 0.76 ms, 6.30 -> 6.40 ms -- hence opt-in).
# PN2_GEO_FORK_LATE=0: fork at the top of the step whatever the caller asked for (round 3's order; A/B runs).
_GEO_FORK_LATE = os.environ.get("PN2_GEO_FORK_LATE", "1") == "1"
_fork_cb = None


def fork_point():
    """Called by a step function at the place where the next batch's geometry branch may start (no-op outside such a capture)."""
    if _fork_cb is not None:
        _fork_cb()



class FpsStartFeed:
    def __init__(self, device, capacity=4096, ring=4):
        self.device = device
        self.dev = torch.zeros(capacity, dtype=torch.int64, device=device)
        self.host = [torch.zeros(capacity, dtype=torch.int64).pin_memory() for _ in range(ring)]
        self.events = [None] * ring
        self.slots = []            # (offset, B, N) in draw order
        self.used = 0
        self.turn = 0

    def take(self, B, N, device):
        """Called at capture time from draw_fps_start: reserve B slots, return their device view."""
        if self.used + B > self.dev.numel():
            raise RuntimeError("FpsStartFeed capacity exceeded")
        off = self.used
        self.slots.append((off, B, N))
        self.used += B
        return self.dev[off:off + B]

    def stage(self):
        """Draw this step's start indices (CPU generator, reference order) and enqueue their upload."""
        h = self.host[self.turn]
        if self.events[self.turn] is not None:
            self.events[self.turn].synchronize()       # the copy that last read this staging buffer is done
        for off, B, N in self.slots:
            h[off:off + B] = torch.randint(0, N, (B,), dtype=torch.long)
        if self.used:
            self.dev[:self.used].copy_(h[:self.used], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[self.turn] = ev
        self.turn = (self.turn + 1) % len(self.host)


def capture_error_mode():
    """`global` (torch's default: an unsafe HIP call from ANY thread fails the capture) unless a process group is alive: the RCCL
    watchdog thread polls its work events with hipEventQuery, and a poll that lands inside a capture window raised
    hipErrorStreamCaptureUnsupported in that thread and aborted the rank (round 6: 1 of 12 one-rank `bench.py --gpus 1` launches under
    torchrun; `tests/test_parallel_gpu.py`).  With a process group the capture restricts its own thread only."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return "thread_local"
    except Exception:                                   # (a torch build without distributed support)
        pass
    return "global"


class GraphedStep:
    """``step = GraphedStep(fn)``; ``loss = step()`` replays ``fn`` (zero grads + forward + loss + backward).

    ``fn`` must be capture-safe: static input tensors, no host synchronisation, gradients accumulated into
    pre-existing ``.grad`` tensors (parallel.FlatGradBucket does that).  The returned loss tensor is static
    (overwritten by every replay).

    The ``warmup`` eager calls of ``fn`` before the capture are REAL steps: BatchNorm running statistics and
    ``num_batches_tracked`` advance, and if ``fn`` contains an optimizer step the parameters move.  Capture a step
    that should start from a pristine state with ``warmup=0`` after warming the allocator up some other way, or
    snapshot / restore ``state_dict()`` around the constructor.

    ``geometry_fn`` (optional): a callable running only the network's geometry on the NEXT batch's static input
    (e.g. ``lambda: net.features(points)``; under a recording GeometryTape the modules skip all feature work).
    The captured graph then has two branches: the main stream runs ``fn`` on the geometry recorded one step
    earlier, a side stream runs FPS / ball query / 3-NN for the following batch -- the 1 000-iteration FPS latency
    chain occupies 16 of the 256 CUs and disappears under the MLP kernels instead of heading every step.  The FPS
    start draws keep the reference's order (one set per batch, sa1 first).

    ``fork_in_step``: the side branch starts where ``fn`` calls ``graph.fork_point()`` (once) instead of at the top of the step --
    e.g. between forward and backward, so that the geometry chain runs under the backward stages that do not fill the chip.
    Worth it when that chain is short against what follows the fork point (B = 16 x 4096 points: MSG-SemSeg 5.84 -> 5.76 ms);
    with a long chain (one set-abstraction level alone, 65 536-point clouds) keep the default.
    """

    def __init__(self, fn, device, warmup=3, geometry_fn=None, fork_in_step=False):
        if fork_in_step and geometry_fn is None:
            raise ValueError("fork_in_step=True needs a geometry_fn: without a geometry branch there is nothing to fork "
                             "(graph.fork_point() in the step function would be a no-op)")
        self.fork_point_reached = None           # prefetch captures with fork_in_step: did fn() call graph.fork_point()?
        if geometry_fn is not None:
            self._init_prefetch(fn, geometry_fn, device, warmup, fork_in_step and _GEO_FORK_LATE)
            return
        self.fn = fn
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):                  # eager warm-up off the default stream (allocator, lazy inits)
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.feed = FpsStartFeed(device)
        self.graph = torch.cuda.CUDAGraph()
        U.set_fps_start_feed(self.feed)
        U.set_capture_scope(object())
        try:
            with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode()):
                self.loss = fn()
        finally:
            U.set_capture_scope(None)
            U.set_fps_start_feed(None)

    def _init_prefetch(self, fn, geometry_fn, device, warmup, fork_in_step=False):
        self.fn = fn
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        # geometry of the first batch, eagerly (its FPS starts are the first draw set, as in sequential execution)
        cur = U.GeometryTape()
        U.set_geometry_tape(cur)
        try:
            with torch.no_grad():
                geometry_fn()
        finally:
            U.set_geometry_tape(None)
        torch.cuda.synchronize(device)
        # Two graphs, replayed alternately, double-buffer the geometry: graph A runs the step on tape T0 while its
        # side branch records the next batch's geometry into T1; graph B runs on T1 and records into T0's tensors.
        # (A single graph has to copy the prefetched tape over the replayed one at the very end of the step --
        # 13 small copies that sit on the critical tail of the main stream, ~0.1 ms.)
        self.graphs, self.feeds, self.losses = [], [], []
        self.parity = 0
        # The tapes are OWNED by this object: T0 was recorded eagerly, i.e. its tensors live in the ordinary allocator pool, and
        # both graphs hold raw pointers to them (A reads them, B's side branch writes the next geometry into them).  Left to a
        # local variable they were released when the constructor returned and the allocator handed the same memory to whatever
        # the caller allocated next -- which the following replays then overwrote with coordinates and indices (found in round
        # 3 by a test that kept small result tensors across replays; bench.py allocates nothing between replays).
        tapes = [cur, None]
        self._tapes = [cur]
        pool = None
        for which in (0, 1):
            feed = FpsStartFeed(device)
            graph = torch.cuda.CUDAGraph()
            read = tapes[which]
            write = U.GeometryTape() if which == 0 else U.GeometryTape(into=tapes[0].items)
            geo_stream = torch.cuda.Stream(device=device)
            U.set_fps_start_feed(feed)
            U.set_capture_scope(object())
            try:
                with torch.cuda.graph(graph, pool=pool, capture_error_mode=capture_error_mode()):
                    main = torch.cuda.current_stream(device)
                    geo_stream.wait_stream(main)

                    def geometry_branch():                         # branch 2: next batch's geometry
                        with torch.cuda.stream(geo_stream):
                            U.set_geometry_tape(write)
                            with torch.no_grad():
                                geometry_fn()

                    forked = [False]

                    def fork_now():                                # (fork_point(): the step function chose where the branch starts)
                        if forked[0]:
                            return
                        forked[0] = True
                        geo_stream.wait_stream(torch.cuda.current_stream(device))
                        keep = U.get_geometry_tape()
                        try:
                            geometry_branch()
                        finally:                                   # (a raising geometry_fn must not leave the step on the wrong tape)
                            U.set_geometry_tape(keep)

                    global _fork_cb
                    if _GEO_FIRST and not fork_in_step:
                        forked[0] = True
                        geometry_branch()
                    read.rewind("replay")                          # branch 1: this batch on the recorded geometry
                    U.set_geometry_tape(read)
                    _fork_cb = fork_now if fork_in_step else None
                    try:
                        loss = fn()
                    finally:
                        _fork_cb = None
                    U.set_geometry_tape(None)
                    if fork_in_step:
                        self.fork_point_reached = forked[0]
                    if not forked[0]:
                        if fork_in_step:
                            import warnings
                            warnings.warn("GraphedStep(fork_in_step=True): the step function never called graph.fork_point(); the "
                                          "geometry branch was captured behind it and depends on the TOP of the step only "
                                          "(the PN2_GEO_FIRST=0 order), not where the caller meant it to start")
                        geometry_branch()
                        U.set_geometry_tape(None)
                    main.wait_stream(geo_stream)
            finally:
                U.set_capture_scope(None)
                U.set_geometry_tape(None)
                U.set_fps_start_feed(None)
            self._tapes.append(write)
            if which == 0:
                tapes[1] = write
                pool = graph.pool()
            self.graphs.append(graph)
            self.feeds.append(feed)
            self.losses.append(loss)
        self.graph, self.feed, self.loss = self.graphs[0], self.feeds[0], self.losses[0]

    def __call__(self):
        if getattr(self, "graphs", None):
            i = self.parity
            self.parity ^= 1
            self.feeds[i].stage()
            self.graphs[i].replay()
            U.bump_param_generation()            # whatever the captured step wrote (BN buffers, an optimiser step)
            U.bump_data_generation()
            return self.losses[i]
        self.feed.stage()
        self.graph.replay()
        U.bump_param_generation()
        U.bump_data_generation()
        return self.loss


def _flatten(items):
    out = []
    for it in items:
        out.extend(it if isinstance(it, tuple) else (it,))
    return out
