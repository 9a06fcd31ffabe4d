"""Data-parallel plumbing: one process per GPU, batch dimension sharded contiguously across ranks,
one flat gradient bucket per network all-reduced (sum) over RCCL (torch.distributed backend
"nccl" on ROCm) between backward and the fused Adam, which applies the 1/world_size scale
(SURVEY.md 8(e)).  BatchNorm statistics stay per-rank (what DDP does).  The reference has no working
multi-GPU behaviour to match (its nn.DataParallel path dereferences attributes the wrapper does not
proxy, dexgangrasp.py:127-129 vs :770-772), so this defines it.

The same code runs on the CPU with the gloo backend (tests/test_parallel_cpu.py, world_size 2).
"""
from __future__ import annotations

import os
import sys

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def forced():
    """DGG_FORCE_DISTRIBUTED=1: treat a ONE-rank process group as distributed, so that the collective path (RCCL
    communicator and stream, feeder stream, gradient-ready events, split optimiser step behind the work handles) runs
    on a 1-GPU box with the real backend.  For tests and overhead measurements; the sums are those of one rank."""
    return os.environ.get("DGG_FORCE_DISTRIBUTED") == "1"


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def rehearsal():
    """DGG_REHEARSE=1: run the multi-rank code path on ONE GPU — every rank on cuda:0, collectives over gloo.
    For tests of the engine's collective / stream ordering on a 1-GPU box; never for reported numbers."""
    return os.environ.get("DGG_REHEARSE") == "1"


def local_device_index():
    """Index of the GPU this process drives: LOCAL_RANK under torchrun (0 in a rehearsal)."""
    _, world, local = env_rank_world()
    return 0 if (world == 1 or rehearsal()) else local


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (no-op for WORLD_SIZE=1) and make this
    rank's GPU the current device BEFORE anything else touches HIP.  Returns (rank, world, device index)."""
    rank, world, _ = env_rank_world()
    dev = local_device_index()
    if torch.cuda.is_available():
        torch.cuda.set_device(dev)
    if (world > 1 or forced()) and not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "gloo" if (rehearsal() or not torch.cuda.is_available()) else "nccl"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, dev


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def rank():
    return dist.get_rank() if is_distributed() else 0


def shard_rows(t, r=None, w=None):
    """Contiguous shard r of w along dim 0 (global batch must divide evenly)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    n = t.shape[0]
    if n % w != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (n, w))
    per = n // w
    return t[r * per:(r + 1) * per]


def shard_batch(batch, r=None, w=None):
    return {k: (shard_rows(v, r, w) if torch.is_tensor(v) else v) for k, v in batch.items()}


def allreduce_sum_(flat_grads, async_op=False):
    """In-place sum over ranks of one flat gradient bucket (27.9 / 28.0 / 42.6 MB fp32 for G / D / E).
    Returns the work handle when async_op (so it can overlap with independent compute)."""
    if not is_distributed():
        return None
    return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, async_op=async_op)


_eva_group = None


def evaluator_group():
    """DGG_EVA_GROUP=1: a process group (communicator, stream) of its own for the evaluator's gradient buckets.  RCCL runs
    the collectives of ONE group in issue order on one stream; the evaluator's 42.6 MB, issued during its backward
    (0.7-1.2 ms into the step), sit in front of the generator's first bucket (~1.5 ms) if the ring is slower than that —
    whether they are out of the way by then depends on the node's real ring bandwidth, which no 1-GPU box can show
    (DESIGN.md 6), so the first run on an 8-GPU node can A/B the two layouts with this switch.  Collective on every rank
    (dist.new_group): all ranks must call it at the same point — engine._arm_reducers does, once.  None = default group."""
    global _eva_group
    if os.environ.get("DGG_EVA_GROUP", "0") != "1" or not is_distributed():
        return None
    if _eva_group is None:
        _eva_group = dist.new_group()
    return _eva_group


class BucketReducer:
    """All-reduce of one network's flat gradient buffer in the buckets a backward completes them in
    (FlatNet.grad_buckets / dgg_set_grad_events): the library records an event per bucket inside the backward's launch
    sequence, a feeder stream waits for it and issues that bucket's collective — so the all-reduce of the later
    ResBlocks (their W13 is ~half of the network) runs while the earlier blocks are still being back-propagated, and
    only the last bucket is exposed.  Same sums as one all-reduce of the whole buffer (element-wise, disjoint ranges)."""

    def __init__(self, flat, accum_only=False, group=None):
        from . import _lib
        self.flat = flat
        self.group = group            # process group of this network's collectives (None: the default group)
        self.accum_only = bool(accum_only)
        self.bounds = flat.grad_buckets()
        dev = flat.grads.device
        self.events = [torch.cuda.Event() for _ in self.bounds]
        for ev in self.events:            # materialise the handles
            ev.record(torch.cuda.current_stream(dev))
        n = len(self.events)
        arr = (_lib.C.c_void_p * n)()
        # bounds are in readiness order [block n-1, .., block 0]; the library indexes events by block
        for i, ev in enumerate(self.events):
            arr[n - 1 - i] = ev.cuda_event
        self._arr = arr
        self._lib = _lib
        self.armed = False

    def arm(self):
        """Register the events with the library (calling thread) — before the phase call whose backward fills them."""
        L = self._lib
        L.check(L.lib().dgg_set_grad_events(L.ptr(self.flat.grads), self._arr, len(self.events), int(self.accum_only)),
                "dgg_set_grad_events")
        self.armed = True

    def disarm(self):
        if self.armed:
            L = self._lib
            L.lib().dgg_set_grad_events(L.ptr(self.flat.grads), None, 0, 0)
            self.armed = False

    _shared_feeder = None   # one per process: every additional HIP stream risks sharing a hardware queue (see below)

    def reduce_async(self, feeder=None, timing=None, tag=""):
        """After the phase call: enqueue the bucket collectives behind their events.  Returns the work handles; wait()
        on each (a stream-level dependency for RCCL) before the optimiser step.

        timing (diagnostics, bench.py's untimed trace steps): a list that receives one (tag, bucket, bytes, begin, end)
        per collective — timed events on the feeding stream, `begin` once the bucket's gradients are ready, `end` once
        the collective has completed (the feeder waits for it: in this mode the buckets of one feeder go out one after
        the other, which is also the order RCCL runs them in).

        `feeder` is the stream that waits for the bucket events and hands each collective to RCCL's own stream.  The HIP
        runtime multiplexes a process's streams onto 4 hardware queues, in order within a queue: with the step's three
        compute streams, RCCL's stream and a feeder per network, the collectives of one network queued up behind the
        kernels of an unrelated stream (measured with a one-rank group, DGG_FORCE_DISTRIBUTED=1: 1.73 -> 2.07 ms per
        step, Adam(D) held back until the evaluator phase had drained).  So the caller lends a compute stream that is
        idle at that point (engine.joint_step); without one the first picked side stream feeds (pick_side_streams).
        feeder="current": no event waits at all — the collectives follow the current stream's tail (a phase that is not
        on the critical path and whose stream has nothing else to do)."""
        works = []

        def one(i, lo, hi, stream):
            e0 = None
            if timing is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            w = dist.all_reduce(self.flat.grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True, group=self.group)
            works.append(w)
            if timing is not None:
                w.wait()
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(stream)
                timing.append((tag, i, 4 * (hi - lo), e0, e1))
        if feeder == "current":
            cur = torch.cuda.current_stream(self.flat.grads.device)
            for i, (lo, hi) in enumerate(self.bounds):
                one(i, lo, hi, cur)
            return works
        if feeder is None:       # the serial entry points: the (idle) first side stream, on a queue of its own
            if BucketReducer._shared_feeder is None:
                BucketReducer._shared_feeder = side_streams_or_plain(self.flat.grads.device, 2)[0]
            feeder = BucketReducer._shared_feeder
        for i, ((lo, hi), ev) in enumerate(zip(self.bounds, self.events)):
            feeder.wait_event(ev)
            with torch.cuda.stream(feeder):
                one(i, lo, hi, feeder)
        return works


_picked = {}
pick_report = {}     # what the last pick_side_streams / side_streams_or_plain call found (bench.py prints it)


def side_streams_or_plain(dev, n=2):
    """pick_side_streams where the probe can run, plain torch streams where it cannot — and the choice on record
    (pick_report).  The probe needs torch.cuda._sleep (a private API): its absence is known before anything is issued, so
    every rank takes the same branch.  Under RCCL the probe issues collectives: an error in it is NOT swallowed — a rank
    that stopped half way would leave its peers waiting in a collective —; on a single GPU a failing probe
    falls back to plain streams (15-20 % slower when they land on a busy queue) and says so."""
    dev = torch.device(dev)
    if not hasattr(torch.cuda, "_sleep"):
        pick_report.update({"mode": "plain", "why": "torch.cuda._sleep unavailable: hardware queues not probed"})
        return [torch.cuda.Stream(device=dev) for _ in range(n)]
    rccl = is_distributed() and dist.get_backend() == "nccl"
    try:
        return pick_side_streams(dev, n)
    except RuntimeError as e:
        if rccl:
            raise
        pick_report.update({"mode": "plain", "why": "queue probe failed: %s" % str(e)[:120]})
        return [torch.cuda.Stream(device=dev) for _ in range(n)]


def _blocked_behind(cand, run_blocker, dev, sleep_cycles):
    """Does a tiny launch on stream `cand` have to wait for a long spin that `run_blocker` puts in front of SOMETHING
    else?  True = `cand` shares a hardware queue with that something."""
    main = torch.cuda.current_stream(dev)
    x = torch.zeros(64, device=dev)
    torch.cuda.synchronize(dev)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(cand)
    run_blocker(sleep_cycles)
    with torch.cuda.stream(cand):
        x.add_(1.0)
    t1.record(cand)
    torch.cuda.synchronize(dev)
    del main
    return t0.elapsed_time(t1)


def pick_side_streams(dev, n=2, candidates=8):
    """n side streams that share a hardware queue neither with the caller's current stream, nor (data parallel) with
    RCCL's stream, nor with each other.

    Why: the HIP runtime binds every stream to one of 4 hardware queues when it is created (round robin: streams 4 apart
    share one) and a queue runs its packets in order — ALSO the event waits of another stream bound to it.  RCCL's
    collectives sit on a stream of torch's pool behind a wait for the producing stream; whichever compute stream shares
    that queue stalls behind that wait (measured with a one-rank group: the prefetched real pass held back until the G
    phase had ended, or Adam(G) until the prefetch had, or the evaluator phase by 0.7 ms, depending on where the pool
    index fell: 1.97 / 2.03 / 2.12 ms per step against 1.79 with a free queue; tools/exp/queue_probe.py prints the
    matrix).  Nothing in the API reports the binding, so it is probed once per process: a long device-side spin is put
    in front of X and a tiny launch on Y — Y finishing only after the spin means X and Y share a queue.  The candidates
    are sorted into queue classes that way, the classes of the current stream and of RCCL's stream (an all-reduce issued
    behind a spin on the current stream) are excluded, and the side streams come from different remaining classes.
    ~30 ms."""
    rccl = is_distributed() and dist.get_backend() == "nccl"
    key = (torch.device(dev).index, n, rccl)     # a process group created later: probe again, RCCL's stream included
    if key in _picked:
        return _picked[key]
    dev = torch.device(dev)
    spin = 3_000_000          # ~1.3 ms at 2.4 GHz
    thresh = 0.5              # ms
    probe = torch.zeros(256, device=dev)
    dbg = os.environ.get("DGG_PICK_DEBUG") == "1"

    def block_main(cycles):
        torch.cuda._sleep(cycles)

    def block_main_and_rccl(cycles):   # RCCL's stream waits for the current stream's tail = the spin
        torch.cuda._sleep(cycles)
        block_main_and_rccl.work = dist.all_reduce(probe, op=dist.ReduceOp.SUM, async_op=True)

    def blocker_on(stream):
        def run(cycles):
            with torch.cuda.stream(stream):
                torch.cuda._sleep(cycles)
        return run

    if rccl:
        dist.all_reduce(probe)   # the communicator and its stream exist from here on
    pool = [torch.cuda.Stream(device=dev) for _ in range(candidates)]
    for cand in pool:        # first use of a stream creates its queue (milliseconds): not part of a measurement
        _blocked_behind(cand, lambda c: None, dev, 0)
    classes = []             # lists of streams sharing a queue
    for cand in pool:
        for cl in classes:
            if _blocked_behind(cand, blocker_on(cl[0]), dev, spin) > thresh:
                cl.append(cand)
                break
        else:
            classes.append([cand])
    with_main = [_blocked_behind(cl[0], block_main, dev, spin) > thresh for cl in classes]
    with_rccl = [False] * len(classes)
    if rccl:
        # one collective per CANDIDATE, whatever was measured: every rank issues the same number of them
        hit = [_blocked_behind(cand, block_main_and_rccl, dev, spin) > thresh for cand in pool]
        for i, (cl, m) in enumerate(zip(classes, with_main)):
            with_rccl[i] = (not m) and hit[pool.index(cl[0])]
    torch.cuda.synchronize(dev)
    free = [cl for cl, m, r in zip(classes, with_main, with_rccl) if not m and not r]
    if dbg:
        print("pick_side_streams: %d queue classes of sizes %s; current stream in %s, RCCL in %s; %d free"
              % (len(classes), [len(c) for c in classes], [i for i, m in enumerate(with_main) if m],
                 [i for i, r in enumerate(with_rccl) if r], len(free)), file=sys.stderr, flush=True)
    chosen = [cl[0] for cl in free[:n]]
    pick_report.update({"mode": "probed", "queue_classes": [len(c) for c in classes],
                        "current_stream_in": [i for i, m in enumerate(with_main) if m],
                        "rccl_stream_in": [i for i, r in enumerate(with_rccl) if r], "free_classes": len(free),
                        "picked_from_free": len(chosen), "plain_fill": max(0, n - len(chosen))})
    while len(chosen) < n:     # not enough free queues (other libraries' streams): plain streams, as on one GPU
        chosen.append(torch.cuda.Stream(device=dev))
    _picked[key] = chosen
    return chosen


def wait_all(works):
    for w in works or ():
        if w is not None:
            w.wait()


def broadcast_(t, src=0):
    """In-place broadcast of a tensor from rank `src` (replicas start / resume identical)."""
    if is_distributed():
        dist.broadcast(t, src=src)
    return t


def gather_counts(t):
    """All-gather of a small per-rank tensor along dim 0 (the inference sweep's filtered-grasp counts, SURVEY.md
    8(e)): every rank gets the concatenation in rank order.  Shards may differ in length.  The exchange runs where
    the backend can (RCCL: device tensors; gloo: host tensors); the result comes back on t's device."""
    if not is_distributed():
        return t
    home = t.device
    where = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = t.to(where)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=where)
    sizes = [torch.zeros_like(n) for _ in range(world_size())]
    dist.all_gather(sizes, n)
    m = int(max(int(x.item()) for x in sizes))
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=where)
    pad[:t.shape[0]] = t
    parts = [torch.zeros_like(pad) for _ in range(world_size())]
    dist.all_gather(parts, pad)
    return torch.cat([p[:int(k.item())] for p, k in zip(parts, sizes)], dim=0).to(home)


def grad_scale():
    """Factor the fused Adam applies to the summed gradients: mean over the global batch."""
    return 1.0 / float(world_size())
