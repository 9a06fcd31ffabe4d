"""Training engine: owns the flat buffers of G, D, E and drives the fused phases of libdgg.so
(dgg_cgan_d_phase -> [all-reduce] -> dgg_adam_l2 -> dgg_cgan_g_phase -> [all-reduce] -> dgg_adam_l2,
dgg_eva_phase -> [all-reduce] -> dgg_adam_l2).  Mirrors the ordering of
DexGanGrasp.update_dexgangrasp / update_dexevaluator (dexgangrasp.py:739-823); see SURVEY.md 3.1."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, ops, parallel
from ._lib import check, lib, ptr, stream_ptr, f32c


SEED_STRIDE = 0x9E3779B97F4A7C15   # what dgg_eva_phase_graph adds to the device-resident dropout seed per step
LOSS_RING = 1024                   # steps a returned loss vector stays valid (TrainEngine.losses)


class _side_work:
    """Context: the library calls inside enqueue side work (dgg_set_side_work), switched off again on exit."""

    def __init__(self, on):
        self.on = int(on)   # 0: off, 1: the default division of the grids, 2: one more, ...

    def __enter__(self):
        if self.on:
            check(lib().dgg_set_side_work(int(self.on)), "dgg_set_side_work")

    def __exit__(self, *exc):
        if self.on:
            lib().dgg_set_side_work(0)
        return False


class GraphStep:
    """One gen+eval train step captured into a HIP graph (TrainEngine.capture_joint_step).  The inputs are the very
    tensors given at capture: refresh their CONTENTS (copy_) between replays, keep them alive.  replay() enqueues the
    whole step — ~200 kernel launches on 3 streams — as one graph launch on the current stream.
    Baked into the captured kernels as arguments: the three learning rates, Adam's betas / eps / weight decay and the
    loss weights — a scheduler step or a changed hyper-parameter needs a new capture.  Not baked: Adam's step numbers and
    the dropout seed (device memory; FlatNet.step's setter refreshes the former when optimiser state is loaded)."""

    def __init__(self, eng, graph, keep, is_train_gen):
        self.eng, self.graph, self.keep, self.is_train_gen = eng, graph, keep, is_train_gen
        self.loss_view = eng.losses   # the ring slot whose address the captured kernels write (TrainEngine.losses)

    def replay(self):
        self.graph.replay()
        e = self.eng
        # the device-resident counters advanced themselves inside the graph: only the host copies follow
        e.D._step += 1
        e.E._step += 1
        if self.is_train_gen:
            e.G._step += 1
        return self.loss_view


class TrainEngine:
    def __init__(self, cfg, flat_g, flat_d, flat_e, device):
        self.cfg = cfg
        self.G, self.D, self.E = flat_g, flat_d, flat_e
        self.device = torch.device(device)
        self.ccfg = _lib.make_cfg(cfg)
        # The loss vector of a step (16 floats, slots _lib.L_*) is written by the kernels of that step into ONE slot of a
        # ring; every step takes the next slot and returns a VIEW of it — no copy kernel on the step's stream (until round 5:
        # `losses.clone()` per call, a launch + its gap on the chain; the reference only reads the values every print_freq
        # steps, train.py:254-258).  A returned vector stays valid for LOSS_RING - 1 further steps of this engine.
        self._loss_ring = torch.zeros((LOSS_RING, 16), dtype=torch.float32, device=self.device)
        self._loss_slot = 0
        self._ws = {}
        self._ws_e = {}
        self._side = None        # evaluator stream
        self._side2 = None       # D real-batch pass (fork inside the D phase, prefetch across steps)
        self._side2_busy = None  # event: last work enqueued on _side2 that the main stream has not joined yet
        self._reducers = None    # data parallel: per-network bucketed all-reduce (parallel.BucketReducer)

    # ---- helpers
    @property
    def losses(self):
        """The loss vector of the step enqueued last (a view of its ring slot)."""
        return self._loss_ring[self._loss_slot]

    def _next_losses(self, kinds=("gan", "eva")):
        """The ring slot this update writes.  A slot holds at most one cGAN update and one evaluator update — one gen+eval
        step: an update whose kind the current slot does not hold yet joins it (cgan_step then eva_step, or the other way
        round, leave ONE vector with both sets of entries, as the single buffer of rounds 1-5 did); otherwise the ring
        advances.  joint_step writes both kinds and always advances."""
        have = getattr(self, "_slot_kinds", None)
        if have is None or have & set(kinds):
            self._loss_slot = (self._loss_slot + 1) % LOSS_RING
            have = set()
        self._slot_kinds = have | set(kinds)
        return self._loss_ring[self._loss_slot]

    def _hp(self, lr_gen=None, lr_dis=None, lr_eva=None):
        hp = ops.make_hparams(self.cfg, parallel.grad_scale())
        if lr_gen is not None:
            hp.lr_gen = lr_gen
        if lr_dis is not None:
            hp.lr_dis = lr_dis
        if lr_eva is not None:
            hp.lr_eva = lr_eva
        return hp

    def _cgan_ws(self, B):
        ws = self._ws.get(B)
        if ws is None:
            n = int(lib().dgg_cgan_workspace_bytes(C.byref(self.ccfg), B))
            if n < 0:
                raise _lib.DggError("dgg_cgan_workspace_bytes: %d" % n)
            ws = _lib.new_workspace(n, self.device)
            self._ws[B] = ws
        return ws

    def _eva_ws(self, B):
        ws = self._ws_e.get(B)
        if ws is None:
            n = int(lib().dgg_workspace_bytes(_lib.NET_EVA, C.byref(self.ccfg), B)) + 8 * B + 4096
            ws = _lib.new_workspace(n, self.device)
            self._ws_e[B] = ws
        return ws

    def _state(self):
        G, D = self.G, self.D
        vals = [G.params, G.grads, G.m, G.v, G.bn, D.params, D.grads, D.m, D.v, D.bn]
        return _lib.GanState(*[t.data_ptr() for t in vals])

    def _reduce(self, which, async_op=False, feeder=None):
        """Sum the gradients of network `which` ("G" / "D" / "E") over the ranks after a phase call whose backward has
        just been enqueued: per-ResBlock buckets behind the library's gradient-ready events (parallel.BucketReducer;
        DGG_BUCKETED_AR=0: one all-reduce of the whole buffer).  Returns work handles when async_op."""
        if not parallel.is_distributed():
            return None
        net = {"G": self.G, "D": self.D, "E": self.E}[which]
        if os.environ.get("DGG_BUCKETED_AR", "1") == "0" or not self._reducers:
            w = parallel.allreduce_sum_(net.grads, async_op=async_op)
            return [w] if async_op else None
        works = self._reducers[which].reduce_async(feeder, timing=getattr(self, "trace_buckets", None), tag=which)
        if async_op:
            return works
        parallel.wait_all(works)
        return None

    def _arm_reducers(self, force=False):
        """Register the gradient-ready events once (calling thread): for the bucketed all-reduce of a data-parallel run,
        and (force) for joint_step's split optimiser step.  D's gradients are final only after the pass that accumulates
        (fake batch) — a prefetched real-batch pass must not signal."""
        if not force and (not parallel.is_distributed() or os.environ.get("DGG_BUCKETED_AR", "1") == "0"):
            return
        if getattr(self, "_reducers", None) is None:
            self._reducers = {"G": parallel.BucketReducer(self.G), "D": parallel.BucketReducer(self.D, accum_only=True),
                              "E": parallel.BucketReducer(self.E, group=parallel.evaluator_group())}
        for r in self._reducers.values():
            if not r.armed:
                r.arm()

    def _adam_split(self, which, lr, hp, works, early):
        """The optimiser step of network `which` in two launches: the bucket the backward completes first — last ResBlock
        + heads, about half of the parameters — on the idle stream `early` as soon as its gradients (and, data
        parallel, its all-reduce) are there, i.e. BESIDE the backward of the remaining blocks; the rest on the current
        stream.  Adam is element-wise: same results as one launch over the whole buffer."""
        net = {"G": self.G, "D": self.D, "E": self.E}[which]
        rd = self._reducers[which]
        works = list(works) if works else [None] * len(rd.bounds)
        if len(works) != len(rd.bounds):      # one all-reduce of the whole buffer: no early part
            parallel.wait_all(works)
            net.adam_step(lr, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
            return
        kw = (lr, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
        net.adam_begin()
        (lo, hi) = rd.bounds[0]
        early.wait_event(rd.events[0])
        with torch.cuda.stream(early):
            parallel.wait_all(works[:1])
            net.adam_range(lo, hi, *kw)
            done = early.record_event()
        parallel.wait_all(works[1:])
        net.adam_range(0, lo, *kw)
        torch.cuda.current_stream(self.device).wait_event(done)

    def flush(self):
        """Make the current stream wait for everything this engine has enqueued on its side streams (a prefetched
        real-batch pass of D writes D.grads, D's running statistics and its workspace).  Called by every entry point
        other than joint_step and by anything that reads the networks' state (checkpoints, eval forwards)."""
        if self._side2_busy is not None:
            torch.cuda.current_stream(self.device).wait_event(self._side2_busy)
            self._side2_busy = None
            # a prefetched pass that nobody has picked up yet: dropped without a trace (D's running statistics go back
            # to what this step left; the next D phase runs its own real pass), so that what the caller reads or does
            # next is the unpipelined order's state
            for key, ws in self._ws.items():
                if not isinstance(key, tuple):
                    check(lib().dgg_cgan_drop_prefetch(ptr(ws), stream_ptr(self.device)), "dgg_cgan_drop_prefetch")

    def _streams(self, two_streams, wgrad=True):
        """(main, s2 as void* or None): the D phase forks its real-batch pass onto _side2.  wgrad: every compute stream may
        also get a library-owned weight-gradient side stream — only with DGG_WGRAD_STREAMS=1: the HIP runtime maps streams
        onto 4 hardware queues in creation order and never rebalances; with more streams than queues, pairs serialise
        (joint_step with six streams: 2.00 instead of 1.92 ms; cgan_step / eva_step in a process that also ran joint_step:
        1.51 / 0.74 instead of 1.49 / 0.74 ms, config 2 1.09 instead of 1.05 ms).  Three streams is what the engine uses."""
        main = torch.cuda.current_stream(self.device)
        if not two_streams:
            return main, None
        if self._side2 is None:
            if self.device.type == "cuda" and os.environ.get("DGG_PICK_STREAMS", "1") != "0":
                # side streams on hardware queues of their own: shared with neither this stream nor (data parallel)
                # RCCL's — where a plain torch.cuda.Stream() lands depends on how many streams the process made before
                # (parallel.pick_report says what was found; a failing probe is fatal under RCCL, see there)
                self._side2, self._side = parallel.side_streams_or_plain(self.device, 2)
            else:
                parallel.pick_report.update({"mode": "plain", "why": "DGG_PICK_STREAMS=0"})
                self._side2 = torch.cuda.Stream(device=self.device)
        if wgrad:
            self._map_wgrad_streams(main, self._side2)
        else:
            self._unmap_wgrad_streams(main, self._side, self._side2)
        return main, C.c_void_p(self._side2.cuda_stream)

    # ---- A10
    def cgan_step(self, bps, rot9, transl, joint, z, is_train_gen=True, lr_gen=None, lr_dis=None, two_streams=False):
        """One update_dexgangrasp step on this rank's shard.  Inputs: fp32 device tensors.
        Returns a (16,) device tensor of losses (slots: _lib.L_*); values are a fresh copy.
        two_streams: intra-step concurrency (D's real-batch pass and the weight-gradient GEMMs on side streams,
        joined before the call returns; bit-identical results)."""
        dev = self.device
        self.flush()
        bps, rot9, transl, joint, z = (f32c(t, dev) for t in (bps, rot9.reshape(-1, 9), transl, joint, z))
        B = bps.shape[0]
        ws = self._cgan_ws(B)
        hp = self._hp(lr_gen, lr_dis)
        st = self._state()
        _, s2 = self._streams(two_streams)
        cfg_p, args = C.byref(self.ccfg), (ptr(bps), ptr(rot9), ptr(transl), ptr(joint), ptr(z))
        self._arm_reducers()
        L = self._next_losses(("gan",))
        self._mark("step_begin")
        check(lib().dgg_cgan_d_phase2(cfg_p, C.byref(st), C.byref(hp), ptr(ws), ws.numel(), B, *args, ptr(L),
                                      stream_ptr(dev), s2), "dgg_cgan_d_phase2")
        self._mark("d_phase_end")
        if getattr(self, "after_d_phase", None) is not None:   # tests: the fake pass' LeakyReLU sides (kink_sides)
            self.after_d_phase(self, B)
        self._reduce("D")
        self.D.adam_step(hp.lr_dis, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
        self._mark("adam_d_end")
        if is_train_gen:
            check(lib().dgg_cgan_g_phase(cfg_p, C.byref(st), C.byref(hp), ptr(ws), ws.numel(), B, *args,
                                         ptr(L), stream_ptr(dev)), "dgg_cgan_g_phase")
            self._mark("g_phase_end")
            self._reduce("G")
            self.G.adam_step(hp.lr_gen, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
            self._mark("adam_g_end")
        return L

    # ---- N3: WGAN-GP variant of the cGAN step (cfg["is_wgan"], dexgangrasp.py:149-211, 234-270)
    def _wgan_ws(self, B):
        key = ("wgan", B)
        ws = self._ws.get(key)
        if ws is None:
            n = int(lib().dgg_wgan_workspace_bytes(C.byref(self.ccfg), B))
            if n < 0:
                raise _lib.DggError("dgg_wgan_workspace_bytes: %d" % n)
            ws = _lib.new_workspace(n, self.device)
            self._ws[key] = ws
        return ws

    def wgan_step(self, bps, rot9, transl, joint, z, alphas, is_train_gen=True, lr_gen=None, lr_dis=None,
                  penalty_gain=10.0):
        """One update_dexgangrasp step with the Wasserstein losses and the gradient penalty.  alphas (B,3): the
        interpolation weights [rot | transl | joint_conf] of calculate_interp (:149-158).  Returns the loss vector
        (slot L_GRAD_PENALTY holds the penalty)."""
        dev = self.device
        self.flush()
        bps, rot9, transl, joint, z = (f32c(t, dev) for t in (bps, rot9.reshape(-1, 9), transl, joint, z))
        alphas = f32c(alphas, dev).reshape(-1, 3)
        B = bps.shape[0]
        ws = self._wgan_ws(B)
        hp = self._hp(lr_gen, lr_dis)
        st = self._state()
        cfg_p = C.byref(self.ccfg)
        args = (ptr(bps), ptr(rot9), ptr(transl), ptr(joint), ptr(z))
        L = self._next_losses(("gan",))
        check(lib().dgg_wgan_d_phase(cfg_p, C.byref(st), C.byref(hp), ptr(ws), ws.numel(), B, *args, ptr(alphas),
                                     float(penalty_gain), ptr(L), stream_ptr(dev)), "dgg_wgan_d_phase")
        parallel.allreduce_sum_(self.D.grads)
        self.D.adam_step(hp.lr_dis, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
        if is_train_gen:
            check(lib().dgg_wgan_g_phase(cfg_p, C.byref(st), C.byref(hp), ptr(ws), ws.numel(), B, *args,
                                         ptr(L), stream_ptr(dev)), "dgg_wgan_g_phase")
            parallel.allreduce_sum_(self.G.grads)
            self.G.adam_step(hp.lr_gen, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
        return L

    # ---- A11
    def eva_step(self, bps, rot9, transl, label, drop_keep=None, seed=0, lr=None, two_streams=False):
        dev = self.device
        self.flush()
        bps, rot9, transl, label = (f32c(t, dev) for t in (bps, rot9.reshape(-1, 9), transl, label.reshape(-1)))
        dk = f32c(drop_keep, dev) if drop_keep is not None else None
        B = bps.shape[0]
        ws = self._eva_ws(B)
        hp = self._hp(lr_eva=lr)
        E = self.E
        if two_streams:
            self._map_wgrad_streams(torch.cuda.current_stream(dev))
        self._arm_reducers()
        L = self._next_losses(("eva",))
        check(lib().dgg_eva_phase(C.byref(self.ccfg), ptr(E.params), ptr(E.grads), ptr(E.bn), C.byref(hp), ptr(ws),
                                  ws.numel(), B, ptr(bps), ptr(rot9), ptr(transl), ptr(label), ptr(dk), int(seed),
                                  ptr(L), stream_ptr(dev)), "dgg_eva_phase")
        self._reduce("E")
        E.adam_step(hp.lr_eva, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
        return L

    def eva_keep_masks(self, B):
        """The dropout keep masks (3, B, 512) uint8 that the last train-mode evaluator pass at batch B wrote into
        its workspace (drawn in-kernel from Philox unless masks were given) — what its backward used."""
        ws = self._eva_ws(B)
        out = []
        for r in range(3):
            off = int(lib().dgg_eva_keep_offset(C.byref(self.ccfg), B, r))
            if off < 0:
                raise _lib.DggError("dgg_eva_keep_offset: %d" % off)
            out.append(ws[off:off + B * 512].view(B, 512).clone())
        return torch.stack(out)

    def kink_sides(self, which, B, ws=None):
        """Which side of every LeakyReLU kink (networks.py:41) the last train-mode pass of a network took, read back
        from its workspace: a list of bool tensors [fc3 branch (B,512), bn1 branch (B,256), block output (B,512)] per
        ResBlock, True = positive side — the order a CPU evaluation of the same pass meets them.  which: "G", "Dreal",
        "Dfake" (cGAN step workspace of batch B; "Dfake" holds the D phase's fake pass until the G phase's pass through
        the updated discriminator overwrites it) or "E".  ws: a snapshot (clone) of the workspace taken earlier.
        Dropped units of the evaluator read -1 in its (int8) block-output masks: no side to follow.  Parity tests hand these to the
        oracle (oracle.dexgan_oracle.forced_slopes) so that gradient comparisons are strict on every data set."""
        net = {"G": _lib.NET_GEN, "Dreal": _lib.NET_DISC, "Dfake": _lib.NET_DISC, "E": _lib.NET_EVA}[which]
        if which == "E":
            return _lib.kink_sides(net, self.ccfg, self._eva_ws(B) if ws is None else ws, B)
        base = int(lib().dgg_cgan_ws_offset(C.byref(self.ccfg), B, {"G": 0, "Dreal": 1, "Dfake": 2}[which]))
        return _lib.kink_sides(net, self.ccfg, self._cgan_ws(B) if ws is None else ws, B, base)

    def close(self):
        """Release what the library remembers about this engine: pending-prefetch slots keyed by its workspaces and
        its weight-gradient side streams (one disable per enable).  Must run on the thread that used the engine."""
        try:
            for r in (getattr(self, "_reducers", None) or {}).values():
                r.disarm()
            for key, ws in self._ws.items():
                if not isinstance(key, tuple):
                    lib().dgg_cgan_release(ptr(ws))
            for key in getattr(self, "_wg_mapped", None) or ():
                lib().dgg_enable_wgrad_stream(C.c_void_p(key), 0)
            self._wg_mapped = set()
        except Exception:
            pass

    def __del__(self):
        self.close()

    def _map_wgrad_streams(self, *streams):
        """Give each compute stream a weight-gradient side stream (dgg_enable_wgrad_stream): dW GEMMs have no
        consumer before Adam, so they overlap the dgrad chain of the following ResBlocks.  The side streams are
        created and owned by the library — streams from torch's pool are shared between users, and one that served
        as a side stream here and as a compute stream (with its own side stream) elsewhere could close a wait
        cycle."""
        mode = os.environ.get("DGG_WGRAD_STREAMS", "0")   # opt-in ("1"): see _streams
        if mode == "0":
            return
        if mode == "main":   # diagnostics: only the stream the step is called on gets one
            streams = [s for s in streams if s is not self._side and s is not self._side2]
        if getattr(self, "_wg_mapped", None) is None:
            self._wg_mapped = set()
        for s in streams:
            key = s.cuda_stream
            if key in self._wg_mapped:
                continue
            check(lib().dgg_enable_wgrad_stream(C.c_void_p(key), 1), "dgg_enable_wgrad_stream")
            self._wg_mapped.add(key)

    def _unmap_wgrad_streams(self, *streams):
        if not getattr(self, "_wg_mapped", None):
            return
        for s in streams:
            if s is not None and s.cuda_stream in self._wg_mapped:
                check(lib().dgg_enable_wgrad_stream(C.c_void_p(s.cuda_stream), 0), "dgg_enable_wgrad_stream")
                self._wg_mapped.discard(s.cuda_stream)

    # ---- gen + eval step with communication/compute overlap (the unit bench.py times)
    def capture_joint_step(self, gan_inputs, z, eva_inputs, is_train_gen=True, seed=0, lr_gen=None, lr_dis=None,
                           lr_eva=None):
        """Capture joint_step into a HIP graph.  What changes from step to step is kept in device memory — the three
        Adam step numbers (dgg_adam_l2_graph) and the dropout seed (dgg_eva_phase_graph; the s-th replay draws masks
        from seed + s * SEED_STRIDE) — so a replay continues training exactly like an eager step (bit-identical,
        tests/test_gpu_fullsize.py).  One eager step in graph form runs first (it is a real training step: streams,
        events, workspaces and the device counters come into being outside the capture).  Single GPU only: collectives
        are not captured.  Returns a GraphStep."""
        if parallel.is_distributed():
            raise _lib.DggError("capture_joint_step: data-parallel steps are not captured (RCCL work is enqueued eagerly)")
        dev = self.device
        keep = (tuple(gan_inputs), z, tuple(eva_inputs))
        for t in keep[0] + (z,) + keep[2]:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.losses.device:
                raise _lib.DggError("capture_joint_step needs resident contiguous float32 tensors (they are used in place)")
        if getattr(self, "_seed_dev", None) is None:
            self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._seed_dev.fill_(int(seed) - (1 << 64) if int(seed) >= (1 << 63) else int(seed))
        for r in (self._reducers or {}).values():   # no gradient-ready events inside a capture (eager steps re-arm)
            r.disarm()
        kw = dict(is_train_gen=is_train_gen, lr_gen=lr_gen, lr_dis=lr_dis, lr_eva=lr_eva, graph_mode=True)
        # (joint_step maps no weight-gradient streams: every stream of the capture is a direct child of the capturing
        # stream — ROCm 7.2 crashes on a captured stream that waits on another forked one, see dgg.h)
        self.joint_step(gan_inputs, z, eva_inputs, **kw)           # priming step, eager
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.joint_step(gan_inputs, z, eva_inputs, **kw)
        return GraphStep(self, g, keep, is_train_gen)

    def _mark(self, name, stream=None):
        """Diagnostics (tools/phase_times.py): with self.trace = [] every joint_step leaves timed events at its phase
        boundaries on the stream that runs the phase."""
        tr = getattr(self, "trace", None)
        if tr is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream(self.device))
            tr.append((name, ev))

    def joint_step(self, gan_inputs, z, eva_inputs, is_train_gen=True, drop_keep=None, seed=0, lr_gen=None,
                   lr_dis=None, lr_eva=None, two_streams=True, next_gan_inputs=None, graph_mode=False):
        """One update_dexgangrasp + one update_dexevaluator.  The two updates touch disjoint networks, batches
        and workspaces, so the evaluator update runs on a SECOND HIP stream concurrently with the cGAN update:
        most launches of either (column kernels, split-K reduces, the K=256 GEMMs) occupy a fraction of the 256
        CUs, and on N > 1 GPUs each gradient all-reduce overlaps with the other stream's compute:
            stream A:  D phase | AR(D) | Adam(D) | G phase | AR(G) | Adam(G)
            stream B:  E phase | AR(E) | Adam(E)
        (RCCL runs the collectives on its own stream; `wait()` is a stream-level dependency, no host sync; all
        ranks enqueue the collectives in the same host order.)  Results are bit-identical to cgan_step + eva_step.

        next_gan_inputs (optional, is_train_gen only): the NEXT call's gan_inputs, already resident on the device as
        contiguous float32 tensors (a prefetching data loader has them).  The real-batch pass of the next D phase is
        then enqueued behind this step's third D forward and overlaps its G phase (dgg_cgan_d_real_prefetch); the next
        call must pass the very same tensors as gan_inputs.  Same results, bit for bit: parameters, Adam state, losses —
        and D's running statistics and gradient buffer once the prefetched pass has been picked up by that next call or
        dropped by flush() (any serial entry point, checkpoint or evaluation goes through flush()); in between they hold
        the next batch's real-pass contribution."""
        dev = self.device
        bps, rot9, transl, joint = (f32c(t, dev) for t in gan_inputs)
        if not two_streams:
            self.flush()
        rot9 = rot9.reshape(-1, 9)
        z = f32c(z, dev)
        ebps, erot9, etransl, elabel = (f32c(t, dev) for t in eva_inputs)
        erot9, elabel = erot9.reshape(-1, 9), elabel.reshape(-1)
        dk = f32c(drop_keep, dev) if drop_keep is not None else None
        B, Be = bps.shape[0], ebps.shape[0]
        ws, wse = self._cgan_ws(B), self._eva_ws(Be)
        hp = self._hp(lr_gen, lr_dis, lr_eva)
        st = self._state()
        cfg_p = C.byref(self.ccfg)
        args = (ptr(bps), ptr(rot9), ptr(transl), ptr(joint), ptr(z))
        G, D, E = self.G, self.D, self.E
        if graph_mode:   # step numbers and dropout seed in device memory; the host counters follow in GraphStep.replay
            adam = lambda net, lr: net.adam_step_graph(lr, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
            next_gan_inputs = None
        else:
            adam = lambda net, lr: net.adam_step(lr, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.grad_scale)
        wg = os.environ.get("DGG_JOINT_WGRAD_STREAMS", "0") == "1" and not graph_mode
        main, s2 = self._streams(two_streams, wgrad=wg)
        if two_streams:
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side
            side.wait_stream(main)          # inputs / previous step are ordered before the evaluator update
            if wg:
                self._map_wgrad_streams(side)
        else:
            side = main
        # the two-launch optimiser step (opt-in since round 6, DGG_SPLIT_ADAM=1): the early bucket beside the rest of the
        # backward.  It paid while the chain was longer (rounds 2-5: -10..-20 us per step); with the chain of round 6 its two
        # cross-stream waits cost more than the overlap gives (1.350 / 1.344 against 1.339 / 1.335 ms, one box, alternating)
        split = (two_streams and not graph_mode and os.environ.get("DGG_SPLIT_ADAM", "0") == "1"
                 and not parallel.is_distributed())
        self._arm_reducers(force=split)
        self._mark("step_begin")
        # this step's loss vector: the next ring slot (a look-ahead real pass enqueued by the previous step has already
        # written its BCE term there); a captured step keeps ONE slot, its address is baked into the graph
        L = self.losses if graph_mode else self._next_losses()
        L_next = self._loss_ring[(self._loss_slot + 1) % LOSS_RING]
        check(lib().dgg_cgan_d_phase2(cfg_p, C.byref(st), C.byref(hp), ptr(ws), ws.numel(), B, *args, ptr(L),
                                      stream_ptr(dev), s2), "dgg_cgan_d_phase2")
        self._mark("d_phase_end")
        if getattr(self, "after_d_phase", None) is not None:   # tests: snapshot the fake pass' workspace (kink_sides)
            self.after_d_phase(self, B)
        self._side2_busy = None             # the D phase joins _side2 (a picked-up prefetch included) into main
        # data parallel: the bucket collectives are fed from a compute stream that is idle at that point, never from
        # streams of their own (parallel.BucketReducer.reduce_async): D's from _side2 (joined above); the evaluator's
        # from _side2 as well, behind D's — RCCL runs a group's collectives in issue order, and fed from the evaluator's
        # own stream its 42.6 MB would start only when its phase ends and sit in front of G's buckets; the prefetched
        # real pass queues behind those waits (it has the whole G phase to run in); G's from the evaluator's stream
        # (done by then)
        w_d = self._reduce("D", async_op=True, feeder=self._side2 if two_streams else None)
        # the evaluator update and the prefetched real pass run BESIDE the cGAN chain: their large GEMMs take half the
        # chip for twice as long (dgg_set_side_work) so that the chain's short kernels keep finding free CUs
        polite = two_streams and os.environ.get("DGG_SIDE_WORK", "1") != "0"
        # (level of the evaluator update: a half of the workgroups beside a B = 1024 step with a look-ahead pass — 1.293 ms
        #  against 1.347 at a third —, a third everywhere else: B = 1024 without look-ahead 1.398 -> 1.371, B = 512 0.981 ->
        #  0.965 / without look-ahead 1.077 -> 1.035, B = 256 0.880 -> 0.811; profiles/r06_experiments.txt 20)
        ahead = is_train_gen and two_streams and next_gan_inputs is not None and next_gan_inputs[0].shape[0] == B
        eva_level = os.environ.get("DGG_SIDE_EVA_LEVEL") or ("1" if (ahead and B > 512) else "2")
        with torch.cuda.stream(side), _side_work(int(polite) * int(eva_level)):
            if graph_mode:
                check(lib().dgg_eva_phase_graph(cfg_p, ptr(E.params), ptr(E.grads), ptr(E.bn), C.byref(hp), ptr(wse),
                                                wse.numel(), Be, ptr(ebps), ptr(erot9), ptr(etransl), ptr(elabel),
                                                ptr(self._seed_dev), ptr(L), stream_ptr(dev)),
                      "dgg_eva_phase_graph")
            else:
                check(lib().dgg_eva_phase(cfg_p, ptr(E.params), ptr(E.grads), ptr(E.bn), C.byref(hp), ptr(wse),
                                          wse.numel(), Be, ptr(ebps), ptr(erot9), ptr(etransl), ptr(elabel), ptr(dk),
                                          int(seed), ptr(L), stream_ptr(dev)), "dgg_eva_phase")
            w_e = self._reduce("E", async_op=True, feeder=self._side2 if two_streams else None)
            self._mark("e_phase_end", side)
            # the evaluator's optimiser step follows on its own stream at once (it blocks nothing else there; enqueued
            # after the G phase it would reach the GPU only when the host gets that far — behind the generator's)
            parallel.wait_all(w_e)
            if w_e:
                self._mark("ar_e_waited", side)     # (trace: e_phase_end -> here = the evaluator's exposed all-reduce time)
            adam(E, hp.lr_eva)
            self._mark("adam_e_end", side)
        # (data parallel: one optimiser launch per network — the stream that would run the early half is the one feeding
        # the collectives, and every extra cross-stream wait costs ~12 us there: 2.05 vs 1.88 ms with a one-rank group)
        if split:    # (_side2 is idle here: the real pass it prefetched for this step ended long ago)
            self._adam_split("D", hp.lr_dis, hp, w_d, self._side2)
        else:
            parallel.wait_all(w_d)
            if w_d:
                self._mark("ar_d_waited")           # (trace: d_phase_end -> here = D's exposed all-reduce time)
            adam(D, hp.lr_dis)
        self._mark("adam_d_end")
        w_g = None
        if is_train_gen:
            check(lib().dgg_cgan_g_phase(cfg_p, C.byref(st), C.byref(hp), ptr(ws), ws.numel(), B, *args,
                                         ptr(L), stream_ptr(dev)), "dgg_cgan_g_phase")
            self._mark("g_phase_end")
            # a ragged next batch (last of an epoch) is simply not prefetched: the workspace is carved per batch size
            if two_streams and next_gan_inputs is not None and next_gan_inputs[0].shape[0] == B:
                nb, nr, nt, nj_ = next_gan_inputs
                nr = nr.reshape(-1, 9)
                for t in (nb, nr, nt, nj_):
                    if t.dtype != torch.float32 or not t.is_contiguous() or t.device != bps.device:
                        raise _lib.DggError("next_gan_inputs must be resident contiguous float32 tensors")
                self._keep_next = (nb, nr, nt, nj_)   # the C side remembers raw pointers
                self._mark("prefetch_begin", self._side2)
                # (the look-ahead pass runs beside the G phase, which has the chain's densest stretch of large launches: its
                #  forward products take a THIRD of the workgroups, three work items each — 1.300 / 1.292 -> 1.287 / 1.283 ms
                #  against the evaluator's half; a quarter gives it back, and the evaluator itself is best at a half)
                with _side_work(int(polite) * int(os.environ.get("DGG_SIDE_PRE_LEVEL", "2"))):
                    check(lib().dgg_cgan_d_real_prefetch(cfg_p, C.byref(st), C.byref(hp), ptr(ws), ws.numel(),
                                                         nb.shape[0], ptr(nb), ptr(nr), ptr(nt), ptr(nj_),
                                                         ptr(L_next), s2), "dgg_cgan_d_real_prefetch")
                self._side2_busy = self._side2.record_event()
                self._mark("prefetch_end", self._side2)
            # (G's collectives are issued AFTER the prefetch is enqueued: see below)
            w_g = self._reduce("G", async_op=True, feeder=side if two_streams else None)
        if is_train_gen:
            if split:    # (the evaluator's stream is idle by now; _side2 runs the prefetched real pass)
                self._adam_split("G", hp.lr_gen, hp, w_g, side)
            else:
                parallel.wait_all(w_g)
                if w_g:
                    self._mark("ar_g_waited")       # (trace: g_phase_end -> here = G's exposed all-reduce time)
                adam(G, hp.lr_gen)
            self._mark("adam_g_end")
        if two_streams:
            main.wait_stream(side)
        if graph_mode:       # eager priming step: the host counters follow the device's (a captured call runs nothing)
            if not torch.cuda.is_current_stream_capturing():
                D._step += 1
                E._step += 1
                if is_train_gen:
                    G._step += 1
            return L
        return L
