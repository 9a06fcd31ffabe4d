"""Synthetic front end of the train step (BASELINE config 4): object point clouds -> BPS encode
on the device -> batch dicts in the DataLoader schema of the reference's datasets.

The reference's DexGeneratorDataSet.__getitem__ (data/dexgenerator_data_set.py:124-186) loads a PRE-COMPUTED BPS
vector and one grasp of that object from an h5 file that is not shipped; its robot scripts encode live clouds
(dexgangrasp_real_robot.py:133-141: centre the cloud, `bps.encode(...)['dists']`).  Here both meet: every batch is B
seeded synthetic clouds (points on a random ellipsoid, centred) resident on the device, encoded by dgg_bps_encode on
the caller's stream, plus seeded synthetic grasps, delivered with the keys / shapes / dtypes of data/eval_batch.pth:
    rot_matrix (B,3,3) f64 · transl (B,3) f64 · joint_conf (B,nj) f64 · bps_object (B,4096) f32 · obj_name list[B]
    [+ label (B,) int64 for the evaluator dataset, dexevaluator_data_set.py:15-175]
`device_tensors=True` keeps the grasp tensors on the device as float32 (what a pinned-memory prefetching loader would
hand to update_*); bps_object always stays on the device — it is produced there.
`prefetch=True`: iteration keeps ONE batch ahead — batch i+1's clouds are encoded on a side stream, enqueued before the
consumer enqueues step i, so the encode runs beside the step instead of in front of it (same tensors, same bits: the
encode is deterministic and touches nothing the step reads).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops, parallel, synth


class SyntheticGraspBatches:
    """Iterable of `n_batches` batches.  Batch i of rank r is a pure function of (seed, i, r): epochs repeat, ranks see
    disjoint data (rank offset in the seed), a prefetching consumer may call `batch(i + 1)` early."""

    def __init__(self, cfg, batch_size, n_batches, basis, n_points=2048, device=None, seed=0, with_label=False,
                 device_tensors=False, prefetch=False):
        self.cfg = cfg
        self.batch_size, self.n_batches, self.n_points = int(batch_size), int(n_batches), int(n_points)
        if device is None:
            device = torch.device('cuda:{}'.format(parallel.local_device_index()))
        self.device = torch.device(device)
        b = torch.as_tensor(np.asarray(basis) if not torch.is_tensor(basis) else basis)
        self.basis = b.to(device=self.device, dtype=torch.float32).contiguous()
        assert self.basis.dim() == 2 and self.basis.shape[1] == 3
        self.seed, self.with_label, self.device_tensors = int(seed), bool(with_label), bool(device_tensors)
        self.last_encode_event = None
        self.prefetch = bool(prefetch)
        self._side = None

    def __len__(self):
        return self.n_batches

    def _seed(self, i):
        return self.seed + 7919 * int(i) + 104729 * parallel.rank()

    def clouds(self, i):
        """(B, N, 3) f32 device tensor: batch i's clouds (centred: the encode's precondition, bps_encoder.py:50)."""
        return synth.make_clouds(self.batch_size, self.n_points, self._seed(i)).to(self.device)

    def encode(self, clouds):
        """dgg_bps_encode on the current stream: (B, N, 3) -> (B, P) distances (what `bps.encode(...)['dists']` is)."""
        return ops.bps_encode(clouds, self.basis, return_ids=False)

    def batch(self, i):
        if not 0 <= i < self.n_batches:
            raise IndexError(i)
        grasp = synth.make_batch(self.batch_size, self._seed(i) + 1, with_label=self.with_label, cfg=self.cfg)
        bps = self.encode(self.clouds(i))
        out = {"rot_matrix": grasp["rot_matrix"], "transl": grasp["transl"], "joint_conf": grasp["joint_conf"]}
        if self.with_label:
            out["label"] = grasp["label"]
        if self.device_tensors:
            out = {k: (v.to(self.device, torch.float32) if v.dtype.is_floating_point else v.to(self.device))
                   for k, v in out.items()}
        out["bps_object"] = bps
        out["obj_name"] = ["synthetic_%d_%d" % (self._seed(i), j) for j in range(self.batch_size)]
        return out

    def batch_on_side_stream(self, i):
        """Batch i produced on the loader's side stream -> (batch, event).  The consumer's stream has to wait for the
        event before it touches the batch (`__iter__` does); the tensors are marked as used by the consumer's stream so
        that the caching allocator does not hand their memory out early."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            b = self.batch(i)
            ev = torch.cuda.Event()
            ev.record(self._side)
        for v in b.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        self.last_encode_event = ev
        return b, ev

    def __iter__(self):
        if not self.prefetch:
            for i in range(self.n_batches):
                yield self.batch(i)
            return
        nxt = self.batch_on_side_stream(0) if self.n_batches > 0 else None
        for i in range(self.n_batches):
            b, ev = nxt
            torch.cuda.current_stream(self.device).wait_event(ev)
            # enqueued NOW, i.e. before the consumer enqueues step i: the two overlap on the device
            nxt = self.batch_on_side_stream(i + 1) if i + 1 < self.n_batches else None
            yield b
