#!/usr/bin/env python3
"""Config 5 of BASELINE.json in miniature (inference sweep, 1 GPU): per object — BPS-encode a 2048-point cloud,
generate 200 grasps, score them with the evaluator and filter at 0.5 / 0.75 / 0.9 — through the reference's Python
surface (DexGanGrasp.generate_grasps / filter_grasps_multi).  Prints objects/s and grasps/s."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dex_gan_grasp_amd import ops, synth  # noqa: E402
from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=64)
    ap.add_argument("--grasps", type=int, default=200)
    ap.add_argument("--points", type=int, default=2048)
    a = ap.parse_args()
    cfg = dict(synth.DEFAULT_CFG)
    cfg["is_train"] = False
    m = DexGanGrasp(cfg)
    m.DexGANGrasp.load_state_dict(synth.make_state_dict("dexgangrasp", 0))
    m.DexEvaluator.load_state_dict(synth.make_state_dict("evaluator", 5))
    m.DexGANGrasp.eval()
    m.DexEvaluator.eval()
    dev = m.device
    basis = (torch.rand((4096, 3), generator=torch.Generator().manual_seed(0)) * 0.2 - 0.1).to(dev)
    clouds = synth.make_clouds(a.objects, a.points, 7).to(dev)

    def one_pass():
        kept = 0
        dists, _ = ops.bps_encode(clouds, basis)                 # all objects at once: (objects, 4096)
        bps_all = dists.cpu().numpy()
        for o in range(a.objects):
            g = m.generate_grasps(bps_all[o], a.grasps, return_arr=True)
            out = m.filter_grasps_multi(bps_all[o], g, thresholds=(0.5, 0.75, 0.90), return_arr=True)
            kept += sum(0 if f is None else f["transl"].shape[0] for f in out)
        return kept
    def batched_pass():
        dists, _ = ops.bps_encode(clouds, basis)
        out = m.sweep_objects(dists, a.grasps, thresholds=(0.5, 0.75, 0.90))
        return int(out["count"].sum())

    for name, fn in (("per object (reference call sequence)", one_pass), ("sweep_objects (one pass for all objects)", batched_pass)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kept = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"mode": name,
                          "workload": "BPS encode + generate %d grasps + evaluate + 3-threshold filter per object" % a.grasps,
                          "objects": a.objects, "points_per_cloud": a.points, "seconds": round(dt, 4),
                          "objects_per_s": round(a.objects / dt, 1),
                          "grasps_per_s": round(a.objects * a.grasps / dt, 1), "kept_total": int(kept)}))


if __name__ == "__main__":
    main()
