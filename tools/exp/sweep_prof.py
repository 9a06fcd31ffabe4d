"""Config-5 sweep (1250 objects x 200 grasps) alone, for the profiler.  Run it as the profiler's own child — never through env, a shell or a
shebang (rocprofv3 has initialised the GPU by then; an exec behind it takes the box down):
    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <outdir> -o t -- python3 <repo>/tools/exp/sweep_prof.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dex_gan_grasp_amd import ops, synth
from dex_gan_grasp_amd.models.dexgangrasp import DexGanGrasp
cfg = dict(synth.DEFAULT_CFG); cfg["is_train"] = False
m = DexGanGrasp(cfg)
m.DexGANGrasp.load_state_dict(synth.make_state_dict("dexgangrasp", 10, 1.0, True))
m.DexEvaluator.load_state_dict(synth.make_state_dict("evaluator", 15, 1.0, True))
dev = m.device
basis = (torch.rand((4096, 3), generator=torch.Generator().manual_seed(0)) * 0.2 - 0.1).to(dev)
clouds = synth.make_clouds(1250, 2048, 7).to(dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d, _ = ops.bps_encode(clouds, basis)
    o = m.sweep_objects(d, 200, thresholds=(0.5, 0.75, 0.90))
    k = int(o["count"].sum())
    torch.cuda.synchronize(); print("pass %d: %.2f ms kept %d" % (i, (time.perf_counter() - t0) * 1e3, k), flush=True)
