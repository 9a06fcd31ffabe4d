"""One cloud of N points through dgg_bps_encode_ws, R times (for rocprofv3 --kernel-trace --stats):
    python3 tools/exp/bps_one.py [N ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dex_gan_grasp_amd import _lib, synth  # noqa: E402

dev = torch.device("cuda:0")
basis = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "bps_real_objects.npz"))["basis"]
bs = torch.from_numpy(basis).to(dev)
lib = _lib.lib()
P = basis.shape[0]
for N in [int(a) for a in sys.argv[1:]] or [73000, 15009]:
    cl = synth.make_clouds(1, N, 5).to(dev).contiguous()
    d = torch.empty((1, P), device=dev)
    ids = torch.empty((1, P), device=dev, dtype=torch.int32)
    nws = int(lib.dgg_bps_workspace_bytes(1, N, P))
    ws = torch.empty((max(nws, 8),), dtype=torch.uint8, device=dev)
    fn = lambda: _lib.check(lib.dgg_bps_encode_ws(_lib.ptr(cl), 1, N, _lib.ptr(bs), P, _lib.ptr(d), _lib.ptr(ids), _lib.ptr(ws), nws,
                                                  _lib.stream_ptr()))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("N = %d: %.1f us per encode (ws %d bytes)" % (N, 1e3 * e0.elapsed_time(e1) / 50, nws))
