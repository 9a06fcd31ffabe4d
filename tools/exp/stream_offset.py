#!/usr/bin/env python3
"""bench.py --no-sub in a process that has already created OFFSET torch streams: where the engine's side streams land among
the 4 hardware queues depends on that count unless they are picked by probing (DGG_PICK_STREAMS=0: plain streams)."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
keep = []
for _ in range(int(os.environ.get("OFFSET", "0"))):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        keep.append(torch.zeros(8, device="cuda"))
    keep.append(st)
sys.argv = ["bench.py", "--no-sub", "--no-cpu-baseline"]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
