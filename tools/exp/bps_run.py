import sys; sys.path.insert(0, ".")
import torch, bench
r = bench.bps_roofline(torch.device("cuda",0)); print(r["ms_per_launch"], r["frac"])
r = bench.bps_roofline(torch.device("cuda",0)); print(r["ms_per_launch"], r["frac"])
