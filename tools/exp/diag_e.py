import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, numpy as np
import test_gpu_fullsize as T
from dex_gan_grasp_amd import synth
import helpers as H
from oracle import dexgan_oracle as O
b = 1024
for seed in (2, 3, 4, 5, 6):
    eng, sd, sde = T._engine("wide")
    ae = O.AdamState(O.param_names(sde))
    eb = synth.make_batch(b, seed, with_label=True)
    e_in = T._dev_inputs(eb, with_label=True)
    eng.eva_step(*e_in, seed=77, two_streams=True)
    torch.cuda.synchronize()
    masks = eng.eva_keep_masks(b).float().cpu()
    refe, g_e, _ = O.eva_step(sde, H.clone_batch(eb), ae, T.CFG, masks, return_grads=True)
    worst = []
    for name, got, ref, train, batch in T._grad_items(eng.E, g_e, "", b):
        if H.is_pre_bn_bias(name): continue
        l2, mx = H.grad_tensor_errors(got, ref)
        worst.append((mx, l2, name))
    worst.sort(reverse=True)
    print("seed", seed, ["%s mx %.1e l2 %.1e" % (n, mx, l2) for mx, l2, n in worst[:4]], flush=True)
    eng.close()
