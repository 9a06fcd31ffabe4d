import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch, numpy as np
import helpers as H
from dex_gan_grasp_amd import synth, _lib as L, ops, engine
from oracle import dexgan_oracle as O
CFG = dict(synth.DEFAULT_CFG); DEV = 'cuda'
variant, bname = 'init', 'syn64'
batch = H.batches()[bname]; b = 64
z = synth.make_z(b, 100)
def oracle_grads(dtype):
    sd = H.gan_sd(variant)
    if dtype == torch.float64:
        sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
        torch.Tensor.float = lambda self: self
    ag = O.AdamState([n for n in O.param_names(sd) if n.startswith('generator.')])
    ad = O.AdamState([n for n in O.param_names(sd) if n.startswith('discriminator.')])
    bt = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in batch.items()}
    out = O.cgan_step(sd, bt, z.to(dtype), ag, ad, CFG, True, return_grads=True)
    return out[1], out[2]
orig_float = torch.Tensor.float
gd32, gg32 = oracle_grads(torch.float32)
gd64, gg64 = oracle_grads(torch.float64)
torch.Tensor.float = orig_float
sd = H.gan_sd(variant)
fg, fd, fe = (L.FlatNet(k, CFG, DEV) for k in (0, 1, 2))
fg.load_state_dict(sd, 'generator.'); fd.load_state_dict(sd, 'discriminator.')
eng = engine.TrainEngine(CFG, fg, fd, fe, DEV)
hp = ops.make_hparams(CFG)
ws = eng._cgan_ws(b); st = eng._state()
bps = batch['bps_object'].to(DEV); rot9 = batch['rot_matrix'].float().reshape(b, 9).to(DEV)
tr, jt = batch['transl'].float().to(DEV), batch['joint_conf'].float().to(DEV); zd = z.to(DEV)
L.check(L.lib().dgg_cgan_d_phase(C.byref(eng.ccfg), C.byref(st), C.byref(hp), L.ptr(ws), ws.numel(), b, L.ptr(bps), L.ptr(rot9), L.ptr(tr), L.ptr(jt), L.ptr(zd), L.ptr(eng.losses), L.stream_ptr()))
gv = {k: v.cpu() for k, v in fd.views('grads').items()}
for n in ['disc_bn1.weight', 'disc_bn1.bias', 'disc_rb1.fc1.weight', 'disc_rb1.bn1.weight', 'disc_rb2.fc3.weight']:
    r64 = gd64['discriminator.' + n]; r32 = gd32['discriminator.' + n]; g = gv[n]
    mx = float(r64.abs().max())
    e_or = (r32.double() - r64).abs(); e_us = (g.double() - r64).abs()
    print('%-22s max|g| %.3e  oracle32-vs-64 max %.3e (%.2e rel)  ours-vs-64 max %.3e (%.2e rel)  ours-vs-oracle32 %.3e' % (
        n, mx, float(e_or.max()), float(e_or.max()) / mx, float(e_us.max()), float(e_us.max()) / mx, float((g - r32).abs().max())))
    if n == 'disc_bn1.weight':
        k = int(e_us.argmax()); print('   worst col (ours)', k, 'ours', float(g[k]), 'o32', float(r32[k]), 'o64', float(r64[k]))
        k = int(e_or.argmax()); print('   worst col (oracle32)', k, 'ours', float(g[k]), 'o32', float(r32[k]), 'o64', float(r64[k]))
        print('   pose cols err ours', e_us[4096:].max().item(), ' bps cols err ours', e_us[:4096].max().item())
