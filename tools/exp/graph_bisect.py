"""Which part of the step survives HIP-graph capture?  usage: graph_bisect.py CASE"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dex_gan_grasp_amd import _lib, engine, synth
from dex_gan_grasp_amd._lib import lib, ptr, check, stream_ptr
case = sys.argv[1]
dev = torch.device("cuda:0")
sys.path.insert(0, "tests")
import test_gpu_fullsize as T
e, _, _ = T._engine("wide")
b = 128
gb = T._dev_inputs(synth.make_batch(b, 1))
eb = T._dev_inputs(synth.make_batch(b, 2, with_label=True), with_label=True)
z = synth.make_z(b, 3).to(dev)
e.joint_step(gb, z, eb, True, None, 1)
e.cgan_step(*gb, z)
e.eva_step(*eb, seed=3)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if case == "adam":
    e.G.adam_step_graph(1e-4); torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        e.G.adam_step_graph(1e-4)
elif case == "cgan1":          # single stream, wgrad streams off
    os.environ["DGG_WGRAD_STREAMS"] = "0"
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        hp = e._hp(None, None); st = e._state(); ws = e._cgan_ws(b)
        args = (ptr(gb[0]), ptr(gb[1]), ptr(gb[2]), ptr(gb[3]), ptr(z))
        check(lib().dgg_cgan_d_phase2(C.byref(e.ccfg), C.byref(st), C.byref(hp), ptr(ws), ws.numel(), b, *args, ptr(e.losses), stream_ptr(dev), None), "d")
        check(lib().dgg_cgan_g_phase(C.byref(e.ccfg), C.byref(st), C.byref(hp), ptr(ws), ws.numel(), b, *args, ptr(e.losses), stream_ptr(dev)), "g")
elif case == "cgan_wg":        # single compute stream + its wgrad side stream
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        e._map_wgrad_streams(torch.cuda.current_stream(dev))
        hp = e._hp(None, None); st = e._state(); ws = e._cgan_ws(b)
        args = (ptr(gb[0]), ptr(gb[1]), ptr(gb[2]), ptr(gb[3]), ptr(z))
        check(lib().dgg_cgan_d_phase2(C.byref(e.ccfg), C.byref(st), C.byref(hp), ptr(ws), ws.numel(), b, *args, ptr(e.losses), stream_ptr(dev), None), "d")
elif case == "cgan2":          # fork of the real pass, no wgrad streams
    os.environ["DGG_WGRAD_STREAMS"] = "0"
    e.close()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        hp = e._hp(None, None); st = e._state(); ws = e._cgan_ws(b)
        args = (ptr(gb[0]), ptr(gb[1]), ptr(gb[2]), ptr(gb[3]), ptr(z))
        check(lib().dgg_cgan_d_phase2(C.byref(e.ccfg), C.byref(st), C.byref(hp), ptr(ws), ws.numel(), b, *args, ptr(e.losses), stream_ptr(dev), C.c_void_p(e._side2.cuda_stream)), "d")
elif case == "eva":
    os.environ["DGG_WGRAD_STREAMS"] = "0"
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        hp = e._hp(None, None, None); ws = e._eva_ws(b); E = e.E
        check(lib().dgg_eva_phase_graph(C.byref(e.ccfg), ptr(E.params), ptr(E.grads), ptr(E.bn), C.byref(hp), ptr(ws), ws.numel(), b,
              ptr(eb[0]), ptr(eb[1]), ptr(eb[2]), ptr(eb[3]), ptr(seed), ptr(e.losses), stream_ptr(dev)), "e")
elif case == "joint_nowg":
    os.environ["DGG_WGRAD_STREAMS"] = "0"
    e.close()
    gs = e.capture_joint_step(gb, z, eb, True, seed=5)
    g = gs.graph
elif case == "cgan2_wg":
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        e._map_wgrad_streams(torch.cuda.current_stream(dev), e._side2)
        hp = e._hp(None, None); st = e._state(); ws = e._cgan_ws(b)
        args = (ptr(gb[0]), ptr(gb[1]), ptr(gb[2]), ptr(gb[3]), ptr(z))
        check(lib().dgg_cgan_d_phase2(C.byref(e.ccfg), C.byref(st), C.byref(hp), ptr(ws), ws.numel(), b, *args, ptr(e.losses), stream_ptr(dev), C.c_void_p(e._side2.cuda_stream)), "d")
elif case == "cgan_wg_full":
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        e._map_wgrad_streams(torch.cuda.current_stream(dev))
        hp = e._hp(None, None); st = e._state(); ws = e._cgan_ws(b)
        args = (ptr(gb[0]), ptr(gb[1]), ptr(gb[2]), ptr(gb[3]), ptr(z))
        check(lib().dgg_cgan_d_phase2(C.byref(e.ccfg), C.byref(st), C.byref(hp), ptr(ws), ws.numel(), b, *args, ptr(e.losses), stream_ptr(dev), None), "d")
        check(lib().dgg_cgan_g_phase(C.byref(e.ccfg), C.byref(st), C.byref(hp), ptr(ws), ws.numel(), b, *args, ptr(e.losses), stream_ptr(dev)), "g")
elif case == "side_eva":
    os.environ["DGG_WGRAD_STREAMS"] = "0"
    e.close()
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        main = torch.cuda.current_stream(dev)
        e._side.wait_stream(main)
        hp = e._hp(None, None, None); ws = e._eva_ws(b); E = e.E
        with torch.cuda.stream(e._side):
            check(lib().dgg_eva_phase_graph(C.byref(e.ccfg), ptr(E.params), ptr(E.grads), ptr(E.bn), C.byref(hp), ptr(ws), ws.numel(), b,
              ptr(eb[0]), ptr(eb[1]), ptr(eb[2]), ptr(eb[3]), ptr(seed), ptr(e.losses), stream_ptr(dev)), "e")
        e.G.adam_step_graph(1e-4)
        main.wait_stream(e._side)
elif case == "joint":
    gs = e.capture_joint_step(gb, z, eb, True, seed=5)
    g = gs.graph
print(case, "captured", flush=True)
g.replay(); g.replay()
torch.cuda.synchronize()
print(case, "replayed OK", flush=True)
