import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dex_gan_grasp_amd import _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4120
M, N, S = 1024, 768, 4
A = torch.randn((M, K), device="cuda"); W = torch.randn((N, K), device="cuda")
slabs = torch.empty((S * M * N,), device="cuda")
lib = _lib.lib()
fn = lambda: _lib.check(lib.dgg_gemm_f32(0, M, N, K, _lib.ptr(A), K, _lib.ptr(W), K, None, N, S, _lib.ptr(slabs), 0, _lib.stream_ptr()))
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 32)()
for _ in range(3): fn()
torch.cuda.synchronize(); raw.dgg_x3_read_timing(buf, 1)
for _ in range(20): fn()
torch.cuda.synchronize(); raw.dgg_x3_read_timing(buf, 0)
v = list(buf)
p = max(v[3], 1); q = max(v[11], 1); c = max(v[18], 1)
print("split-first producer per k-tile: park %.0f  fetch %.0f  barrier %.0f" % (v[0]/p, v[1]/p, v[2]/p))
print("load-first  producer per k-tile: fetch %.0f  park %.0f  barrier %.0f" % (v[8]/q, v[9]/q, v[10]/q))
print("consumer per k-tile: reads+mfma %.0f  barrier %.0f" % (v[16]/c, v[17]/c))
