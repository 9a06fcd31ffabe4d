// Is v_dot2c_f32_bf16 with a (-1, 0) selector an exact "x - float(bf16 half)"?  (it would halve the residual
// arithmetic of the bf16x3 split).  Compares with the shift/and + v_sub form on random and special inputs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* yd, float* ys, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    f32x2 v = {x[2 * i], x[2 * i + 1]};
    bf16x2 hb = __builtin_convertvector(v, bf16x2);
    unsigned h = __builtin_bit_cast(unsigned, hb);
    // the selectors go through registers: as immediates the compiler emits the INLINE constant -1.0 for (-1, 0),
    // which the hardware does not read as bf16 (wrong results); the literal 0xbf800000 for (0, -1) is fine
    unsigned s0 = 0x0000bf80u, s1 = 0xbf800000u;
    asm volatile("" : "+s"(s0), "+s"(s1));
    bf16x2 m0 = __builtin_bit_cast(bf16x2, s0);
    bf16x2 m1 = __builtin_bit_cast(bf16x2, s1);
    yd[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(hb, m0, v.x, false);
    yd[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(hb, m1, v.y, false);
    ys[2 * i] = v.x - __builtin_bit_cast(float, h << 16);
    ys[2 * i + 1] = v.y - __builtin_bit_cast(float, h & 0xffff0000u);
}
int main() {
    const int n = 1 << 22;
    float* hx = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
        int mode = i & 7;
        if (mode == 0) u = (u & 0x807fffffu) | ((unsigned)(1 + rand() % 20) << 23);        // tiny normals
        else if (mode == 1) u = (u & 0x807fffffu);                                       // denormals
        else if (mode == 2) u = (u & 0x807fffffu) | ((unsigned)(230 + rand() % 24) << 23); // huge
        else u = (u & 0x807fffffu) | ((unsigned)(100 + rand() % 50) << 23);
        memcpy(&hx[i], &u, 4);
    }
    hx[8] = 0.f; hx[9] = -0.f; hx[10] = 1.f; hx[11] = INFINITY;
    float *dx, *dd, *ds;
    hipMalloc(&dx, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, dd, ds, n);
    float* hd = (float*)malloc(n * 4); float* hs = (float*)malloc(n * 4);
    hipMemcpy(hd, dd, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost);
    long bad[8] = {0}, tot[8] = {0};
    for (int i = 0; i < n; ++i) {
        int mode = i & 7; ++tot[mode];
        if (memcmp(&hd[i], &hs[i], 4) != 0 && !(hd[i] == 0.f && hs[i] == 0.f) && !(isnan(hd[i]) && isnan(hs[i]))) {
            if (bad[mode]++ < 3) printf("mode %d x=%a dot=%a sub=%a\n", mode, hx[i], hd[i], hs[i]);
        }
    }
    for (int m = 0; m < 8; ++m) printf("mode %d: %ld / %ld differ\n", m, bad[m], tot[m]);
    return 0;
}
