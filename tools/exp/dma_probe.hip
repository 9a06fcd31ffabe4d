// Probe of LDS-DMA (global_load_lds_dwordx4) on gfx950 for the GEMM staging patterns of gemm_x3:
//  (1) functional: lane-linear LDS image from per-lane global addresses, 16-byte aligned and 4-byte aligned sources;
//  (2) rate: L2/MALL-resident fp32 matrix streamed as k-tiles by 4 loader waves per CU, pieces of 8 rows x 128 B
//      (k-contiguous operand) or 2 rows x 512 B (k-strided operand), row pitch 16480 B, one workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) float lds_f32;
__global__ void func_kernel(const float* __restrict__ src, int shift, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    const int lane = threadIdx.x;
    // the compiler does not model the DMA's lane-dependent LDS write: launder the LDS offset so that reads of the
    // staged bytes cannot be folded to "never stored"
    unsigned base = (unsigned)(size_t)(lds_f32*)lds + 1024;  // second KiB: also checks the M0 base
    asm volatile("" : "+s"(base));
    const float* p = src + shift + lane * 4;  // 16 B per lane, source aligned to 4*(shift) bytes
    __builtin_amdgcn_global_load_lds(p, (lds_f32*)base, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = ((lds_f32*)base)[i];
}

// rate probe: each workgroup streams `ntile` k-tiles of (rows x 32 floats) [KC] or (32 x cols) [!KC]
template <int MODE>  // 0: KC 256 rows x 128 B per tile (A+B of a 128x128 tile); 1: KS 32 k-rows x 1024 B
__global__ __launch_bounds__(256) void rate_kernel(const float* __restrict__ G, int ld, int ntile, int rows_total,
                                                    unsigned long long* __restrict__ cyc, int unaligned) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 32768];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wg = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntile; ++t) {
        char* stage = lds + (t & 3) * 32768;
        if (MODE == 0) {
            // 256 rows x 32 floats: 32 instr of 8 rows; this wave: instr wave, wave+4, ...
            const int row_base = (wg * 64) % (rows_total - 256);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int instr = wave + 4 * i;
                const int row = row_base + instr * 8 + (lane >> 3);
                const float* p = G + (size_t)row * ld + t * 32 + (lane & 7) * 4 + unaligned;
                __builtin_amdgcn_global_load_lds(p, (float*)(stage + instr * 1024), 16, 0, 0);
            }
        } else {
            // 32 k-rows x 256 floats (1 KiB per row): instr = k-row
            const int col_base = (wg * 64) % (ld - 256 - 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int instr = wave + 4 * i;
                const int krow = (t * 32 + instr) % rows_total;
                const float* p = G + (size_t)krow * ld + col_base + lane * 4 + unaligned;
                __builtin_amdgcn_global_load_lds(p, (float*)(stage + instr * 1024), 16, 0, 0);
            }
        }
        // keep 3 tiles in flight
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[wg] = t1 - t0;
    // consume something so the LDS writes are not dead
    if (threadIdx.x == 1) cyc[gridDim.x + wg] = *(unsigned long long*)(lds + 64 * (wg & 7));
}

int main() {
    // ---- functional
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *d, *o;
    CK(hipMalloc(&d, 4096));
    CK(hipMalloc(&o, 1024));
    CK(hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice));
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(func_kernel, dim3(1), dim3(64), 0, 0, d, shift, o);
        CK(hipDeviceSynchronize());
        std::vector<float> r(256);
        CK(hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + shift);
        printf("functional: source offset %d floats (alignment %d B): %s (%d wrong)\n", shift, shift ? 4 * (shift & -shift) : 16, bad ? "WRONG" : "ok", bad);
    }
    // ---- rate
    const int rows = 1024 + 768, ld = 4120;   // operands of the fc1||fc3 GEMM: 29.5 MB, Infinity-Cache resident
    float* G;
    CK(hipMalloc(&G, (size_t)rows * ld * 4 + 4096));
    CK(hipMemset(G, 0, (size_t)rows * ld * 4 + 4096));
    unsigned long long* cyc;
    CK(hipMalloc(&cyc, 8 * 1024 * 2));
    const int ntile = 128;
    for (int mode = 0; mode < 2; ++mode)
        for (int una = 0; una < 2; ++una)
            for (int nwg : {64, 192, 256}) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0, 0);
                    if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(nwg), dim3(256), 0, 0, G, ld, ntile, rows, cyc, una);
                    else hipLaunchKernelGGL(rate_kernel<1>, dim3(nwg), dim3(256), 0, 0, G, ld, ntile, rows, cyc, una);
                    hipEventRecord(e1, 0);
                    CK(hipDeviceSynchronize());
                }
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> c(nwg);
                CK(hipMemcpy(c.data(), cyc, nwg * 8, hipMemcpyDeviceToHost));
                double avg = 0;
                for (auto x : c) avg += (double)x;
                avg /= nwg;
                const double bytes_wg = (double)ntile * 32768;
                printf("rate mode %d (%s) unaligned %d, %3d WGs: %.1f us, %.0f cycles/WG -> %.1f B/clk/CU, %.2f TB/s chip\n", mode,
                       mode ? "2 rows x... 1 KiB k-rows" : "8 rows x 128 B", una, nwg, ms * 1e3, avg, bytes_wg / avg,
                       bytes_wg * nwg / (ms * 1e-3) / 1e12);
            }
    return 0;
}
