#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 4);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
    unsigned h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%d:%x ", i, h[i]);
    printf("\n");
    return 0;
}
