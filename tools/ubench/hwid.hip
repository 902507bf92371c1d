// Which SIMD does wave w of a 4-wave workgroup land on?  Same launch shape as k_farneback_fused
// (256 threads, ~40 KB LDS, 4 workgroups per CU), some busy work so that workgroups co-reside.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 4) void k(unsigned* out, int spin)
{
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) { a = a * 1.0001f + 0.5f; lds[threadIdx.x] = a; __syncthreads(); a += lds[(threadIdx.x + 1) & 255]; }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (a == 123.f);
    }
}
int main()
{
    const int nb = 10240;
    unsigned* d; hipMalloc(&d, nb * 4 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 40000, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
    long hist[4][4] = {};   // [wave in block][simd]
    long distinct[5] = {};  // number of distinct SIMDs used by a block
    for (int b = 0; b < nb; b++) {
        int used = 0;
        for (int w = 0; w < 4; w++) { int simd = (h[(b * 4 + w) * 2] >> 4) & 3; hist[w][simd]++; used |= 1 << simd; }
        distinct[__builtin_popcount(used)]++;
    }
    for (int w = 0; w < 4; w++) printf("wave %d -> simd: %ld %ld %ld %ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("blocks using 1/2/3/4 distinct SIMDs: %ld %ld %ld %ld\n", distinct[1], distinct[2], distinct[3], distinct[4]);
    for (int b = 0; b < 24; b++) {
        printf("block %d xcc %u cu %u se %u: simd", b, h[b * 8 + 1] & 15, (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
        for (int w = 0; w < 4; w++) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
