// Which SIMD does wave w of an EIGHT-wave workgroup land on (the two-band build of k_farneback_fused: 512 threads,
// ~80 KB LDS, 2 workgroups per CU)?  Prints the wave -> SIMD histogram and the mapping of the first blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(512, 4) void k(unsigned* out, int spin)
{
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) { a = a * 1.0001f + 0.5f; lds[threadIdx.x] = a; __syncthreads(); a += lds[(threadIdx.x + 1) & 511]; }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (a == 123.f);
    }
}
int main()
{
    const int nb = 5120;
    unsigned* d; hipMalloc(&d, nb * 8 * 2 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 80000, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 16);
    hipMemcpy(h.data(), d, nb * 64, hipMemcpyDeviceToHost);
    long hist[8][4] = {};
    std::map<std::vector<int>, long> patterns;
    for (int b = 0; b < nb; b++) {
        std::vector<int> p;
        for (int w = 0; w < 8; w++) { int simd = (h[(b * 8 + w) * 2] >> 4) & 3; hist[w][simd]++; p.push_back(simd); }
        patterns[p]++;
    }
    for (int w = 0; w < 8; w++) printf("wave %d -> simd: %ld %ld %ld %ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (auto& kv : patterns) { printf("pattern"); for (int s : kv.first) printf(" %d", s); printf(" : %ld blocks\n", kv.second); }
    for (int b = 0; b < 40; b++) {
        printf("block %d xcc %u cu %u se %u: simd", b, h[b * 16 + 1] & 15, (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
        for (int w = 0; w < 8; w++) printf(" %u", (h[(b * 8 + w) * 2] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
