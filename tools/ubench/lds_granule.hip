// How many 128-thread workgroups fit a CU as a function of dynamic LDS bytes (the allocation granule decides whether
// k_farneback_iter's 25.3 KB + a few hundred bytes still fits six times into 160 KB).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(128) void k(float* out) { extern __shared__ float s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); out[blockIdx.x] = s[(threadIdx.x + 1) & 127]; }
int main()
{
    for (int bytes = 25600; bytes <= 28672; bytes += 128) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 128, bytes);
        printf("%d:%d ", bytes, n);
    }
    printf("\n");
    return 0;
}
