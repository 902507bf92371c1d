// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the access widths the fused kernel uses:
// streams a known number of bytes with 4-, 8- (dword-aligned, overlapping like the bilinear taps) and
// 16-byte per-lane loads.  Run under `rocprofv3 --pmc FETCH_SIZE` and compare with the byte counts printed.
#include <hip/hip_runtime.h>
#include <cstdio>
struct __attribute__((packed, aligned(4))) f2u { float a, b; };
__global__ void rd4(const float* p, size_t n, float* out) { float s = 0; for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) s += p[i]; if (s == 1.2345f) out[0] = s; }
__global__ void rd8u(const float* p, size_t n, float* out) { float s = 0; for (size_t i = blockIdx.x * 256ull + threadIdx.x; i + 1 < n; i += gridDim.x * 256ull) { f2u v = *(const f2u*)(p + i); s += v.a + v.b; } if (s == 1.2345f) out[0] = s; }
__global__ void rd16(const float4* p, size_t n4, float* out) { float s = 0; for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += gridDim.x * 256ull) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; } if (s == 1.2345f) out[0] = s; }
int main()
{
    size_t n = (size_t)1 << 30;  // 4 GiB of floats: well past the 256 MiB Infinity Cache
    float* d; float* o;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&o, 4); (void)hipMemset(d, 0, n * 4);
    hipLaunchKernelGGL(rd4, dim3(8192), dim3(256), 0, 0, d, n, o);
    hipLaunchKernelGGL(rd8u, dim3(8192), dim3(256), 0, 0, d, n, o);
    hipLaunchKernelGGL(rd16, dim3(8192), dim3(256), 0, 0, (const float4*)d, n / 4, o);
    (void)hipDeviceSynchronize();
    printf("bytes per kernel: %zu (rd4, rd8u touch the same 4 GiB once; rd8u requests each dword twice)\n", n * 4);
    return 0;
}
