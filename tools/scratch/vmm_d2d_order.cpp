// Is hipMemcpyAsync (device to device) INTO memory placed with the virtual-memory API ordered with the kernels before and after
// it on the same stream, as it is for hipMalloc'ed memory?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_fill(float* p, float v, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = v; }
__global__ void k_copy(const float* in, float* out, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = in[i]; }
static int vmm_alloc(float** out, size_t bytes)
{
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 4096; void* va; hipMemGenericAllocationHandle_t hd;
    const size_t need = (bytes + 15) & ~(size_t)15, map = (need + gran - 1) / gran * gran;
    CHECK(hipMemAddressReserve(&va, map + 2 * gran, gran, nullptr, 0)); CHECK(hipMemCreate(&hd, map, &prop, 0));
    CHECK(hipMemMap((char*)va + gran, map, 0, hd, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess((char*)va + gran, map, &acc, 1));
    *out = (float*)((char*)va + gran + (map - need));
    return 0;
}
int main(int argc, char** argv)
{
    const bool vmm = argc > 1 && argv[1][0] == '1';
    const size_t n = 14 * 131 * 97, part = 6 * 131 * 97, off = 4 * 131 * 97;
    float *src, *dst, *res;
    CHECK(hipMalloc((void**)&src, part * 4)); CHECK(hipMalloc((void**)&res, n * 4));
    if (vmm) { if (vmm_alloc(&dst, n * 4)) return 2; } else CHECK(hipMalloc((void**)&dst, n * 4));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<float> back(n);
    size_t bad_rounds = 0;
    for (int round = 0; round < 200; round++) {
        hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, src, (float)(round + 2), part);
        hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, dst, 1.f, n);                       // pad value everywhere
        CHECK(hipMemcpyAsync(dst + off, src, part * 4, hipMemcpyDeviceToDevice, st));              // the volume's slices in the middle
        hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, dst, res, n);                       // a consumer kernel
        CHECK(hipMemcpyAsync(back.data(), res, n * 4, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) bad += back[i] != ((i >= off && i < off + part) ? (float)(round + 2) : 1.f);
        if (bad) { bad_rounds++; if (bad_rounds <= 3) printf("round %d: %zu wrong values\n", round, bad); }
    }
    printf("%s destination: %zu of 200 rounds wrong\n", vmm ? "VMM" : "hipMalloc", bad_rounds);
    return 0;
}
