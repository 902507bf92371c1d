// Does memory placed with HIP's virtual-memory API (reserve / create / map / set access) behave like hipMalloc'ed memory for
// copies and kernels, at the minimum and at the recommended granularity?  (the FDN_GUARD_ALLOC debugging allocator depends on it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_add1(const float* in, float* out, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) out[i] = in[i] + 1.f; }
int main()
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CHECK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CHECK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    for (size_t gran : {gmin, grec}) {
        for (size_t n : {(size_t)1000, (size_t)300000, (size_t)5000000}) {
            const size_t bytes = n * 4, need = (bytes + 15) & ~(size_t)15, map = (need + gran - 1) / gran * gran;
            void* va[2]; hipMemGenericAllocationHandle_t hd[2]; float* p[2];
            for (int i = 0; i < 2; i++) {
                CHECK(hipMemAddressReserve(&va[i], map + 2 * gran, gran, nullptr, 0));
                CHECK(hipMemCreate(&hd[i], map, &prop, 0));
                CHECK(hipMemMap((char*)va[i] + gran, map, 0, hd[i], 0));
                hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                CHECK(hipMemSetAccess((char*)va[i] + gran, map, &acc, 1));
                p[i] = (float*)((char*)va[i] + gran + (map - need));
            }
            std::vector<float> h(n), back(n);
            for (size_t i = 0; i < n; i++) h[i] = (float)(i % 1000);
            CHECK(hipMemcpy(p[0], h.data(), bytes, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_add1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p[0], p[1], n);
            CHECK(hipGetLastError());
            CHECK(hipMemcpy(back.data(), p[1], bytes, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < n; i++) bad += back[i] != h[i] + 1.f;
            printf("gran %zu, %zu floats: %zu wrong\n", gran, n, bad);
            for (int i = 0; i < 2; i++) { CHECK(hipMemUnmap((char*)va[i] + gran, map)); CHECK(hipMemRelease(hd[i])); CHECK(hipMemAddressFree(va[i], map + 2 * gran)); }
        }
    }
    return 0;
}
