#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main() {
    void* d; hipMalloc(&d, 64u << 20);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    size_t sizes[] = {256u << 10, 1092000, 6u << 20, 40u << 20};
    for (size_t s : sizes) {
        char* a = (char*)malloc(s); memset(a, 1, s);
        fprintf(stderr, "#### H2D %zu\n", s);
        hipMemcpyAsync(d, a, s, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        fprintf(stderr, "#### D2H %zu\n", s);
        hipMemcpyAsync(a, d, s, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        free(a);
    }
    return 0;
}
