// micro-benchmark: issue rates of the instruction classes the fused kernel is made of
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int OP> __global__ void k(double* out, int iters, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7;
    int lane = threadIdx.x & 63;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }           // v_add_f64 x8
        if (OP == 1) { a0 *= a1; a1 *= a2; a2 *= a3; a3 *= a4; a4 *= a5; a5 *= a6; a6 *= a7; a7 *= a0; }           // v_mul_f64 x8
        if (OP == 2) { a0 = fma(a0, a1, a2); a1 = fma(a1, a2, a3); a2 = fma(a2, a3, a4); a3 = fma(a3, a4, a5); a4 = fma(a4, a5, a6); a5 = fma(a5, a6, a7); a6 = fma(a6, a7, a0); a7 = fma(a7, a0, a1); }
        if (OP == 3) { f0 += f1; f1 += f2; f2 += f3; f3 += f4; f4 += f5; f5 += f6; f6 += f7; f7 += f0; }           // v_add_f32 x8
        if (OP == 4) { f0 = __shfl(f0, (lane + 1) & 63, 64); f1 = __shfl(f1, (lane + 2) & 63, 64); f2 = __shfl(f2, (lane + 3) & 63, 64); f3 = __shfl(f3, (lane + 4) & 63, 64);
                       f4 = __shfl(f4, (lane + 5) & 63, 64); f5 = __shfl(f5, (lane + 6) & 63, 64); f6 = __shfl(f6, (lane + 7) & 63, 64); f7 = __shfl(f7, (lane + 8) & 63, 64); }   // ds_bpermute x8
        if (OP == 5) { a0 = 1.0 / a0; a1 = 1.0 / a1; a2 = 1.0 / a2; a3 = 1.0 / a3; a4 = 1.0 / a4; a5 = 1.0 / a5; a6 = 1.0 / a6; a7 = 1.0 / a7; }  // f64 division x8
        if (OP == 6) { a0 = (double)f0; a1 = (double)f1; a2 = (double)f2; a3 = (double)f3; f4 = (float)a4; f5 = (float)a5; f6 = (float)a6; f7 = (float)a7; a4 += a0; a5 += a1; f0 += f4; f1 += f5; } // cvt mix
        if (OP == 7) { f0 = __builtin_amdgcn_update_dpp(f0, f0, 0x138, 0xf, 0xf, false); f1 = __builtin_amdgcn_update_dpp(f1, f1, 0x130, 0xf, 0xf, false);
                       f2 = __builtin_amdgcn_update_dpp(f2, f2, 0x138, 0xf, 0xf, false); f3 = __builtin_amdgcn_update_dpp(f3, f3, 0x130, 0xf, 0xf, false);
                       f4 = __builtin_amdgcn_update_dpp(f4, f4, 0x138, 0xf, 0xf, false); f5 = __builtin_amdgcn_update_dpp(f5, f5, 0x130, 0xf, 0xf, false);
                       f6 = __builtin_amdgcn_update_dpp(f6, f6, 0x138, 0xf, 0xf, false); f7 = __builtin_amdgcn_update_dpp(f7, f7, 0x130, 0xf, 0xf, false); }   // v_mov_dpp wave_shr/shl x8
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int OP> void run(const char* name, double* d, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 SIMDs x waves) / 4 waves per block
    int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double inst_per_simd = (double)iters * 8 * waves_per_simd;
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.1f cycles @2.4GHz)\n", name, waves_per_simd, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}
int main()
{
    double* d; hipMalloc(&d, sizeof(double) * 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f64", d, w); run<1>("v_mul_f64", d, w); run<2>("v_fma_f64", d, w); run<3>("v_add_f32", d, w);
        run<4>("ds_bpermute", d, w); run<5>("f64 division", d, w); run<6>("cvt mix(12)", d, w); run<7>("dpp wave_shift", d, w);
    }
    return 0;
}
