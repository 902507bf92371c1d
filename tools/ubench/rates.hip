// micro-benchmark: what one wave-instruction of each class the Farneback kernels are made of costs a SIMD, in SHADER
// CYCLES read inside the kernel (s_memtime: independent of the clock the chip holds under load), at 1, 2 and 4 waves per
// SIMD.  One workgroup per CU (its LDS request keeps a second one out), 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD;
// every wave runs ITERS trips over 16 independent instructions of one class (inline asm on 16 registers of its own, so the
// compiler can neither merge nor reorder them); cycles per wave-instruction per SIMD = (t1 - t0) / (ITERS * 16 * waves per
// SIMD), median over the workgroups.  Also the in-kernel clock: delta s_memtime / delta s_memrealtime x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6).
//   hipcc -O2 --offload-arch=gfx950 -o rates rates.hip && ./rates [cus]      (cus: CUs to occupy, default all)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 4000
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Op { ADD_F32, FMA_F32, MUL_F32, PK_ADD_F32, PK_MUL_F32, PK_FMA_F32, ADD_F64, MUL_F64, FMA_F64, RCP_F64, CVT_F64_F32, CVT_F32_F64,
          MOV_B32, MOV_DPP_WAVE_SHR, MOV_DPP_ROW_SHR, CNDMASK, MED3_I32, ADD_U32, MUL_U24, MAD_U24, RNDNE_F32, LDS_READ_B128, LDS_READ_B64,
          LDS_READ2_B32, LDS_WRITE_B64, BPERMUTE, NOPS,
          CNDMASK_SGPR, CMP_CNDMASK, CMP_F32, CMP_E64, LSHL, AND_B32, SUB_F32, FLOOR_F32, CVT_I32_F32, CVT_F32_UBYTE0, MOV_B64, FMAC_F64, ADD_LSHL, ASHR,
          FMAC_F32, MAX_F32, MED3_F32, CVT_F64_U32, LDS_WRITE2ST64_B32, READFIRSTLANE, CMP_CNDMASK_SGPR };

template <int OP> __global__ void __launch_bounds__(1024) k(unsigned long long* stamps, float seed)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    float f[16];
    double d[16];
    typedef float v2 __attribute__((ext_vector_type(2)));
    typedef float v4 __attribute__((ext_vector_type(4)));
    v2 p[16];
    v4 q[4];
    for (int i = 0; i < 16; i++) { f[i] = seed + i + lane; d[i] = 1.0 + 1e-9 * (seed + i + lane); p[i] = v2{f[i], f[i] * 0.5f}; }
    for (int i = 0; i < 4; i++) q[i] = v4{0, 0, 0, 0};
    const float c = 1.0f + 1e-7f * seed;
    const double cd = 1.0 + 1e-12 * seed;
    const v2 cp = v2{c, c};
    int iv[16];
    unsigned long long sm[4] = {0, 0, 0, 0};
    int si[4] = {0, 0, 0, 0};
    const unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)blockIdx.x;      // wave-uniform: an SGPR pair
    for (int i = 0; i < 16; i++) iv[i] = i + lane;
    const unsigned a128 = (unsigned)(threadIdx.x & 63) * 16u, a64 = (unsigned)(threadIdx.x & 63) * 8u, a32 = (unsigned)(threadIdx.x & 63) * 4u;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
#define A_ADD_F32(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[i]) : "v"(c));
#define A_FMA_F32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(c));
#define A_MUL_F32(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "v"(c));
#define A_PK_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
#define A_PK_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
#define A_PK_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cp));
#define A_ADD_F64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define A_MUL_F64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define A_FMA_F64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));
#define A_RCP_F64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define A_CVT_D_F(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define A_CVT_F_D(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(f[(i + 1) & 15]));
#define A_DPP_W(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
#define A_DPP_R(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(c) : );
#define A_MED3(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(iv[i]) : "v"(iv[(i + 1) & 15]), "v"(iv[(i + 2) & 15]));
#define A_ADD_U32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[i]) : "v"(lane));
#define A_MUL_U24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(iv[i]) : "v"(lane));
#define A_MAD_U24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(iv[i]) : "v"(lane));
#define A_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(f[i]));
#define A_LDS128(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i & 3]) : "v"(a128), "n"((i & 3) * 1024));
#define A_LDS64(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(a64), "n"(i * 512));
#define A_LDS2X32(i) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d[i]) : "v"(a32), "n"(i), "n"(i + 64));
#define A_LDSW64(i) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(a64), "v"(d[i]), "n"(i * 512) : "memory");
#define A_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(f[i]) : "v"(a32));
#define A_NOP(i) asm volatile("s_nop 0");
#define A_CNDMASK_S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c), "s"(smask));
#define A_CMP_CNDMASK(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(c) : "vcc");
#define A_CMP_F32(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0" :: "v"(f[i]), "v"(c) : "vcc");
#define A_CMP_E64(i) asm volatile("v_cmp_lt_f32_e64 %0, %2, %1" : "=s"(sm[i & 3]) : "v"(f[i]), "v"(c));
#define A_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(iv[i]));
#define A_AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(iv[i]) : "v"(lane));
#define A_SUB_F32(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));
#define A_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(f[i]));
#define A_CVT_I_F(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(iv[i]) : "v"(f[i]));
#define A_CVT_F_UB(i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(iv[i]));
#define A_MOV_B64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 15]));
#define A_FMAC_F64(i) asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(d[i]) : "v"(cd));
#define A_ADD_LSHL(i) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(iv[i]) : "v"(lane));
#define A_ASHR(i) asm volatile("v_ashrrev_i32 %0, 5, %0" : "+v"(iv[i]));
#define A_FMAC_F32(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(f[i]) : "v"(c));
#define A_MAX_F32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c));
#define A_MED3_F32(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(c));
#define A_CVT_D_U(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(iv[i]));
#define A_LDSW2ST(i) asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(a32), "v"(f[i]), "v"(f[(i + 1) & 15]), "n"(i & 7), "n"((i & 7) + 8) : "memory");
#define A_CMP_CND_S(i) asm volatile("v_cmp_lt_f32_e64 %1, %2, %0\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(f[i]), "=&s"(sm[i & 3]) : "v"(c));
#define A_RFL(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(si[i & 3]) : "v"(iv[i]));
        if (OP == ADD_F32) { REP16(A_ADD_F32) }
        if (OP == FMA_F32) { REP16(A_FMA_F32) }
        if (OP == MUL_F32) { REP16(A_MUL_F32) }
        if (OP == PK_ADD_F32) { REP16(A_PK_ADD) }
        if (OP == PK_MUL_F32) { REP16(A_PK_MUL) }
        if (OP == PK_FMA_F32) { REP16(A_PK_FMA) }
        if (OP == ADD_F64) { REP16(A_ADD_F64) }
        if (OP == MUL_F64) { REP16(A_MUL_F64) }
        if (OP == FMA_F64) { REP16(A_FMA_F64) }
        if (OP == RCP_F64) { REP16(A_RCP_F64) }
        if (OP == CVT_F64_F32) { REP16(A_CVT_D_F) }
        if (OP == CVT_F32_F64) { REP16(A_CVT_F_D) }
        if (OP == MOV_B32) { REP16(A_MOV) }
        if (OP == MOV_DPP_WAVE_SHR) { REP16(A_DPP_W) }
        if (OP == MOV_DPP_ROW_SHR) { REP16(A_DPP_R) }
        if (OP == CNDMASK) { REP16(A_CNDMASK) }
        if (OP == MED3_I32) { REP16(A_MED3) }
        if (OP == ADD_U32) { REP16(A_ADD_U32) }
        if (OP == MUL_U24) { REP16(A_MUL_U24) }
        if (OP == MAD_U24) { REP16(A_MAD_U24) }
        if (OP == RNDNE_F32) { REP16(A_RNDNE) }
        if (OP == LDS_READ_B128) { REP16(A_LDS128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == LDS_READ_B64) { REP16(A_LDS64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == LDS_READ2_B32) { REP16(A_LDS2X32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == LDS_WRITE_B64) { REP16(A_LDSW64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == BPERMUTE) { REP16(A_BPERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == NOPS) { REP16(A_NOP) }
        if (OP == CNDMASK_SGPR) { REP16(A_CNDMASK_S) }
        if (OP == CMP_CNDMASK) { REP16(A_CMP_CNDMASK) }
        if (OP == CMP_F32) { REP16(A_CMP_F32) }
        if (OP == CMP_E64) { REP16(A_CMP_E64) }
        if (OP == LSHL) { REP16(A_LSHL) }
        if (OP == AND_B32) { REP16(A_AND) }
        if (OP == SUB_F32) { REP16(A_SUB_F32) }
        if (OP == FLOOR_F32) { REP16(A_FLOOR) }
        if (OP == CVT_I32_F32) { REP16(A_CVT_I_F) }
        if (OP == CVT_F32_UBYTE0) { REP16(A_CVT_F_UB) }
        if (OP == MOV_B64) { REP16(A_MOV_B64) }
        if (OP == FMAC_F64) { REP16(A_FMAC_F64) }
        if (OP == ADD_LSHL) { REP16(A_ADD_LSHL) }
        if (OP == ASHR) { REP16(A_ASHR) }
        if (OP == FMAC_F32) { REP16(A_FMAC_F32) }
        if (OP == MAX_F32) { REP16(A_MAX_F32) }
        if (OP == MED3_F32) { REP16(A_MED3_F32) }
        if (OP == CVT_F64_U32) { REP16(A_CVT_D_U) }
        if (OP == LDS_WRITE2ST64_B32) { REP16(A_LDSW2ST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == READFIRSTLANE) { REP16(A_RFL) }
        if (OP == CMP_CNDMASK_SGPR) { REP16(A_CMP_CND_S) }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int i = 0; i < 16; i++) acc += f[i] + (float)d[i] + p[i].x + p[i].y + (float)iv[i];
    for (int i = 0; i < 4; i++) acc += q[i].x + q[i].w + (float)(sm[i] & 255) + (float)si[i];
    if (acc == 12345.678f) lds[0] = acc;                      // keeps every register live without a store anybody pays for
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

struct Res { double cyc; double ghz; };
template <int OP> Res run(unsigned long long* d_st, int cus, int wps)
{
    const int threads = 256 * wps, waves = cus * 4 * wps;
    const size_t lds = 96 * 1024;                             // one workgroup per CU
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(threads), lds, 0, d_st, 1.0f);
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(threads), lds, 0, d_st, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(2 * waves);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int w = 0; w < waves; w++) { cyc[w] = (double)st[2 * w]; clk[w] = (double)st[2 * w] / (double)st[2 * w + 1] * 0.1; }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    return {cyc[waves / 2] / ((double)ITERS * 16 * wps), clk[waves / 2]};
}

#define ROW(OP, NAME) { Res a = run<OP>(d, cus, 1), b = run<OP>(d, cus, 2), c4 = run<OP>(d, cus, 4); \
    printf("%-26s %7.2f %7.2f %7.2f     %.2f / %.2f / %.2f\n", NAME, a.cyc, b.cyc, c4.cyc, a.ghz, b.ghz, c4.ghz); }

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = argc > 1 ? atoi(argv[1]) : prop.multiProcessorCount;
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * 256 * 16 * 2);
    printf("# %s, %d CUs occupied (one workgroup each), %d trips x 16 independent wave-instructions per wave\n", prop.gcnArchName, cus, ITERS);
    printf("# shader cycles per wave-instruction per SIMD (s_memtime) at 1 / 2 / 4 waves per SIMD; in-kernel clock GHz (s_memtime / s_memrealtime x 100 MHz)\n");
    printf("%-26s %7s %7s %7s     %s\n", "class", "1w", "2w", "4w", "GHz at 1w / 2w / 4w");
    ROW(NOPS, "s_nop 0")
    ROW(ADD_F32, "v_add_f32")
    ROW(MUL_F32, "v_mul_f32")
    ROW(FMA_F32, "v_fma_f32")
    ROW(PK_ADD_F32, "v_pk_add_f32")
    ROW(PK_MUL_F32, "v_pk_mul_f32")
    ROW(PK_FMA_F32, "v_pk_fma_f32")
    ROW(ADD_F64, "v_add_f64")
    ROW(MUL_F64, "v_mul_f64")
    ROW(FMA_F64, "v_fma_f64")
    ROW(RCP_F64, "v_rcp_f64")
    ROW(CVT_F64_F32, "v_cvt_f64_f32")
    ROW(CVT_F32_F64, "v_cvt_f32_f64")
    ROW(MOV_B32, "v_mov_b32")
    ROW(MOV_DPP_WAVE_SHR, "v_mov_b32_dpp wave_shr:1")
    ROW(MOV_DPP_ROW_SHR, "v_mov_b32_dpp row_shr:1")
    ROW(CNDMASK, "v_cndmask_b32 (VCC never written)")
    ROW(MED3_I32, "v_med3_i32")
    ROW(ADD_U32, "v_add_u32")
    ROW(MUL_U24, "v_mul_u32_u24")
    ROW(MAD_U24, "v_mad_u32_u24")
    ROW(RNDNE_F32, "v_rndne_f32")
    ROW(CNDMASK_SGPR, "v_cndmask_b32_e64 (sgpr)")
    ROW(CMP_CNDMASK, "v_cmp_lt_f32 + v_cndmask (pair)")
    ROW(CMP_CNDMASK_SGPR, "v_cmp_e64 + v_cndmask_e64 (pair)")
    ROW(CMP_F32, "v_cmp_lt_f32 vcc")
    ROW(CMP_E64, "v_cmp_lt_f32_e64 sgpr")
    ROW(LSHL, "v_lshlrev_b32")
    ROW(AND_B32, "v_and_b32")
    ROW(ASHR, "v_ashrrev_i32")
    ROW(ADD_LSHL, "v_add_lshl_u32")
    ROW(SUB_F32, "v_sub_f32")
    ROW(FMAC_F32, "v_fmac_f32")
    ROW(MAX_F32, "v_max_f32")
    ROW(MED3_F32, "v_med3_f32")
    ROW(FLOOR_F32, "v_floor_f32")
    ROW(CVT_I32_F32, "v_cvt_i32_f32")
    ROW(CVT_F32_UBYTE0, "v_cvt_f32_ubyte0")
    ROW(CVT_F64_U32, "v_cvt_f64_u32")
    ROW(MOV_B64, "v_mov_b64")
    ROW(FMAC_F64, "v_fmac_f64")
    ROW(READFIRSTLANE, "v_readfirstlane_b32")
    ROW(LDS_WRITE2ST64_B32, "ds_write2st64_b32 (16/wait)")
    ROW(LDS_READ_B128, "ds_read_b128 (16/wait)")
    ROW(LDS_READ_B64, "ds_read_b64 (16/wait)")
    ROW(LDS_READ2_B32, "ds_read2_b32 (16/wait)")
    ROW(LDS_WRITE_B64, "ds_write_b64 (16/wait)")
    ROW(BPERMUTE, "ds_bpermute_b32 (16/wait)")
    return 0;
}
