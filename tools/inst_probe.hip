// Developer tool: issue cost of the VALU instructions the traversal loops are made of, relative to v_fma_f32, measured on the
// GPU: N back-to-back copies on 8 independent registers per lane, 8 waves per SIMD, every CU busy.
// hipcc --offload-arch=gfx950 -O3 -o tools/_bin/inst_probe tools/inst_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(float * out, int iters)                           \
    {                                                                                                 \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;  \
        float b = 1.0001f, c = 0.5f;                                                                  \
        for (int i = 0; i < iters; ++i) {                                                             \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   : "v"(b), "v"(c) : "vcc", "s2", "s3", "s4"); \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;           \
    }

#define A_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_FMAMIX(n) "v_fma_mix_f32 %" #n ", %" #n ", %8, %9 op_sel_hi:[1,0,0]\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define A_MINDPP(n) "v_min_u32_dpp %" #n ", %" #n ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_MOVDPP(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define A_BCNT(n) "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_CNDMASK64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[2:3]\n"
#define A_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define A_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define A_CMPSEL_VCC(n) "v_cmp_lt_f32 vcc, %" #n ", %8\ns_nop 1\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_CMPSEL_SGPR(n) "v_cmp_lt_f32_e64 s[2:3], %" #n ", %8\ns_nop 1\nv_cndmask_b32_e64 %" #n ", %" #n ", %9, s[2:3]\n"
#define A_CMPSEL2_VCC(n) "v_cmp_lt_f32 vcc, %" #n ", %8\ns_nop 1\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_CMPSEL2_SGPR(n) "v_cmp_lt_f32_e64 s[2:3], %" #n ", %8\ns_nop 1\nv_cndmask_b32_e64 %" #n ", %" #n ", %9, s[2:3]\nv_cndmask_b32_e64 %" #n ", %" #n ", %8, s[2:3]\n"
#define A_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define A_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x32\n"
#define A_CVT(n) "v_cvt_f32_f16 %" #n ", %" #n "\n"
#define A_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define A_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define A_MIN(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define A_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define A_DIVFIX(n) "v_div_fixup_f32 %" #n ", %" #n ", %8, %9\n"
#define A_BPERM(n) "ds_bpermute_b32 %" #n ", %8, %" #n "\ns_waitcnt lgkmcnt(0)\n"

BODY(fma, A_FMA) BODY(fma_mix, A_FMAMIX) BODY(perm, A_PERM) BODY(min_dpp, A_MINDPP) BODY(mov_dpp, A_MOVDPP) BODY(max3, A_MAX3)
BODY(and_or, A_ANDOR) BODY(bcnt, A_BCNT) BODY(cndmask, A_CNDMASK) BODY(cmp, A_CMP) BODY(lshl_add, A_LSHLADD) BODY(mul_lo, A_MULLO)
BODY(mul_u24, A_MUL24) BODY(bitop3, A_BITOP3) BODY(cvt_f16, A_CVT) BODY(rcp, A_RCP) BODY(sqrt, A_SQRT) BODY(min, A_MIN)
BODY(div_fixup, A_DIVFIX) BODY(bpermute, A_BPERM) BODY(cndmask_sgpr, A_CNDMASK64) BODY(cmpsel_vcc, A_CMPSEL_VCC) BODY(cmpsel_sgpr, A_CMPSEL_SGPR) BODY(cmpsel2_vcc, A_CMPSEL2_VCC) BODY(cmpsel2_sgpr, A_CMPSEL2_SGPR) BODY(add_f32, A_ADD) BODY(mul_f32, A_MUL) BODY(mov, A_MOV)

// ---- round 4: which integer / logic / compare / select forms issue at the fast rate of v_fma_f32 / v_mov_b32 / v_bitop3_b32 ----
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_LSHL(n) "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define A_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_SUBU(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define A_MAXF(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define A_MINU(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define A_MINI(n) "v_min_i32 %" #n ", %" #n ", %8\n"
#define A_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_MIN3(n) "v_min3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define A_FMANEG(n) "v_fma_f32 %" #n ", %" #n ", %8, -%9\n"
#define A_SUBF(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define A_CMPLE64(n) "v_cmp_le_f32_e64 s[2:3], %" #n ", %8\n"
#define A_CMPNEU(n) "v_cmp_ne_u32 vcc, %" #n ", %8\n"
#define A_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 1, 1\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %" #n ", %8, %9\n"
#define A_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 2, %8\n"
#define A_OR3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define A_ADDLSHL(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 2\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 5\n"
#define A_CVTU(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define A_ADDDPP(n) "v_add_f32_dpp %" #n ", %" #n ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_ACCW(n) "v_accvgpr_write_b32 a" #n ", %" #n "\n"
#define A_ACCR(n) "v_accvgpr_read_b32 %" #n ", a" #n "\n"
#define A_MOVI(n) "v_mov_b32 %" #n ", -1\n"
#define A_FMASGPR(n) "v_fma_f32 %" #n ", %" #n ", s4, %9\n"
#define A_MAXSGPR(n) "v_max_f32 %" #n ", s4, %" #n "\n"
#define A_MULLEG(n) "v_mul_legacy_f32 %" #n ", %" #n ", %8\n"
#define A_LDEXP(n) "v_ldexp_f32 %" #n ", %" #n ", 1\n"
#define A_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n"
BODY(and_b32, A_AND) BODY(or_b32, A_OR) BODY(xor_b32, A_XOR) BODY(lshlrev, A_LSHL) BODY(lshrrev, A_LSHR) BODY(add_u32, A_ADDU) BODY(sub_u32, A_SUBU)
BODY(add3_u32, A_ADD3) BODY(mad_u32_u24, A_MAD24) BODY(max_f32, A_MAXF) BODY(min_u32, A_MINU) BODY(min_i32, A_MINI) BODY(med3_f32, A_MED3) BODY(min3_f32, A_MIN3)
BODY(fmac_f32, A_FMAC) BODY(fma_neg, A_FMANEG) BODY(sub_f32, A_SUBF) BODY(cmp_le_e64, A_CMPLE64) BODY(cmp_ne_u32, A_CMPNEU) BODY(bfe_u32, A_BFE) BODY(bfi_b32, A_BFI)
BODY(lshl_or, A_LSHLOR) BODY(or3, A_OR3) BODY(add_lshl, A_ADDLSHL) BODY(alignbit, A_ALIGNBIT) BODY(cvt_f32_u32, A_CVTU) BODY(add_f32_dpp, A_ADDDPP)
BODY(mov_imm, A_MOVI) BODY(fma_sgpr, A_FMASGPR) BODY(max_sgpr, A_MAXSGPR) BODY(mul_legacy, A_MULLEG) BODY(ldexp_f32, A_LDEXP)

// ---- binary64 forms (the correctly rounded exp of rvb_math.h air_attenuation is made of these) ----
#define BODY64(NAME, ASM)                                                                            \
    __global__ __launch_bounds__(256) void k_##NAME(float * out, int iters)                           \
    {                                                                                                 \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;  \
        double b = 1.0001, c = 0.5;                                                                   \
        for (int i = 0; i < iters; ++i) {                                                             \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc"); \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float) (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);  \
    }
#define A_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
#define A_MUL64(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
#define A_ADD64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define A_RNDNE64(n) "v_rndne_f64 %" #n ", %" #n "\n"
#define A_LDEXP64(n) "v_ldexp_f64 %" #n ", %" #n ", 1\n"
// conversions between a 64-bit pair (operand n) and a 32-bit register (operand n + 8)
#define BODY64M(NAME, ASM)                                                                           \
    __global__ __launch_bounds__(256) void k_##NAME(float * out, int iters)                           \
    {                                                                                                 \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;  \
        float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;                 \
        for (int i = 0; i < iters; ++i) {                                                             \
            asm volatile(ASM(0, 8) ASM(1, 9) ASM(2, 10) ASM(3, 11) ASM(4, 12) ASM(5, 13) ASM(6, 14) ASM(7, 15)   \
                         ASM(0, 8) ASM(1, 9) ASM(2, 10) ASM(3, 11) ASM(4, 12) ASM(5, 13) ASM(6, 14) ASM(7, 15)   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),       \
                           "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));      \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float) (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;  \
    }
#define A_CVT3264(n, m) "v_cvt_f32_f64 %" #m ", %" #n "\n"
#define A_CVT6432(n, m) "v_cvt_f64_f32 %" #n ", %" #m "\n"
#define A_CVTI64(n, m) "v_cvt_i32_f64 %" #m ", %" #n "\n"
#define A_RCP64(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define A_MINU64(n) "v_cmp_lt_u64 vcc, %" #n ", %8\n"
BODY64(fma_f64, A_FMA64) BODY64(mul_f64, A_MUL64) BODY64(add_f64, A_ADD64) BODY64(rndne_f64, A_RNDNE64) BODY64(ldexp_f64, A_LDEXP64)
BODY64M(cvt_f32_f64, A_CVT3264) BODY64M(cvt_f64_f32, A_CVT6432) BODY64M(cvt_i32_f64, A_CVTI64) BODY64(rcp_f64, A_RCP64) BODY64(cmp_u64, A_MINU64)

// 64-bit register pairs: packed binary32 arithmetic and the 64-bit move
#define A_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_PKMUL64(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define A_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define A_PKMOV(n) "v_pk_mov_b32 %" #n ", %" #n ", %8\n"
#define A_MOV64(n) "v_mov_b64 %" #n ", %8\n"
#define A_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 2, %8\n"
BODY64(pk_fma_f32, A_PKFMA) BODY64(pk_mul_f32, A_PKMUL64) BODY64(pk_add_f32, A_PKADD) BODY64(pk_mov_b32, A_PKMOV) BODY64(mov_b64, A_MOV64) BODY64(lshl_add_u64, A_LSHLADD64)

static int g_blocks_per_cu = 8;
template <class K> float run(K kernel, float * out)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kernel, dim3(256 * g_blocks_per_cu), dim3(256), 0, 0, out, 4096);      // g_blocks_per_cu blocks x 4 waves per CU = that many waves per SIMD
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r && ms < best) best = ms;
    }
    return best;
}

int main()
{
    float * out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    // cycles per wave-instruction per SIMD = time x clock / (instructions per wave x waves per SIMD); the clock is not known to
    // the probe, so every form is printed relative to v_fma_f32 at the same occupancy and in ns per wave-instruction per SIMD
    const int occupancies[4] = {1, 2, 4, 8};
    for (int o = 0; o < 4; ++o) {
        g_blocks_per_cu = occupancies[o];
        const double per_simd = 4096.0 * 16.0 * g_blocks_per_cu;        // wave-instructions issued on one SIMD
        const float base = run(k_fma, out);
        printf("== %d wave(s) per SIMD ==\n", g_blocks_per_cu);
        printf("%-16s %.3f ms = 1.00  (%.3f ns per wave-instruction per SIMD)\n", "v_fma_f32", base, base * 1e6 / per_simd);
#define SHOW(NAME) { const float t = run(k_##NAME, out); printf("%-16s %.3f ms = %.2f  (%.3f ns)\n", #NAME, t, t / base, t * 1e6 / per_simd); }
        SHOW(fma_mix) SHOW(perm) SHOW(min_dpp) SHOW(mov_dpp) SHOW(max3) SHOW(and_or) SHOW(bcnt) SHOW(cndmask) SHOW(cmp) SHOW(lshl_add)
        SHOW(mul_lo) SHOW(mul_u24) SHOW(bitop3) SHOW(cvt_f16) SHOW(rcp) SHOW(sqrt) SHOW(min) SHOW(div_fixup) SHOW(bpermute)
        SHOW(fma_f64) SHOW(mul_f64) SHOW(add_f64) SHOW(rndne_f64) SHOW(ldexp_f64) SHOW(cvt_f32_f64) SHOW(cvt_f64_f32) SHOW(cvt_i32_f64)
        SHOW(rcp_f64) SHOW(cmp_u64) SHOW(cndmask_sgpr) SHOW(cmpsel_vcc) SHOW(cmpsel_sgpr) SHOW(cmpsel2_vcc) SHOW(cmpsel2_sgpr) SHOW(add_f32) SHOW(mul_f32) SHOW(mov)
        SHOW(and_b32) SHOW(or_b32) SHOW(xor_b32) SHOW(lshlrev) SHOW(lshrrev) SHOW(add_u32) SHOW(sub_u32) SHOW(add3_u32) SHOW(mad_u32_u24) SHOW(max_f32)
        SHOW(min_u32) SHOW(min_i32) SHOW(med3_f32) SHOW(min3_f32) SHOW(fmac_f32) SHOW(fma_neg) SHOW(sub_f32) SHOW(cmp_le_e64) SHOW(cmp_ne_u32) SHOW(bfe_u32)
        SHOW(bfi_b32) SHOW(lshl_or) SHOW(or3) SHOW(add_lshl) SHOW(alignbit) SHOW(cvt_f32_u32) SHOW(add_f32_dpp) SHOW(mov_imm) SHOW(fma_sgpr) SHOW(max_sgpr)
        SHOW(mul_legacy) SHOW(ldexp_f32) SHOW(pk_fma_f32) SHOW(pk_mul_f32) SHOW(pk_add_f32) SHOW(pk_mov_b32) SHOW(mov_b64) SHOW(lshl_add_u64)
    }
    return 0;
}
