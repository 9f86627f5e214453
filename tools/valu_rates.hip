// tools/valu_rates.hip -- issue cost (shader cycles per wave-instruction) of the vector instructions the
// sampling kernels are made of, measured on the GPU box: one workgroup on one CU, 1 or 4 wavefronts per
// SIMD, 32 independent instructions per loop iteration, s_memtime around the loop.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
#define ITER 2000

#define KERNEL(NAME, ASM, CONSTR)                                                          \
    __global__ void NAME(unsigned long long* out, double seed) {                           \
        double a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;             \
        double b = seed * 0.37 + 1.25, c = seed * 0.11 + 0.5;                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
        for (int i = 0; i < ITER; ++i) {                                                   \
            REP8(asm volatile(ASM : "+v"(a0) : CONSTR(b), CONSTR(c));                      \
                 asm volatile(ASM : "+v"(a1) : CONSTR(b), CONSTR(c));                      \
                 asm volatile(ASM : "+v"(a2) : CONSTR(b), CONSTR(c));                      \
                 asm volatile(ASM : "+v"(a3) : CONSTR(b), CONSTR(c));)                     \
        }                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                      \
        if (a0 + a1 + a2 + a3 == 12345.678) out[16383] = 1;                                  \
    }

// 64-bit operands live in "+v"(double) register pairs; %0 is the pair, 32-bit forms use its low half
KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %1, %2", "v")
KERNEL(k_mul_f64, "v_mul_f64 %0, %0, %1", "v")
KERNEL(k_add_f64, "v_add_f64 %0, %0, %1", "v")
KERNEL(k_rcp_f64, "v_rcp_f64 %0, %0", "v")
KERNEL(k_sqrt_f64, "v_sqrt_f64 %0, %0", "v")
KERNEL(k_ldexp_f64, "v_ldexp_f64 %0, %0, 3", "v")
KERNEL(k_lshl_b64, "v_lshlrev_b64 %0, 4, %0", "v")
KERNEL(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1", "v")

#define KERNEL32(NAME, ASM)                                                                \
    __global__ void NAME(unsigned long long* out, unsigned seed) {                         \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;           \
        unsigned b = seed * 77 + 5, c = seed * 13 + 1;                                     \
        asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[10:11], 0x3333" ::: "vcc", "s10", "s11");       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
        for (int i = 0; i < ITER; ++i) {                                                   \
            REP8(asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c));                            \
                 asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c));                            \
                 asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c));                            \
                 asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c));)                           \
        }                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                      \
        if (a0 + a1 + a2 + a3 == 12345678u) out[16383] = 1;                                  \
    }

KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cndmask_nd, "v_cndmask_b32 %0, %1, %2, vcc")
KERNEL32(k_cndmask_e64vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL32(k_cmp_cnd_pair, "v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_cnd_pair_s, "v_cmp_lt_u32_e64 s[10:11], %1, %0\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_cmp_cnd2_e32, "v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_cmp_cnd2_e64, "v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %0, %0, %2, vcc")
KERNEL32(k_cnd_e32_add, "v_cndmask_b32 %0, %0, %1, vcc\n v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_cmp_cnd4_e32, "v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_cndmask_d, "v_cndmask_b32_e64 %0, %2, %1, s[10:11]")
KERNEL32(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_cmp_u32_s, "v_cmp_lt_u32_e64 s[10:11], %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL32(k_cvt_f64_u32, "v_cvt_f32_u32 %0, %0")
KERNEL32(k_log_f32, "v_log_f32 %0, %0")
KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 4")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")


// ---- round 6: every vector mnemonic of the k_sample<2> pool loop, k_sample<0>'s read loop and k_gl2's loops (tools/isa_hist.py reads
// the classes from this program's output: profiles/r06_valu_rates.txt)
KERNEL32(k_mad_u24_lit, "v_mad_u32_u24 %0, %1, 5, %0")
KERNEL32(k_xor_lit, "v_xor_b32 %0, 0x80000000, %0")
KERNEL32(k_or, "v_or_b32 %0, %0, %1")
KERNEL32(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL32(k_mul_f32_lit, "v_mul_f32 %0, 0x3fdb98c8, %0")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL32(k_sub_f32_lit, "v_sub_f32 %0, 0x4f800000, %0")
KERNEL32(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
KERNEL32(k_fmac_f32_lit, "v_fmac_f32 %0, 0xbe826aa9, %1")
KERNEL32(k_fmamk_f32, "v_fmamk_f32 %0, %0, 0x2f800000, %1")
KERNEL32(k_fma_f32_abs_s, "v_fma_f32 %0, |%0|, s12, %1")
KERNEL32(k_fma_f32_neg, "v_fma_f32 %0, -%0, %1, s12")
KERNEL32(k_cmp_f32_s, "v_cmp_lt_f32_e64 s[10:11], s12, %0")
KERNEL32(k_cmp_f32_vcc, "v_cmp_nlt_f32 vcc, s12, %0")
KERNEL32(k_cmp_i32_s, "v_cmp_lt_i32_e64 s[10:11], %0, %1")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, 7, %0")
KERNEL32(k_and_lit, "v_and_b32 %0, 0xff0, %0")
KERNEL32(k_and_or, "v_and_or_b32 %0, %1, 3, %0")
KERNEL32(k_mov_lit, "v_mov_b32 %0, 0x200")
KERNEL32(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD")
KERNEL32(k_min_sdwa, "v_min_i32_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL32(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL32(k_subrev_u32, "v_subrev_u32 %0, %1, %0")
KERNEL32(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL32(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 5")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 2, %1")
KERNEL32(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 2")
KERNEL32(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_readlane, "v_readlane_b32 s12, %0, 3")
KERNEL32(k_readfirst, "v_readfirstlane_b32 s12, %0")
KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_cvt_f64_f32_lo, "v_exp_f32 %0, %0")
KERNEL32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL32(k_rcp_f32_abs, "v_rcp_f32_e64 %0, |%0|")
KERNEL32(k_ldexp_f32, "v_ldexp_f32 %0, %0, %1")
KERNEL32(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
KERNEL32(k_mul_hi_i32, "v_mul_hi_i32 %0, %0, %1")

// 64-bit forms on register pairs ("+v"(double) %0 is the pair)
KERNEL(k_max_f64, "v_max_f64 %0, %0, %1", "v")
KERNEL(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1", "v")
KERNEL(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1", "v")
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2", "v")
KERNEL(k_lshr_b64, "v_lshrrev_b64 %0, 4, %0", "v")
KERNEL(k_fract_f64, "v_fract_f64 %0, %0", "v")
KERNEL(k_floor_f64, "v_floor_f64 %0, %0", "v")


// mixed widths: %0 = a 64-bit pair, %1 = a 32-bit register
#define KERNELMIX(NAME, ASM)                                                               \
    __global__ void NAME(unsigned long long* out, double seed) {                           \
        double a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;             \
        float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
        for (int i = 0; i < ITER; ++i) {                                                   \
            REP8(asm volatile(ASM : "+v"(a0), "+v"(f0));                                   \
                 asm volatile(ASM : "+v"(a1), "+v"(f1));                                   \
                 asm volatile(ASM : "+v"(a2), "+v"(f2));                                   \
                 asm volatile(ASM : "+v"(a3), "+v"(f3));)                                  \
        }                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                      \
        if (a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 == 12345.678) out[16383] = 1;              \
    }
KERNELMIX(k_cvt_f32_f64, "v_cvt_f32_f64 %1, %0")
KERNELMIX(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
KERNELMIX(k_cvt_f64_u32_real, "v_cvt_f64_u32 %0, %1")
KERNELMIX(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
KERNELMIX(k_cvt_i32_f64, "v_cvt_i32_f64 %1, %0")

// v_mad_u64_u32 with a scalar multiplier, as the generator step issues it
__global__ void k_mad_u64_s(unsigned long long* out, unsigned seed) {
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    unsigned b = seed * 77 + 5, c = seed * 13 + 1;
    asm volatile("s_mov_b32 s12, 0xDEECE66D" ::: "s12");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        REP8(asm volatile("v_mad_u64_u32 %0, s[10:11], %1, s12, %0" : "+v"(a0) : "v"(b), "v"(c) : "s10", "s11");
             asm volatile("v_mad_u64_u32 %0, s[10:11], %1, s12, %0" : "+v"(a1) : "v"(b), "v"(c) : "s10", "s11");
             asm volatile("v_mad_u64_u32 %0, s[10:11], %1, s12, %0" : "+v"(a2) : "v"(b), "v"(c) : "s10", "s11");
             asm volatile("v_mad_u64_u32 %0, s[10:11], %1, s12, %0" : "+v"(a3) : "v"(b), "v"(c) : "s10", "s11");)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 == 12345678u) out[16383] = 1;
}


// ---- mixes: do the 2.6-cycle and the 4.3-cycle classes share one issue slot, or does a wavefront of one class issue beside a wavefront of the other?
KERNEL32(k_mix_mad24_fma, "v_mad_u32_u24 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_mix_cnd_mul, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_mul_f32 %0, %0, %1")
KERNEL32(k_mix_cnd_mul2, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2")
KERNEL32(k_mix_align_cvt_fma, "v_alignbit_b32 %0, %0, %1, 20\n v_cvt_f32_u32 %0, %0\n v_fma_f32 %0, %0, %1, %2")
// the generator step's three instructions (a 64-bit pair and a 32-bit register per chain)
__global__ void k_mix_lcg(unsigned long long* out, unsigned seed) {
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    unsigned b0 = seed * 77 + 5, b1 = b0 * 3, b2 = b0 * 5, b3 = b0 * 7, c = seed * 13 + 1;
#define LCG3 "v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u32_u24 %1, %1, %2, %1\n v_mad_u32_u24 %1, %1, 5, %1"
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        REP8(asm volatile(LCG3 : "+v"(a0), "+v"(b0) : "v"(c) : "s10", "s11");
             asm volatile(LCG3 : "+v"(a1), "+v"(b1) : "v"(c) : "s10", "s11");
             asm volatile(LCG3 : "+v"(a2), "+v"(b2) : "v"(c) : "s10", "s11");
             asm volatile(LCG3 : "+v"(a3), "+v"(b3) : "v"(c) : "s10", "s11");)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 == 12345678u) out[16383] = 1;
}
KERNEL32(k_mix_rcp_mul3, "v_rcp_f32 %0, %0\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %1")
KERNEL32(k_salu_mix, "v_mad_u32_u24 %0, %0, %1, %2\n s_and_b64 s[12:13], s[10:11], vcc\n s_or_b64 s[14:15], s[12:13], vcc")


// ---- independent mixes (round 6): chain A takes the first instruction, chain B the others -- no instruction depends on its neighbour, so what is
// measured is whether the 32-lane-per-clock class (v_mul_f32, v_add_u32 ...: 2.3 cycles alone) issues in the shadow of the 16-lane-per-clock
// class (v_mad_u32_u24, v_cndmask_b32, conversions, compares: 4.15 alone) or queues behind it
#define KERNELMIX2(NAME, ASM)                                                              \
    __global__ void NAME(unsigned long long* out, unsigned seed) {                         \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;           \
        unsigned b0 = a0 * 11, b1 = a0 * 13, b2 = a0 * 17, b3 = a0 * 19;                   \
        unsigned e0 = a0 * 23, e1 = a0 * 29, e2 = a0 * 31, e3 = a0 * 37;                   \
        unsigned c = seed * 77 + 5, d = seed * 13 + 1;                                     \
        asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[10:11], 0x3333" ::: "vcc", "s10", "s11");       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
        for (int i = 0; i < ITER; ++i) {                                                   \
            REP8(asm volatile(ASM : "+v"(a0), "+v"(b0), "+v"(e0) : "v"(c), "v"(d) : "s12", "s13", "v30", "v31");              \
                 asm volatile(ASM : "+v"(a1), "+v"(b1), "+v"(e1) : "v"(c), "v"(d) : "s12", "s13", "v30", "v31");              \
                 asm volatile(ASM : "+v"(a2), "+v"(b2), "+v"(e2) : "v"(c), "v"(d) : "s12", "s13", "v30", "v31");              \
                 asm volatile(ASM : "+v"(a3), "+v"(b3), "+v"(e3) : "v"(c), "v"(d) : "s12", "s13", "v30", "v31");)             \
        }                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;  \
        if (a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + e0 + e1 + e2 + e3 == 12345678u) out[16383] = 1; \
    }
#define SLOW "v_mad_u32_u24 %0, %0, %3, %4\n"
#define FASTB "v_mul_f32 %1, %1, %3\n"
#define FASTE "v_add_u32 %2, %2, %4\n"
KERNELMIX2(m_s1f1, SLOW FASTB)
KERNELMIX2(m_s1f2, SLOW FASTB FASTE)
KERNELMIX2(m_s1f3, SLOW FASTB FASTE FASTB)
KERNELMIX2(m_s1f4, SLOW FASTB FASTE FASTB FASTE)
KERNELMIX2(m_s2f1, SLOW "v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n" FASTB)
KERNELMIX2(m_s2f2, SLOW FASTB "v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n" FASTB)
KERNELMIX2(m_s3f1, SLOW "v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n" "v_alignbit_b32 %0, %0, %3, 20\n" FASTB)
KERNELMIX2(m_s1fma1, SLOW "v_fma_f32 %1, %1, %3, %4\n")
KERNELMIX2(m_cmp1f1, "v_cmp_lt_f32_e64 s[12:13], %0, %3\n" FASTB)
KERNELMIX2(m_cmp1s1, "v_cmp_lt_f32_e64 s[12:13], %0, %3\n" "v_mad_u32_u24 %1, %1, %3, %4\n")
KERNELMIX2(m_t1f1, "v_rcp_f32 %0, %0\n" FASTB)
KERNELMIX2(m_t1f3, "v_rcp_f32 %0, %0\n" FASTB FASTE FASTB)
KERNELMIX2(m_t1s1, "v_rcp_f32 %0, %0\n" "v_mad_u32_u24 %1, %1, %3, %4\n")
KERNELMIX2(m_t1s1f2, "v_rcp_f32 %0, %0\n" "v_mad_u32_u24 %1, %1, %3, %4\n" FASTE "v_mul_f32 %2, %2, %3\n")
KERNELMIX2(m_u64_f1, "v_mad_u64_u32 v[30:31], s[12:13], %0, %3, v[30:31]\n" FASTB)
KERNELMIX2(m_s1salu2, SLOW "s_and_b64 s[12:13], s[10:11], vcc\n s_or_b64 s[12:13], s[12:13], vcc\n")
KERNELMIX2(m_f1salu2, FASTB "s_and_b64 s[12:13], s[10:11], vcc\n s_or_b64 s[12:13], s[12:13], vcc\n")
KERNELMIX2(m_s1f1salu2, SLOW FASTB "s_and_b64 s[12:13], s[10:11], vcc\n s_or_b64 s[12:13], s[12:13], vcc\n")

// v_mad_u64_u32 vdst[2], sdst(carry), src0, src1, src2[2]
__global__ void k_mad_u64(unsigned long long* out, unsigned seed) {
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    unsigned b = seed * 77 + 5, c = seed * 13 + 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        REP8(asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c) : "s10", "s11");
             asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c) : "s10", "s11");
             asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c) : "s10", "s11");
             asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c) : "s10", "s11");)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 == 12345678u) out[16383] = 1;
}

static int g_ncu = 256;
template <typename K, typename S>
static void run8(const char* name, K kern, S seed, unsigned long long* d, double insts_per_asm) {
    // EIGHT wavefronts per SIMD (what k_sample<2> runs at): a workgroup has at most 1024 threads = four per SIMD, so two workgroups per CU on
    // every CU of the chip (grid = 2 x CUs; the dispatcher fills the CUs evenly); the slowest wavefront's time is taken
    const int blocks = 2 * g_ncu;
    static unsigned long long h[16384];
    kern<<<blocks, 1024>>>(d, seed);
    hipDeviceSynchronize();
    kern<<<blocks, 1024>>>(d, seed);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(unsigned long long) * blocks * 16, hipMemcpyDeviceToHost);
    unsigned long long mx = 0, mn = ~0ULL;
    for (int i = 0; i < blocks * 16; ++i) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
    printf("%-14s waves/SIMD 8: %.2f cyc per wave-instruction (SIMD time), %.2f per wave (chip-wide, fastest wavefront %.2f)\n", name,
           (double)mx / (32.0 * insts_per_asm * ITER * 8), (double)mx / (32.0 * insts_per_asm * ITER), (double)mn / (32.0 * insts_per_asm * ITER * 8));
}
template <typename K, typename S>
static void run(const char* name, K kern, S seed, unsigned long long* d, double insts_per_asm = 1.0) {
    unsigned long long h[16];
    run8(name, kern, seed, d, insts_per_asm);
    for (int wps : {1, 2, 4}) {
        const int waves = 4 * wps;
        kern<<<1, 64 * waves>>>(d, seed);
        hipDeviceSynchronize();
        kern<<<1, 64 * waves>>>(d, seed);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
        // cycles of SIMD time per wave-instruction: wall cycles / (instructions per wave * waves per SIMD)
        printf("%-14s waves/SIMD %d: %.2f cyc per wave-instruction (SIMD time), %.2f per wave\n", name, wps,
               (double)mx / (32.0 * insts_per_asm * ITER * wps), (double)mx / (32.0 * insts_per_asm * ITER));
    }
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * 16384);
    hipMemset(d, 0, 8 * 16384);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess && prop.multiProcessorCount > 0) g_ncu = prop.multiProcessorCount;
    printf("# %d CUs; cycles = s_memtime ticks; 'waves/SIMD 8' lines: two 1024-thread workgroups per CU on every CU\n", g_ncu);
#define R64(k) run(#k, k, 1.5, d)
#define R32(k) run(#k, k, 12345u, d)
    R64(k_fma_f64); R64(k_mul_f64); R64(k_add_f64); R64(k_rcp_f64); R64(k_sqrt_f64); R64(k_ldexp_f64);
    R64(k_lshl_b64); R64(k_cmp_f64);
    R32(k_mul_lo_u32); R32(k_mul_hi_u32); R32(k_mul_u24); R32(k_mul_hi_u24); R32(k_mad_u24); R32(k_mad_u64);
    R32(k_add_u32); R32(k_add3_u32); R32(k_lshl_add); R32(k_cndmask); R32(k_cndmask_nd); R32(k_cndmask_e64vcc); R32(k_cmp_cnd_pair); R32(k_cmp_cnd_pair_s); R32(k_cmp_cnd2_e32); R32(k_cmp_cnd2_e64); R32(k_cnd_e32_add); R32(k_cmp_cnd4_e32); R32(k_cndmask_s); R32(k_cndmask_d); R32(k_cmp_u32); R32(k_cmp_u32_s); R32(k_mov); R32(k_and); R32(k_lshl); R32(k_mbcnt); R32(k_cvt_f64_u32); R32(k_log_f32); R32(k_rcp_f32); R32(k_fma_f32);
    R32(k_alignbit); R32(k_perm);
    // round 6
    R32(k_mad_u64_s); R32(k_mad_u24_lit); R32(k_xor_lit); R32(k_or); R32(k_cvt_f32_i32); R32(k_cvt_i32_f32); R32(k_mul_f32); R32(k_mul_f32_lit); R32(k_add_f32);
    R32(k_sub_f32_lit); R32(k_fmac_f32); R32(k_fmac_f32_lit); R32(k_fmamk_f32); R32(k_fma_f32_abs_s); R32(k_fma_f32_neg); R32(k_cmp_f32_s); R32(k_cmp_f32_vcc);
    R32(k_cmp_i32_s); R32(k_lshr); R32(k_and_lit); R32(k_and_or); R32(k_mov_lit); R32(k_add_sdwa); R32(k_min_sdwa); R32(k_sub_u32); R32(k_subrev_u32);
    R32(k_max_f32); R32(k_min_u32); R32(k_bfe_u32); R32(k_lshl_or); R32(k_add_lshl); R32(k_bcnt); R32(k_addc); R32(k_add_co); R32(k_readlane); R32(k_readfirst);
    R32(k_mov_dpp); R32(k_add_dpp); R32(k_cvt_f64_f32_lo); R32(k_sqrt_f32); R32(k_rcp_f32_abs); R32(k_ldexp_f32); R32(k_cvt_u32_f32); R32(k_mul_hi_i32);
    R64(k_cvt_f32_f64); R64(k_cvt_f64_f32); R64(k_max_f64); R64(k_pk_mul_f32); R64(k_pk_add_f32); R64(k_pk_fma_f32); R64(k_lshr_b64); R64(k_cvt_f64_u32_real);
    R64(k_cvt_f64_i32); R64(k_fract_f64); R64(k_floor_f64); R64(k_cvt_i32_f64);
    // mixes (cycles per INSTRUCTION of the mix)
    run("k_mix_mad24_fma", k_mix_mad24_fma, 12345u, d, 2.0); run("k_mix_cnd_mul", k_mix_cnd_mul, 12345u, d, 2.0); run("k_mix_cnd_mul2", k_mix_cnd_mul2, 12345u, d, 3.0);
    run("k_mix_align_cvt_fma", k_mix_align_cvt_fma, 12345u, d, 3.0); run("k_mix_lcg", k_mix_lcg, 12345u, d, 3.0); run("k_mix_rcp_mul3", k_mix_rcp_mul3, 12345u, d, 4.0);
    run("k_salu_mix", k_salu_mix, 12345u, d, 1.0);
    // independent mixes: cycles per asm GROUP (the whole mix once)
#define RM(k) run(#k, k, 12345u, d, 1.0)
    RM(m_s1f1); RM(m_s1f2); RM(m_s1f3); RM(m_s1f4); RM(m_s2f1); RM(m_s2f2); RM(m_s3f1); RM(m_s1fma1); RM(m_cmp1f1); RM(m_cmp1s1); RM(m_t1f1); RM(m_t1f3); RM(m_t1s1);
    RM(m_t1s1f2); RM(m_u64_f1); RM(m_s1salu2); RM(m_f1salu2); RM(m_s1f1salu2);
    return 0;
}
