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
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                      \
        if (a0 + a1 + a2 + a3 == 12345.678) out[100] = 1;                                  \
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
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                      \
        if (a0 + a1 + a2 + a3 == 12345678u) out[100] = 1;                                  \
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
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (a0 + a1 + a2 + a3 == 12345678u) out[100] = 1;
}

template <typename K, typename S>
static void run(const char* name, K kern, S seed, unsigned long long* d) {
    unsigned long long h[16];
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
               (double)mx / (32.0 * ITER * wps), (double)mx / (32.0 * ITER));
    }
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * 256);
    hipMemset(d, 0, 8 * 256);
#define R64(k) run(#k, k, 1.5, d)
#define R32(k) run(#k, k, 12345u, d)
    R64(k_fma_f64); R64(k_mul_f64); R64(k_add_f64); R64(k_rcp_f64); R64(k_sqrt_f64); R64(k_ldexp_f64);
    R64(k_lshl_b64); R64(k_cmp_f64);
    R32(k_mul_lo_u32); R32(k_mul_hi_u32); R32(k_mul_u24); R32(k_mul_hi_u24); R32(k_mad_u24); R32(k_mad_u64);
    R32(k_add_u32); R32(k_add3_u32); R32(k_lshl_add); R32(k_cndmask); R32(k_cndmask_nd); R32(k_cndmask_e64vcc); R32(k_cmp_cnd_pair); R32(k_cmp_cnd_pair_s); R32(k_cmp_cnd2_e32); R32(k_cmp_cnd2_e64); R32(k_cnd_e32_add); R32(k_cmp_cnd4_e32); R32(k_cndmask_s); R32(k_cndmask_d); R32(k_cmp_u32); R32(k_cmp_u32_s); R32(k_mov); R32(k_and); R32(k_lshl); R32(k_mbcnt); R32(k_cvt_f64_u32); R32(k_log_f32); R32(k_rcp_f32); R32(k_fma_f32);
    R32(k_alignbit); R32(k_perm);
    return 0;
}
