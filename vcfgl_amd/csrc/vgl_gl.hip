// vgl_gl.hip -- per-site allele order (k_site), genotype likelihoods and their PL / GP / AD
// epilogue (k_gl<A>), order-dependent per-site float sums (k_siteagg).
// (vcfgl.cpp:396-404, 665-970, 982-1074; gl_methods.cpp:4-369)
#include <type_traits>

#include "vgl_common.hip.h"

// ------------------------------------------------------------------------------------
// status + allele order of one site from its summed depths (vcfgl.cpp:396-404, 665-766; no-reads :228-315)
// Registers only (no indexed local arrays: in the fused build of k_gl a whole workgroup waits for this): the four bases are ordered by a
// sorting network on keys (depth << 2 | 3 - base) -- descending depth, ties in the order A < C < G < T, as the reference's stable
// insertion sort leaves them -- and the two allele tables are built as words of 4-bit entries (0xF = none), the form VglSiteInfo keeps
struct VglSiteOrder { int status, nAll, nObs; uint32_t a2b, b2a; };      // a2b: allele -> base (4 = unobserved), b2a: base -> allele; entries 0..4
__device__ __forceinline__ int site_nib(const uint32_t w, const int i) { const int v = (int)((w >> (4 * i)) & 0xFu); return v == 0xF ? -1 : v; }
// status and allele count alone (the part of site_order() that needs no ordering): what the likelihood loops of the fused kernel look at,
// worked out by every thread while the allele order itself is off the workgroup's critical path
__device__ __forceinline__ void site_status_nall(const VglDevParams& P, const int info_dp, const int ad[4], int& status, int& nAll) {
    const bool add_unobs = (P.A == 5);
    status = SITE_OK; nAll = 0;
    if (0 == info_dp) {
        if (P.rm_empty_sites) status = SITE_SKIP_EMPTY;
        else {
            status = SITE_NO_READS;
            nAll = (P.do_unobserved <= 2) ? 1 : (P.do_unobserved == 3 ? 4 : 5);
        }
    } else {
        const int nObservedBases = (ad[0] > 0) + (ad[1] > 0) + (ad[2] > 0) + (ad[3] > 0);
        if ((P.rm_invar_sites & 4) && 1 == nObservedBases) status = SITE_SKIP_INVAR;
        else nAll = (P.do_unobserved >= 3 ? 4 : nObservedBases) + (add_unobs ? 1 : 0);
    }
}
__device__ __forceinline__ VglSiteOrder site_order(const VglDevParams& P, const int info_dp, const int ad[4]) {
    VglSiteOrder o;
    const bool add_unobs = (P.A == 5);
    uint32_t a2b = 0xFFFFFu, b2a = 0xFFFFFu;
    int status = SITE_OK, nAll = 0, nObs = 0;
    auto set = [](uint32_t w, const int i, const int v) { return (w & ~(0xFu << (4 * i))) | ((uint32_t)v << (4 * i)); };
    if (0 == info_dp) {
        if (P.rm_empty_sites) status = SITE_SKIP_EMPTY;
        else {
            status = SITE_NO_READS;
            if (P.do_unobserved == 0) { nAll = 1; nObs = 0; }
            else if (P.do_unobserved <= 2) { nAll = 1; nObs = 0; a2b = 0xFFFF4u; }
            else if (P.do_unobserved == 3) { nAll = 4; nObs = 4; a2b = 0xF3210u; }
            else { nAll = 5; nObs = 4; a2b = 0x43210u; }
        }
    } else {
        const int nObservedBases = (ad[0] > 0) + (ad[1] > 0) + (ad[2] > 0) + (ad[3] > 0);
        if ((P.rm_invar_sites & 4) && 1 == nObservedBases) status = SITE_SKIP_INVAR;
        else {
            uint64_t k0 = ((uint64_t)(uint32_t)ad[0] << 2) | 3u, k1 = ((uint64_t)(uint32_t)ad[1] << 2) | 2u, k2 = ((uint64_t)(uint32_t)ad[2] << 2) | 1u, k3 = ((uint64_t)(uint32_t)ad[3] << 2);
            auto cx = [](uint64_t& hi, uint64_t& lo) { const uint64_t a = hi > lo ? hi : lo, b = hi > lo ? lo : hi; hi = a; lo = b; };
            cx(k0, k1); cx(k2, k3); cx(k0, k2); cx(k1, k3); cx(k1, k2);                           // descending
            const bool explode = P.do_unobserved >= 3;
            const int n_all = explode ? 4 : nObservedBases;                                       // the bases listed: all four, or those with reads (they sort first)
            const uint64_t ks[4] = {k0, k1, k2, k3};
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < n_all) { const int b = 3 - (int)(ks[a] & 3u); a2b = set(a2b, a, b); b2a = set(b2a, b, a); }
            if (add_unobs) { a2b = set(a2b, n_all, 4); b2a = set(b2a, 4, n_all); }
            nObs = n_all; nAll = n_all + (add_unobs ? 1 : 0);
        }
    }
    o.status = status; o.nAll = nAll; o.nObs = nObs; o.a2b = a2b; o.b2a = b2a;
    return o;
}
// k_gl (GL model 2) leaves an evaluation's accumulators in rows indexed by the RANKS of its present bases (0..3, 4 = an allele it
// has no read of): row(ri, rj) = max(tri) + min(rank), tri = rank (rank + 1) / 2.  Which row holds genotype (i, j) of the site's
// alleles depends only on the site's allele order and on which of the four bases the evaluation shows -- 16 cases, tabulated
// once per site instead of being worked out by every (site, sample) thread: entry [present mask] = 15 x 4 bits in bcf_alleles2gt order
template <int A>
__device__ __forceinline__ uint64_t site_rowmap_entry_t(const uint32_t a2b, const uint32_t pm) {
    uint64_t m = 0;
    int pr[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const uint32_t bb = (a2b >> (4 * i)) & 0xFu;
        pr[i] = (bb < 4u && ((pm >> bb) & 1u)) ? __popc(pm & ((1u << bb) - 1u)) : 4;
    }
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const int ti = pr[i] * (pr[i] + 1) / 2, tj = pr[j] * (pr[j] + 1) / 2;
            const int row = (ti > tj ? ti : tj) + (pr[i] < pr[j] ? pr[i] : pr[j]);
            m |= (uint64_t)row << (4 * (i * (i + 1) / 2 + j));
        }
    return m;
}
__device__ __forceinline__ uint64_t site_rowmap_entry(const int A, const uint32_t a2b, const uint32_t pm) {
    return A == 5 ? site_rowmap_entry_t<5>(a2b, pm) : site_rowmap_entry_t<4>(a2b, pm);
}
__device__ __forceinline__ VglSiteInfo site_info_of(const VglSiteOrder& o) {
    VglSiteInfo si;
    si.status = o.status; si.n_alleles = o.nAll;
    si.acgt2alleles = o.b2a; si.alleles2acgt = o.a2b;
    return si;
}
// the per-site outputs (one thread per site): `acc` = the site's sums, [0] INFO/DP, [1..4] per-base depth, [5..8] forward-strand depth
__device__ __forceinline__ void site_outputs(const VglDevParams& P, const VglTilePtrs& T, const int ls, const VglSiteOrder& o, const int32_t* acc) {
    const int A = P.A;
    T.site_status[ls] = o.status;
    T.n_alleles[ls] = o.nAll;
    if (T.n_alleles_obs) T.n_alleles_obs[ls] = o.nObs;
#pragma unroll
    for (int k = 0; k < 5; k++) T.alleles2acgt[(size_t)ls * 5 + k] = (int8_t)site_nib(o.a2b, k);
    if (T.info_dp) T.info_dp[ls] = acc[0];
    const bool have = (o.status == SITE_OK);
    if (T.info_ad || T.info_adf || T.info_adr)
        for (int a = 0; a < A; a++) {
            const int b = (have && a < o.nAll) ? site_nib(o.a2b, a) : -1;
            const bool real = (b >= 0 && b < 4);
            int tot = 0, totf = 0;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) if (real && b == bb) { tot = acc[1 + bb]; totf = acc[5 + bb]; }
            if (T.info_ad) T.info_ad[(size_t)ls * A + a] = tot;
            if (T.info_adf) T.info_adf[(size_t)ls * A + a] = totf;
            if (T.info_adr) T.info_adr[(size_t)ls * A + a] = tot - totf;
        }
}

// k_gl2's compact accumulator array (below): the rows an evaluation leaves depend on how many bases its reads show -- 3 / 6 / 10 / 15 for
// 1 .. 4 -- so the rows are kept in THAT order (physical row: 0, 10, 14 | 1, 2, 11 | 3, 4, 5, 12 | 6, 7, 8, 9, 13 of k_gl's numbering):
// the first six, which all but a few per cent of the evaluations stop at, as full rows of the workgroup's 1024 columns (4096 bytes each),
// the other nine as rows of VGL_GL2_OVC columns (1024 bytes each) for the evaluations at the head of the sorted order, which are the ones
// with three or four bases.  Code of a row = its offset in units of 512 bytes: 8 phys for the first six, 36 + 2 phys for the rest.
#define VGL_GL2_OVC 256
__device__ __forceinline__ constexpr int gl2_row_phys(const int row) {
    constexpr int ph[15] = {0, 3, 4, 6, 7, 8, 10, 11, 12, 13, 1, 5, 9, 14, 2};
    return ph[row];
}
__device__ __forceinline__ constexpr int gl2_row_code(const int row) { return gl2_row_phys(row) < 6 ? 8 * gl2_row_phys(row) : 36 + 2 * gl2_row_phys(row); }
static_assert(VGL_GL2_OVC * 4 == 2 * 512 && 36 + 2 * 14 < 256, "row codes: full rows of 1024 columns = 8 units, pool rows of VGL_GL2_OVC columns = 2 units of 512 bytes");

// one lane per site.  With the row table of k_gl (T.rowmap, GL model 2) SIXTEEN lanes per site: all work out the site (a few hundred
// instructions), lane `pm` writes table entry `pm`, lane 0 everything else
__global__ __launch_bounds__(256) void k_site(const VglDevParams P, const VglTilePtrs T) {
    const size_t gt_ = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ls = (int)(T.rowmap ? (gt_ >> 4) : gt_);                 // (a tile's sites fit an int: vgl_simulate_tile checks)
    const uint32_t pm = T.rowmap ? (uint32_t)(gt_ & 15) : 0u;
    if (ls >= T.n_sites) return;
    const int32_t* acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
    const int ad[4] = {acc[1], acc[2], acc[3], acc[4]};
    const VglSiteOrder o = site_order(P, acc[0], ad);
    if (T.rowmap) {
        const uint64_t e = site_rowmap_entry(P.A, o.a2b, pm);
        T.rowmap[(size_t)ls * 16 + pm] = e;
        if (T.rowmap8) {                                               // k_gl2: the same entry as LDS offsets of its compact accumulator array, one byte per genotype
            uint64_t lo = 0, hi = 0;
#pragma unroll
            for (int g = 0; g < 15; ++g) {
                const uint64_t c = (uint64_t)gl2_row_code((int)((e >> (4 * g)) & 15u));
                if (g < 8) lo |= c << (8 * g); else hi |= c << (8 * (g - 8));
            }
            T.rowmap8[((size_t)ls * 16 + pm) * 2] = lo; T.rowmap8[((size_t)ls * 16 + pm) * 2 + 1] = hi;
        }
        if (pm != 0) return;
    }
    T.sinfo[ls] = site_info_of(o);
    site_outputs(P, T, ls, o, acc);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) uint32_t vgl_gu32;             // a global-memory word (an address the compiler cannot trace keeps its address space)

// ------------------------------------------------------------------------------------
__device__ __forceinline__ float f32_missing() { return __uint_as_float(F32_MISSING_BITS); }
__device__ __forceinline__ int nib(uint32_t v, int k) { return (int)((v >> (4 * k)) & 0xF); }
// PL of one likelihood (vcfgl.cpp:907-939)
// lroundf(-10.0 * gl) capped at MAXPL, MAXPL for gl = -inf.  The double product of a float and -10 is exact, so its conversion to
// float is the float product x; gl <= 0 (the maximum has been subtracted), so x >= 0 and lroundf(x) = floor(x + 0.5).  In float32
// trunc(x + 0.5f) equals that for EVERY x in [0, 255] except x = 0.5 - 2^-25 (the sum rounds up to 1), which one compare puts right
// (checked over all 1.1e9 float32 values of the range; x = -10 gl does reach that value, for gl = -0x1.999998p-5); capping x at 255
// first takes x = +inf (gl = -inf) and keeps the conversion in range.  Branch-free, seven instructions.
__device__ __forceinline__ uint32_t pl_of(const float v, const bool valid) {
    const float x = __builtin_fminf(v * -10.0f, (float)MAXPL);
    uint32_t r = (uint32_t)(int32_t)(x + 0.5f);
    r -= (x == 0x1.fffffep-2f) ? 1u : 0u;
    return valid ? r : (uint32_t)I32_MISSING;
}
// log10(x) for 0 < x <= 1 (--precise-gl 1: three per read, gl_methods.cpp:171-220), relative error < 1e-15 -- the mode is held to 1e-6,
// like the device log10() it replaces: x = m 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m - 1) / (m + 1)) by its series up to
// s^19 (|s| <= 0.172: remainder 2e-17 relative), log10 x = (e ln 2 + ln m) log10(e).  x <= 0 gives -inf (log10(0) of a certain base).
__device__ __forceinline__ double log10_unit(const double x) {
    if (!(x > 0.0)) return -INFINITY;
    int e;
    double m = frexp(x, &e);                                         // [1/2, 1)
    const bool lo = m < 0.70710678118654752;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double s = div_inrange(m - 1.0, m + 1.0);                  // the correctly rounded quotient without the exponent handling of the IEEE sequence (|m - 1| < 0.42, m + 1 in [1.7, 2.42))
    const double z = s * s;
    double p = 1.0 / 19.0;
    p = __builtin_fma(p, z, 1.0 / 17.0); p = __builtin_fma(p, z, 1.0 / 15.0); p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0); p = __builtin_fma(p, z, 1.0 / 9.0); p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0); p = __builtin_fma(p, z, 1.0 / 3.0); p = __builtin_fma(p, z, 1.0);
    const double ln = __builtin_fma((double)e, 0.69314718055994531, (s + s) * p);
    return ln * 0.43429448190325183;
}
// 10^x for x <= 0 (GP = 10^GL, vcfgl.cpp:941-970), relative error 2e-16 (numpy check of the same steps against 10**x) -- GP is held to 1e-6, like the device pow()
// it replaces (≈200 vector instructions per call, fifteen calls per evaluation): 2^n 2^f with n = rint(x log2 10), f in [-1/2, 1/2]
// taken as x log2 10 - n in two pieces, 2^f = e^(f ln 2) by a degree-13 Taylor polynomial (|f ln 2| <= 0.35: remainder < 2e-17).
__device__ __forceinline__ double exp10_nonpos(const double x) {
    if (!(x > -330.0)) return 0.0;                                // underflow of the float result, -inf
    const double L2_10_hi = 3.3219280948873622, L2_10_lo = 1.6616175169735921e-16;     // log2(10) = hi + lo
    const double y = x * L2_10_hi;
    const double n = rint(y);
    const double f = __builtin_fma(x, L2_10_hi, -n) + x * L2_10_lo;
    const double t = f * 0.69314718055994531;
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, t, 1.0 / 479001600.0); p = __builtin_fma(p, t, 1.0 / 39916800.0); p = __builtin_fma(p, t, 1.0 / 3628800.0);
    p = __builtin_fma(p, t, 1.0 / 362880.0); p = __builtin_fma(p, t, 1.0 / 40320.0); p = __builtin_fma(p, t, 1.0 / 5040.0);
    p = __builtin_fma(p, t, 1.0 / 720.0); p = __builtin_fma(p, t, 1.0 / 120.0); p = __builtin_fma(p, t, 1.0 / 24.0);
    p = __builtin_fma(p, t, 1.0 / 6.0); p = __builtin_fma(p, t, 0.5); p = __builtin_fma(p, t, 1.0); p = __builtin_fma(p, t, 1.0);
    return ldexp(p, (int)n);
}
__device__ __forceinline__ int cnt_of(uint64_t ad4, int b) { return b < 4 ? (int)((ad4 >> (16 * b)) & 0xFFFF) : 0; }

// Evaluations differ in depth -- and, for GL model 2, in how many distinct bases their reads show, which decides how many
// accumulators their loop carries (below) -- and the likelihood loop runs once per read: a wavefront is busy for its deepest,
// most diverse evaluation.  The 64 x WPB evaluations of a workgroup are therefore re-dealt to the lanes in (distinct bases, depth)
// order (P.gl_sort != 0: LDS counting sort of the thread ids), so that each wavefront works on similar evaluations.  The sorted
// lanes leave their accumulators in LDS, column = the evaluation's natural position; after one barrier every thread computes and
// stores all tags of the evaluation at its OWN natural position, so every store of a wavefront is one contiguous segment of a
// plane.  gl_sort 0: natural order throughout (the same path with the identity permutation).
// Which wavefront of a workgroup takes the heaviest 64 evaluations rotates with the workgroup index.
// GLM, PREC: the GL model (1 / 2) and --precise-gl as template parameters -- the paths share no code, and the
// model-1 tables and the double log10 of --precise-gl 1 would cost the plain model-2 loop registers (occupancy)
// FUSED (GL model 2, one fixed quality score, the default tag surface, every mean depth below 12, 128 < N <= 512: 256 or 512 threads): the workgroup IS one
// site, and what k_sample, k_site and the loads of this kernel would pass through HBM stays in the workgroup -- each thread first draws
// its own evaluation's depth and reads (k_sample's fixed-score path), the per-base depths are summed over the site in LDS, sixteen
// threads work out the allele order and the row table (k_site's code), and the likelihood pass below takes per-base depths, site
// record, row table and the staged reads (two bits each: the score is the same for all) from LDS.  One launch per tile instead of three.
// FUSEDW (round 4): 0 = not fused; else the staged words per evaluation of the fused build, 4 (at most 64 reads) or 8 (at most 128: depth 20
// has a staging capacity of 72).  A site with more samples than the workgroup has threads is SPLIT over P.fused_split consecutive
// workgroups; each publishes its per-base depth sums in a flagged slot of the site (T.fslot) and waits, bounded, for the others' (they
// are neighbours in dispatch order); a workgroup that runs out of patience computes the missing sums itself -- nothing depends on two
// workgroups being resident at the same time.  Mean depths of 12 and more take the depth draws from k_depth (T.dp_pre).
template <bool STORE, class Emit>
__device__ __forceinline__ uint64_t fused_sample_eval(const VglDevParams& P, const VglTilePtrs& T, const int site, const int samp, const int N, int& dps_out, Emit&& emit) {
    // k_sample<0, LEAN> for the evaluation (site, samp): vcfgl.cpp:364-389, 469-613 (vgl_sample.hip)
    const size_t ev = (size_t)site * N + (size_t)samp;
    const uint32_t g = T.gt[ev];                                        // (first: the one load of this phase that misses every cache, in flight during the depth loop)
    const uint64_t xe = aff(P.samp_tab[samp], T.site_base[site]);
    uint64_t st_hap16 = aff(P.off[1], xe) << 16, st_base16 = aff(P.off[2], xe) << 16;      // sample_read_base16: states carried shifted by 16
    int em;
    if (P.depth_pre == 1) em = T.dp_pre[ev];                            // the rejection method's draws: k_depth ahead of this kernel
    else {
        const uint64_t st_depth = aff(P.off[0], xe);
        VglPois pc = P.pois0;
        if (P.per_sample_depth) pc = P.pois[samp];
        double t = 1.0;                                                 // the product method (rng.h:289-299), state carried shifted by 4
        em = -1;
        uint64_t s52 = st_depth << 4;
        uint32_t k3ff = 0x3FF00000u;
        asm volatile("" : "+v"(k3ff));
        do { ++em; s52 = lcg_next52r(s52); t *= bits_1xxx_52r(s52, k3ff) - 1.0; } while (t > pc.g);
    }
    const int a0 = (int)(g & 0xF), a1 = (int)((g >> 4) & 0xF);
    int dps = (a0 == 0xF || a1 == 0xF) ? 0 : em;
    if (dps > P.read_cap) { if (STORE) atomicOr(T.errflag, VGL_DEVERR_CAPACITY); dps = P.read_cap; }
    const uint64_t err_thresh16 = sample_thresh16(P.err_thresh);
    dps_out = dps;                                                      // (before the reads: emit looks at it)
    uint64_t ad4s;
    if (__ballot(dps > 0 && a0 != a1) == 0) ad4s = sample_reads_fixed<true>(st_hap16, st_base16, a0, a1, dps, err_thresh16, emit);
    else ad4s = sample_reads_fixed<false>(st_hap16, st_base16, a0, a1, dps, err_thresh16, emit);
    if (STORE && T.fmt_dp) T.fmt_dp[ev] = dps;
    return ad4s;
}
// polls (two system-scope loads + s_sleep(4): about a microsecond each) before a split workgroup stops waiting for a neighbour and samples
// that part's depths itself.  Neighbours are consecutive workgroups and normally arrive within microseconds; when one is not resident
// (several contexts sharing the GPU) waiting longer than the recomputation costs only holds a CU slot -- so the bound is about that cost
#define VGL_FUSED_SPIN_LIMIT 2048

// (the kernel's body: hw_block = the hardware workgroup index of a grid of hw_grid, or -- hw_grid 0 -- the logical index itself: k_gl_redo)
// The fused build's split sites (more than 512 samples: up to four consecutive workgroups per site): the exchange of the parts' per-base depth sums.
// Its own device function since round 6: only the GENERAL fused build (FUSEDW_ + 16) calls it -- the plain build the depth-5
// configuration runs does not carry it (round 5: its mere presence cost that kernel 51 scalar spills).  f_acc[1..4]: this part's sums in, the site's out.
template <int WG>
__device__ __forceinline__ void fused_split_exchange(const VglDevParams& P, const VglTilePtrs& T, const int site, const int S, const int part, const int tid, const int N, int32_t* const f_acc) {
    // ---- the other workgroups' shares.  Every part publishes its four sums as TWO flagged 8-byte words (system-scope stores: A | C << 32
    // and G | T << 32, bit 63 = valid; the slots are zero at the start of the tile) and lane q of the first wavefront polls part q's
    // words with system-scope loads -- one store and, when the neighbour is already there, one load round trip (the first version's
    // four returning atomics + counter + polls + four reads by one lane cost about 8 us per workgroup with seven wavefronts waiting)
    __syncthreads();
    unsigned long long* const slots = T.fslot + (size_t)site * (size_t)S * 2;
    __shared__ int s_alone;
    if (tid == 0) {
        s_alone = 0;
        const unsigned long long V = 1ULL << 63;
        __hip_atomic_store(&slots[part * 2], (unsigned long long)(uint32_t)f_acc[1] | ((unsigned long long)(uint32_t)f_acc[2] << 32) | V, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&slots[part * 2 + 1], (unsigned long long)(uint32_t)f_acc[3] | ((unsigned long long)(uint32_t)f_acc[4] << 32) | V, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();                                            // (s_alone cleared; f_acc still holds this part's own sums)
    if (tid < S && tid != part) {
        const unsigned long long V = 1ULL << 63;
        unsigned long long w0 = 0, w1 = 0;
        const bool absent = (P.dbg_fuse_alone >> tid) & 1;     // test hook: part `tid` is treated as one that never arrives
        for (int spin = 0; spin < VGL_FUSED_SPIN_LIMIT && !absent; ++spin) {
            w0 = __hip_atomic_load(&slots[tid * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            w1 = __hip_atomic_load(&slots[tid * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((w0 & w1 & V) != 0) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if ((w0 & w1 & V) != 0 && !absent) {
            atomicAdd(&f_acc[1], (int)(uint32_t)w0); atomicAdd(&f_acc[2], (int)(uint32_t)((w0 & ~V) >> 32));
            atomicAdd(&f_acc[3], (int)(uint32_t)w1); atomicAdd(&f_acc[4], (int)(uint32_t)((w1 & ~V) >> 32));
        } else atomicOr(&s_alone, 1 << tid);
    }
    __syncthreads();
    const int alone_m = s_alone;
    if (alone_m) {
        // those neighbours did not show up in time (nothing promises that they run beside this workgroup): their evaluations' depths are
        // sampled here as well, counts only -- same streams, same sums
        for (int p2 = 0; p2 < S; ++p2) {
            if (!((alone_m >> p2) & 1)) continue;
            const int s2 = p2 * WG + tid;
            uint64_t ad2 = 0;
            if (s2 < N) { int d2; ad2 = fused_sample_eval<false>(P, T, site, s2, N, d2, [](const int, const uint32_t) {}); }
            int v2[4];
            wave_sum_ad4(ad2, v2);
            if ((tid & 63) == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (v2[k]) atomicAdd(&f_acc[1 + k], v2[k]);
            }
        }
    }
}

template <int A, int GLM, bool PREC, int WPB, int FUSEDW_ = 0>
__device__ __forceinline__ void k_gl_body(const VglDevParams& P, const VglTilePtrs& T, const uint32_t hw_block, const uint32_t hw_grid) {
    // FUSEDW_: the staged words of the fused build (4 / 8), + 16 for its GENERAL build.  The plain fused build is the one the depth-5 configuration
    // runs -- one workgroup per site, planes layout, GL / PL / DP only -- and carries nothing else: the exchange of a split site's depth sums with
    // its polling loop and the fallback that samples an absent neighbour's depths, the sample-major stores and the GP / FORMAT-AD epilogue are
    // several hundred instructions such a tile never runs, and their mere presence cost the kernel its scalar registers (49 scalar spills) and
    // C5 11 % (round 5, second session: 1.17-1.19 -> 1.04-1.06 ms, same box; vgl_launch_fused picks the build per tile)
    constexpr int FUSEDW = FUSEDW_ & 15;
    constexpr bool GEN = (FUSEDW_ & 16) != 0;
    constexpr bool SPLIT = GEN, SMB = GEN;
    constexpr bool FUSED = FUSEDW != 0;
    static_assert(!FUSED || (GLM == 2 && !PREC), "the fused build exists for GL model 2 with the score table");
    static_assert(FUSEDW == 0 || FUSEDW == 4 || FUSEDW == 8, "staged words per evaluation of the fused build");
    static_assert(WPB == 4 || WPB == 8, "workgroups of 256 or 512 threads");
    constexpr int WG = 64 * WPB;                                        // evaluations (threads) per workgroup: 256 or 512
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    constexpr int QL = 96;                                              // quality scores below QL take their three terms from LDS
    constexpr int NG = A * (A + 1) / 2;
    // one LDS object, so that its members lie in this order: the per-quality terms first -- at LDS offset 0 their three rows are
    // reached by the immediate offsets of ds_read2_b64 / ds_read_b64 (an add per read otherwise)
    struct Lds {
        double q2gl[(GLM == 2 && !PREC && !FUSED) ? 3 * QL : 1];
        uint32_t x[(GLM == 1 ? 16 : 15) * WG];                          // the accumulators (up to 15 rows) of the workgroup's evaluations on their way from sorted to natural order; before that,
                                                                        // GL model 1 with per-read scores: a (base, quality) histogram per lane, 64 one-byte rows per wavefront
        uint16_t perm[WG];
        int32_t ws[2 * WPB];                                            // (site, first sample) of the workgroup's wavefronts in natural order
        // FUSED: per-base depths of the site's evaluations, the site's row table, its summed depths and its record
        uint64_t f_ad4[FUSED ? WG : 1];
        uint64_t f_rowmap[FUSED ? 16 : 1];
        int32_t f_acc[FUSED ? 16 : 1];
        VglSiteInfo f_si;
    };
    __shared__ Lds s_lds;
    double* const s_q2gl = s_lds.q2gl;
    uint32_t* const s_x = s_lds.x;
    uint16_t* const s_perm = s_lds.perm;
    int32_t* const s_ws = s_lds.ws;
    uint32_t* const s_hist = s_x;                                       // [1026] bins of the depth sort (done before s_x is used)
    uint32_t* const s_fw = s_x + 1536;                                  // FUSED: [FUSEDW][WG] staged reads, sixteen per word (behind the sort's bins; free before the deposits)
    static_assert(!FUSED || 1536 + FUSEDW * WG <= 15 * WG, "the staged words of the fused build lie inside the accumulator array");
    const int N = P.n_samples;
    const int tid = threadIdx.x;
    if (GLM == 2 && !PREC && !FUSED) {
        if (P.error_qs == 2) {                                          // the per-quality terms of the read loop, in LDS
            for (int i = tid; i < 3 * QL; i += WG) s_q2gl[i] = P.q2gl[(i / QL) * 257 + (i % QL)];
            if (!P.gl_sort) __syncthreads();                            // (with the sort, its barriers stand between these stores and the loop)
        }
    }
    const int fsplit = SPLIT ? P.fused_split : 1;
    const uint32_t chunks_k = FUSED ? (uint32_t)(WPB * fsplit) : (uint32_t)P.chunks;   // FUSED: a site takes fused_split whole workgroups (wavefronts beyond its samples idle)
    const uint32_t nwaves = (uint32_t)T.n_sites * chunks_k;                             // < 2^31 (checked by the launcher)
    // logical workgroup: XCD-contiguous (xcd_block) -- except for the fused build with one workgroup per site, whose workgroups share
    // nothing: there the dispatch order measured 1 % faster (C5 k_gl 1.062 / 1.068 against 1.073 / 1.081 ms; the three-kernel k_gl loses
    // 4 % without the mapping)
    const uint32_t bx = (hw_grid != 0u && P.xcd_map && !(FUSED && fsplit == 1)) ? xcd_block(hw_block, hw_grid) : hw_block;
    uint64_t f_a = 0;                                                   // FUSED: this thread's per-base depths
    // wave k of this workgroup is wave 4 bx + k of the tile = (site, 64-sample chunk).  One scalar division per workgroup; a lane
    // then finds the site of ANY of the four waves with three compares (a per-lane 64-bit division was 60+ vector instructions,
    // three times per lane)
    const uint32_t wg_ls = (bx * (uint32_t)WPB) / chunks_k, wg_rem = (bx * (uint32_t)WPB) - wg_ls * chunks_k;
    auto wave_site = [&](const int k, int& ls_, int& s_base) {
        const int t = (int)wg_rem + k, c = (int)chunks_k;
        int add = 0;
#pragma unroll
        for (int j = 1; j < WPB; ++j) add += (t >= j * c) ? 1 : 0;
        ls_ = (int)wg_ls + add;
        s_base = (t - add * c) * 64;
    };
    // ---- depth of the evaluation this thread would own in natural order
    int dp0 = -1;                                                      // -1: no evaluation (padding lane)
    int k0 = 1;                                                        // GL model 2: distinct bases among its reads
    // In natural order a wavefront is 64 consecutive samples of ONE site: its site, first sample and site record are wave-uniform
    // and are kept in scalar registers (the compiler cannot see that tid >> 6 is uniform), so that the natural-order loads and the
    // epilogue's stores address memory as scalar base + lane offset and the allele table is decoded by scalar instructions
    int ls0 = 0, sb0 = 0, s0 = N;
    uint64_t a = 0;                                                    // its per-base depths (kept for the epilogue, which runs in natural order)
    uint64_t rmap = 0;                                                 // GL model 2: accumulator row of each genotype for this evaluation's set of bases (k_site's table)
    uint32_t rmask = 0;                                                // FUSED: which bases the evaluation shows (index of its row-table entry, fetched once the site's order is known)
    auto natural_setup = [&]() {
        const int wv_s = (GLM == 2) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
        const uint32_t w = bx * (uint32_t)WPB + (uint32_t)wv_s;
        if (w < nwaves) {
            wave_site(wv_s, ls0, sb0);
            if (GLM == 2) { ls0 = __builtin_amdgcn_readfirstlane(ls0); sb0 = __builtin_amdgcn_readfirstlane(sb0); }   // (measured: the GL model 1 kernel is 8 % slower with it)
            if (P.gl_sort && (tid & 63) == 0) { s_ws[2 * (tid >> 6)] = ls0; s_ws[2 * (tid >> 6) + 1] = sb0; }       // for the lane the sort hands an evaluation of this wavefront to
            s0 = sb0 + (tid & 63);
            if (s0 < N) {
                if constexpr (FUSED) a = f_a;
                else a = T.ad4[(size_t)ls0 * N + (size_t)sb0 + (size_t)(tid & 63)];
                dp0 = (int)((a & 0xFFFF) + ((a >> 16) & 0xFFFF) + ((a >> 32) & 0xFFFF) + ((a >> 48) & 0xFFFF));
                if (dp0 > 1023) dp0 = 1023;
                if (GLM == 2) {
                    const uint32_t q0 = (a & 0xFFFFULL) != 0, q1 = ((a >> 16) & 0xFFFF) != 0, q2 = ((a >> 32) & 0xFFFF) != 0, q3 = (a >> 48) != 0;
                    k0 = (int)(q0 + q1 + q2 + q3);
                    if constexpr (FUSED) rmask = q0 | (q1 << 1) | (q2 << 2) | (q3 << 3);
                    else rmap = T.rowmap[(size_t)ls0 * 16 + (q0 | (q1 << 1) | (q2 << 2) | (q3 << 3))];   // (in flight during the loop)
                }
            }
        }
    };
    // order of the sort: (GL model 2) most distinct bases first -- the loop a wavefront runs is the one of its lane with the most -- then
    // deepest first; evaluations without reads and padding lanes last.  At most 4 * 256 + 2 bins.
    constexpr int NK = (GLM == 2) ? 4 : 1;
    const int sort_sh = P.read_cap > 511 ? 2 : (P.read_cap > 255 ? 1 : 0);
    const int nbd = (P.read_cap >> sort_sh) + 1;                       // depth bins
    const int nb = NK * nbd + 2;
    auto sort_key = [&]() -> int {
        int key = nb - 1;
        if (dp0 == 0) key = nb - 2;
        else if (dp0 > 0) {
            const int dq = (dp0 > P.read_cap ? P.read_cap : dp0) >> sort_sh;
            // deepest first, except the two-base group: its shallow end shares a wavefront with the (few, expensive) evaluations of
            // three or four bases, whose loop then ends sooner; its deep end joins the cheap one-base loop's wavefronts
            key = (NK - k0) * nbd + ((GLM == 2 && k0 == 2 && P.gl_flip2) ? dq : nbd - 1 - dq);
        }
        return key;
    };
    auto sort_scan = [&](const int ln) {                               // exclusive scan of the bins by one wavefront (ln = its lane)
        uint32_t run = 0;
        for (int base = 0; base < nb; base += 64) {
            const int i = base + ln;
            const uint32_t v = (i < nb) ? s_hist[i] : 0u;
            const uint32_t incl = wave_incl_scan_u32(v);
            if (i < nb) s_hist[i] = run + incl - v;
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    };
    int key = 0;
    int f_status = SITE_OK, f_nall = 0;                                 // FUSED: the site's status and allele count (site_status_nall)
    if constexpr (FUSED) {
        const int S = fsplit;                                           // workgroups of this site
        const int site = (int)(bx / (uint32_t)S), part = (int)bx - site * S;
        const int samp = part * WG + tid;
        if (tid < 16) s_lds.f_acc[tid] = 0;
        // (round 4) the depth sort's histogram is filled in the same barrier interval as the site's sums, and its scan runs (second
        // wavefront) beside the sixteen threads of the allele order (first): six barriers per site instead of nine
        if (P.gl_sort) for (int i = tid; i < nb; i += WG) s_hist[i] = 0;
        __syncthreads();
        uint64_t ad4s = 0;
        if (samp < N) {
            uint32_t wcur = 0;
            int dps;
            int* const dps_p = &dps;
            auto emit = [&](const int trip, const uint32_t bases) {
                wcur |= bases << (8 * (trip & 3));
                if ((trip & 3) == 3 || 4 * trip + 4 >= *dps_p) { s_fw[(trip >> 2) * WG + tid] = wcur; wcur = 0; }
            };
            ad4s = fused_sample_eval<true>(P, T, site, samp, N, dps, emit);
        }
        f_a = ad4s;
        s_lds.f_ad4[tid] = ad4s;
        // ---- the site's summed depths (every read shows one base; without strand draws every read counts as forward)
        int v[5];
        wave_sum_ad4(ad4s, &v[1]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int k = 1; k < 5; ++k) if (v[k]) atomicAdd(&s_lds.f_acc[k], v[k]);
        }
        if (P.gl_sort) { natural_setup(); key = sort_key(); atomicAdd(&s_hist[key], 1u); }
        if constexpr (SPLIT) if (S > 1) fused_split_exchange<WG>(P, T, site, S, part, tid, N, s_lds.f_acc);
        __syncthreads();
        // ---- the site's sums are complete.  The likelihood loops need the site's status and allele COUNT only: every thread works them
        // out itself (a dozen instructions); the allele ORDER, the row table and the site's outputs (k_site's code, sixteen threads) are
        // needed by the epilogue only and are left to the wavefront with the lightest share of the sorted evaluations, after its loops
        // and ahead of the deposit barrier -- no wavefront waits for them (they held all eight for a tenth of the site's time)
        {
            const int ad_s[4] = {s_lds.f_acc[1], s_lds.f_acc[2], s_lds.f_acc[3], s_lds.f_acc[4]};
            site_status_nall(P, ad_s[0] + ad_s[1] + ad_s[2] + ad_s[3], ad_s, f_status, f_nall);
        }
        if (P.gl_sort && tid >= 64 && tid < 128) sort_scan(tid - 64);   // (bins complete behind the barrier above)
        __syncthreads();
        if (!P.gl_sort) natural_setup();
    } else {
        if (P.gl_sort) for (int i = tid; i < nb; i += WG) s_hist[i] = 0;   // (ahead of natural_setup(): the bins are cleared while its loads are in flight)
        natural_setup();
    }
    int otid = tid;
    if (P.gl_sort) {
        if constexpr (!FUSED) {
            __syncthreads();
            key = sort_key();
            atomicAdd(&s_hist[key], 1u);
            __syncthreads();
            if (tid < 64) sort_scan(tid);
            __syncthreads();
        }
        s_perm[atomicAdd(&s_hist[key], 1u)] = (uint16_t)tid;
        __syncthreads();
        // thread id whose evaluation this lane processes.  Which wavefront of the workgroup takes the deepest 64 evaluations
        // rotates with the workgroup index: wavefront k of every workgroup tends to land on the same SIMD of its CU, and
        // the deep groups must not all queue on one of them
        otid = s_perm[(tid + 64 * (int)(bx & (WPB - 1))) & (WG - 1)];
    }
    const int lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t w = bx * (uint32_t)WPB + (uint32_t)(otid >> 6);
    VglSiteInfo si_nv;                                                 // the natural evaluation's site (epilogue); in flight during the loop (FUSED: taken from LDS behind the deposit barrier)
    if constexpr (!FUSED) si_nv = T.sinfo[ls0];
    int ls = 0, s = N;
    if (w < nwaves) {
        int sb;
        if (P.gl_sort) { ls = s_ws[2 * (otid >> 6)]; sb = s_ws[2 * (otid >> 6) + 1]; }      // (written before the sort's barriers)
        else wave_site(otid >> 6, ls, sb);
        s = sb + (otid & 63);
    }
    const bool live = s < N;
    if (!live) { ls = 0; s = 0; }                                      // padding lane: reads stay in range, nothing is deposited
    const size_t ev = (size_t)ls * N + s;
    const size_t plane = (size_t)T.n_sites * N;
    VglSiteInfo si;
    uint64_t ad4;
    if constexpr (FUSED) { si.status = f_status; si.n_alleles = f_nall; si.acgt2alleles = 0; si.alleles2acgt = 0; ad4 = s_lds.f_ad4[otid]; }   // (GL model 2 looks at the count and the status only)
    else { si = T.sinfo[ls]; ad4 = T.ad4[ev]; }
    const int nA = si.n_alleles;
    const bool have = (si.status == SITE_OK);
    const float MISS = f32_missing();
    const int dp = (have && live) ? (int)((ad4 & 0xFFFF) + ((ad4 >> 16) & 0xFFFF) + ((ad4 >> 32) & 0xFFFF) + ((ad4 >> 48) & 0xFFFF)) : 0;
    // FUSED: the staged reads of the evaluation (at most 64, sixteen per word) from LDS into registers; behind the barrier the words'
    // place belongs to the deposits
    uint32_t fw[FUSED ? FUSEDW : 1];
    if constexpr (FUSED) {
#pragma unroll
        for (int k = 0; k < FUSEDW; ++k) { fw[k] = 0; if (dp > 16 * k) fw[k] = s_fw[k * WG + otid]; }
        __syncthreads();
    } else fw[0] = 0;

    float acc[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) acc[i] = -0.0f;                                     // bcf_utils.h:310

    if (GLM == 2) {
        if (dp > 0) {
            // gl_methods.cpp:22-59 / :94-139 / :171-220.  The reference updates every genotype (i, j) per read with one of three
            // terms chosen by (read allele == i, == j), then subtracts the maximum: genotypes that see the same sequence of
            // choices hold the same value after every read.  For an evaluation whose reads show k distinct bases those
            // sequences are: k homozygotes of a present allele, k (k - 1) / 2 heterozygotes of two present alleles, k "present /
            // absent" heterozygotes, and ONE for every genotype of absent alleles (term homF at every read) -- 3 / 6 / 10 / 15
            // accumulators for k = 1 .. 4 instead of A (A + 1) / 2 = 15, each updated by exactly the reference's operations.  A
            // wavefront runs the loop built for the largest k among its lanes (a lane with fewer present bases leaves the higher
            // slots to duplicates of its absent-allele sequences); the genotype-ordered values are read back through LDS.
            const bool per_read = !FUSED && (P.error_qs == 2);
            const uint32_t p0 = (ad4 & 0xFFFFULL) != 0, p1 = ((ad4 >> 16) & 0xFFFF) != 0, p2 = ((ad4 >> 32) & 0xFFFF) != 0, p3 = (ad4 >> 48) != 0;
            const int k_pres = (int)(p0 + p1 + p2 + p3);
            const uint32_t cmap = (p0 << 2) | ((p0 + p1) << 4) | ((p0 + p1 + p2) << 6);          // 2 bits per base: its rank among the present bases
            const bool has_abs = nA > k_pres;                                                       // an allele of the site this evaluation has no read of
            const int K = 1 + (__ballot(k_pres >= 2) != 0) + (__ballot(k_pres >= 3) != 0) + (__ballot(k_pres >= 4) != 0);   // wave-uniform
            auto read_loop = [&](auto k_tag, auto full_tag) {
                constexpr int KK = decltype(k_tag)::value;
                constexpr bool FULL = decltype(full_tag)::value;                                  // every lane has an absent allele: all slots stand for real genotypes
                constexpr int NT = (KK + 1) * (KK + 2) / 2;                                       // genotypes of KK + 1 "alleles": present base 0 .. KK-1, and KK = any absent allele
                float tr[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) tr[i] = -0.0f;                                       // bcf_utils.h:310
                double homT = P.pre_homT, het = P.pre_het, homF = P.pre_homF;
                // the staged reads of an evaluation: four per 32-bit word, words one plane (n_sites x n_samples words) apart
                // (vgl_read_byte); three words = 12 reads are kept in flight
                const uint32_t* colw = (const uint32_t*)T.reads + ev;
                const int lastw = (dp - 1) >> 2;
                uint32_t w0 = 0, w1 = 0, w2 = 0;
                if constexpr (!FUSED) { w0 = colw[0]; w1 = colw[(size_t)(1 < lastw ? 1 : lastw) * plane]; w2 = colw[(size_t)(2 < lastw ? 2 : lastw) * plane]; }
                // --precise-gl 1: the trip's four error probabilities (evaluation-major planes, vgl_errp_index: 32 contiguous, 32-byte aligned bytes per lane -- read_cap is
                // a multiple of four) as two 16-byte loads, asked for one trip ahead like the staged words
                typedef double v2d __attribute__((ext_vector_type(2)));
                const v2d* const ecol = (PREC && per_read) ? (const v2d*)(T.errp + vgl_errp_index(0, ev, P.read_cap)) : nullptr;
                v2d en0 = {0.0, 0.0}, en1 = {0.0, 0.0};
                if constexpr (PREC && !FUSED) { if (per_read) { en0 = ecol[0]; en1 = ecol[1]; } }
                auto one_read = [&](const uint32_t rb, const int r, const double e_r) {
                    const int ci = (int)((cmap >> ((rb & 3) * 2)) & 3);                           // the read's base among the present ones
                    if (per_read) {
                        if (!PREC) {
                            const int q = (int)(rb >> 2);
                            if (__builtin_expect(q < QL, 1)) { homT = s_q2gl[q]; het = s_q2gl[QL + q]; homF = s_q2gl[2 * QL + q]; }
                            else { homT = P.q2gl[q]; het = P.q2gl[257 + q]; homF = P.q2gl[514 + q]; }
                        } else {
                            const double e = e_r;
                            if (0.0 == e) { homT = 0.0; het = -0.30103; homF = -INFINITY; }
                            else {
                                // e / 3.0 and e / 6.0 correctly rounded, as the reference's divisions are: q0 = e RN(1/3), one residual and one
                                // correction (Markstein: exact for a correctly rounded reciprocal), and e / 6 = (e / 3) / 2 (halving is exact)
                                const double third = 0.33333333333333331;
                                const double q0 = e * third;
                                const double e3 = __builtin_fma(__builtin_fma(-3.0, q0, e), third, q0);
                                homT = log10_unit(1.0 - e); het = log10_unit((1.0 - e) / 2.0 + e3 * 0.5); homF = log10_unit(e3);
                            }
                        }
                    }
                    float mx = -INFINITY;
                    auto step = [&](float& a, const double t, const bool valid) {
                        const float v = (float)((double)a + t);
                        a = v;
                        if (FULL) mx = __builtin_fmaxf(mx, v);                                    // = (v > mx) ? v : mx: v is never NaN-greater, mx never NaN
                        else if (valid) mx = (v > mx) ? v : mx;
                    };
#pragma unroll
                    for (int b = 0; b <= KK; ++b) {
#pragma unroll
                        for (int a = 0; a <= b; ++a) {
                            double t;
                            if (b == KK) t = (a == KK) ? homF : ((KK == 1 || ci == a) ? het : homF);            // one or both alleles absent
                            else if (a == b) t = (KK == 1 || ci == a) ? homT : homF;
                            else t = (KK == 2 || ci == a || ci == b) ? het : homF;
                            step(tr[b * (b + 1) / 2 + a], t, b < k_pres || has_abs);
                        }
                    }
                    // tr[i] -= mx, two per instruction (v_pk_add_f32: an IEEE float32 subtraction per half, the same result as scalar subtractions)
                    {
                        const v2f m2 = {mx, mx};
#pragma unroll
                        for (int i = 0; i + 1 < NT; i += 2) {
                            v2f a = {tr[i], tr[i + 1]};
                            a = a - m2;
                            tr[i] = a.x; tr[i + 1] = a.y;
                        }
                        if (NT & 1) tr[NT - 1] -= mx;
                    }
                };
                // four reads per trip: one staged word, its bytes taken with constant shifts
                if constexpr (FUSED) {
                    for (int r0 = 0; r0 < dp; r0 += 4) {                                      // sixteen reads per word, two bits each (the base; the score is fixed)
                        const int wi = r0 >> 4;
                        uint32_t curw = fw[0];
#pragma unroll
                        for (int k = 1; k < (FUSED ? FUSEDW : 1); ++k) curw = (wi == k) ? fw[k] : curw;
                        const uint32_t cur = curw >> ((r0 & 15) * 2);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (r0 + j < dp) one_read((cur >> (2 * j)) & 3u, r0 + j, 0.0);
                    }
                } else
                for (int r0 = 0; r0 < dp; r0 += 4) {                                          // (r0 is the same in every active lane)
                    const uint32_t cur = w0;
                    w0 = w1; w1 = w2;
                    const int nw = (r0 >> 2) + 3;
                    w2 = colw[(size_t)(nw < lastw ? nw : lastw) * plane];
#ifndef VGL_ERRP_NOT_AHEAD
                    const v2d ec0 = en0, ec1 = en1;
                    if constexpr (PREC) { if (per_read) { const int nt = (r0 >> 2) + 1; const int t = nt < lastw ? nt : lastw; en0 = ecol[2 * t]; en1 = ecol[2 * t + 1]; } }
#else
                    v2d ec0 = en0, ec1 = en1;
                    if constexpr (PREC) { if (per_read) { ec0 = ecol[2 * (r0 >> 2)]; ec1 = ecol[2 * (r0 >> 2) + 1]; } }
#endif
                    const double e4[4] = {ec0.x, ec0.y, ec1.x, ec1.y};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (r0 + j < dp) one_read((cur >> (8 * j)) & 0xFFu, r0 + j, e4[j]);
                }
                // deposit: column otid (= the evaluation's natural position in the workgroup), rows = genotype index over (present
                // base ranks 0..3, 4 = absent) whatever KK this wavefront ran with, so that the reader needs no KK
#pragma unroll
                for (int b = 0; b <= KK; ++b) {
#pragma unroll
                    for (int a_ = 0; a_ <= b; ++a_) {
                        const int row = (b < KK) ? b * (b + 1) / 2 + a_ : (a_ < KK ? 10 + a_ : 14);
                        s_x[row * WG + otid] = __float_as_uint(tr[b * (b + 1) / 2 + a_]);
                    }
                }
            };
            const bool all_full = __ballot(!has_abs) == 0;
            if (K == 1 && !per_read && P.gl2_run != nullptr) {
                // one fixed score and every evaluation of this wavefront shows ONE base: the accumulators after dp identical updates are a
                // function of dp alone, tabulated on the host by the same operations (VglDevParams::gl2_run) -- no read loop, no staged reads
                const float* const row = P.gl2_run + ((size_t)(has_abs ? 0 : (P.read_cap + 1)) + (size_t)(dp < P.read_cap ? dp : P.read_cap)) * 3;
                const float t0 = row[0], t1 = row[1], t2 = row[2];
                s_x[0 * WG + otid] = __float_as_uint(t0);                 // rows as read_loop<1> deposits them: hom present, present / absent, absent / absent
                s_x[10 * WG + otid] = __float_as_uint(t1);
                s_x[14 * WG + otid] = __float_as_uint(t2);
            } else
            if (all_full) {
                if (K == 1) read_loop(std::integral_constant<int, 1>{}, std::true_type{});
                else if (K == 2) read_loop(std::integral_constant<int, 2>{}, std::true_type{});
                else if (K == 3) read_loop(std::integral_constant<int, 3>{}, std::true_type{});
                else read_loop(std::integral_constant<int, 4>{}, std::true_type{});
            } else {
                if (K == 1) read_loop(std::integral_constant<int, 1>{}, std::false_type{});
                else if (K == 2) read_loop(std::integral_constant<int, 2>{}, std::false_type{});
                else if (K == 3) read_loop(std::integral_constant<int, 3>{}, std::false_type{});
                else read_loop(std::integral_constant<int, 4>{}, std::false_type{});
            }
        }
    } else {
        // GL model 1 (gl_methods.cpp:233-369; htslib errmod.c restated in vgl_host.cpp).  With one fixed qScore errmod_cal() reduces
        // to table lookups on the per-base depths; with per-read scores each base accumulates fk[i] beta[q][n][i] over its own reads
        // in descending quality (below).  `on`: the lane has an evaluation with reads (wave-level loops run for every lane).
        const bool on = dp > 0;
        int n = dp;
        int c[5]; double bs[5];
#pragma unroll
        for (int b = 0; b < 4; ++b) { c[b] = (int)((ad4 >> (16 * b)) & 0xFFFF); bs[b] = 0.0; }
        c[4] = 0; bs[4] = 0.0;
        if (__builtin_expect(n > 255, 0)) {
            // errmod_cal(): "if we exceed 255 bases, shuffle them to sample at random" -- ks_shuffle (a Fisher-Yates pass from
            // the end, j = (int)(hts_drand48() * i)) on htslib's private rand48 stream, then the first 255 reads.  The lane
            // shuffles its own column of the staged reads in place (scratch of the context, 1 byte per read).
            uint64_t st;
            if (T.hts_off) st = rand48_jump(*T.hts_base, (uint64_t)T.hts_off[ev]);                      // serial: the process-wide stream
            else st = rand48_jump(VGL_HTS_RAND48_X0, (T.site_hash[ls] * (uint64_t)N + (uint64_t)s) * VGL_HTS_TILE_STRIDE);
            uint8_t* const rd = T.reads;
            for (int i = n; i > 1; --i) {
                st = lcg_next(st);
                const int j = (int)(u01(st) * (double)i);
                const size_t pj = vgl_read_byte(j, plane, ev), pi = vgl_read_byte(i - 1, plane, ev);
                const uint8_t tmp = rd[pj]; rd[pj] = rd[pi]; rd[pi] = tmp;
            }
            n = 255;
            c[0] = c[1] = c[2] = c[3] = 0;
            for (int r = 0; r < n; ++r) {
                const int b = (int)(rd[vgl_read_byte(r, plane, ev)] & 3);
                c[0] += (b == 0); c[1] += (b == 1); c[2] += (b == 2); c[3] += (b == 3);
            }
        }
        if (P.error_qs != 2) {
            if (on) {
#pragma unroll
                for (int b = 0; b < 4; ++b) bs[b] = P.gl1_bsum[n * 256 + c[b]];
            }
        } else {
            // per-read qScores (gl_methods.cpp:233-302): errmod_cal() walks the reads in descending (qual, base) order and adds, per
            // base, fk[i] beta[qual][n][i] for the base's i-th read in that order (qual clamped to [4, 63]).  A counting sort per
            // lane replaces the sort: (1) the quality range [qbot, qtop] of the wavefront's reads; (2) per window of 16 quality
            // values from the top, a one-byte (base, quality) histogram per lane in LDS (64 rows x 64 lanes per wavefront, laid
            // over s_x) and a 64-bit presence mask in registers (a bin's count is only read where its bit is set: no zeroing);
            // (3) per base, a flat loop over the base's reads in descending quality: the next non-empty bin by a count-trailing-
            // zeros of the mask, one 8-byte gather of fk[i] beta[q][n][i] (host table, compact in n and i: L2 resident) and one
            // double add per read.  Each lane takes its bases in the order of their depth, so that the first loop carries the
            // lanes' main base and the others are short.  (Round 2: 256 zeroed bins per lane in 16 KB of LDS per wavefront, a scan
            // of all 256 bins with a divergent inner loop, gathers from a 32 MB table.)
            uint8_t* const hw = (uint8_t*)s_x + (size_t)wib * 4096 + lane;                     // row r of this lane: hw[64 r]
            const uint32_t* const colw = (const uint32_t*)T.reads + ev;
            const int nwords = on ? (n + 3) >> 2 : 0;
            int qhi = 0, qlo = 63;
            for (int w = 0; w < nwords; ++w) {
                const uint32_t word = colw[(size_t)w * plane];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int q = (int)((word >> (8 * j + 2)) & 63u);
                    q = q < 4 ? 4 : q;
                    const bool in = 4 * w + j < n;
                    qhi = (in && q > qhi) ? q : qhi; qlo = (in && q < qlo) ? q : qlo;
                }
            }
#pragma unroll
            for (int m_ = 32; m_ >= 1; m_ >>= 1) { const int a_ = __shfl_xor(qhi, m_, 64), b_ = __shfl_xor(qlo, m_, 64); qhi = a_ > qhi ? a_ : qhi; qlo = b_ < qlo ? b_ : qlo; }
            const int qtop = __builtin_amdgcn_readfirstlane(qhi), qbot = __builtin_amdgcn_readfirstlane(qlo);
            // the lane's bases by descending depth (a 5-exchange sorting network on depth << 2 | base)
            uint32_t k0_ = ((uint32_t)c[0] << 2) | 0u, k1_ = ((uint32_t)c[1] << 2) | 1u, k2_ = ((uint32_t)c[2] << 2) | 2u, k3_ = ((uint32_t)c[3] << 2) | 3u;
            auto cx = [](uint32_t& a_, uint32_t& b_) { const uint32_t hi_ = a_ > b_ ? a_ : b_, lo_ = a_ > b_ ? b_ : a_; a_ = hi_; b_ = lo_; };
            cx(k0_, k1_); cx(k2_, k3_); cx(k0_, k2_); cx(k1_, k3_); cx(k1_, k2_);
            const uint32_t ob[4] = {k0_ & 3u, k1_ & 3u, k2_ & 3u, k3_ & 3u};
            int ik[4] = {0, 0, 0, 0};                                   // reads of the lane's k-th base taken so far = the next read's rank
            double acck[4] = {0.0, 0.0, 0.0, 0.0};
            const uint32_t NC = (uint32_t)P.gl1_nc;
            const uint32_t row_n = (uint32_t)n * NC;
            const double* const tab = P.gl1_fkbeta;
            for (int wtop = qtop; wtop >= qbot; wtop -= 16) {          // wave-uniform; one round unless the scores span more than 16 values
                uint64_t pm = 0;                                        // bit (16 base + wtop - q): the bin holds reads
                for (int w = 0; w < nwords; ++w) {
                    const uint32_t word = colw[(size_t)w * plane];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t rb = (word >> (8 * j)) & 0xFFu;
                        int q = (int)(rb >> 2);
                        q = q < 4 ? 4 : q;
                        const int qi = wtop - q;
                        if (4 * w + j < n && qi >= 0 && qi < 16) {
                            const uint32_t row = ((rb & 3u) << 4) | (uint32_t)qi;
                            const uint64_t bit = 1ULL << row;
                            const uint32_t h = hw[row * 64u];
                            hw[row * 64u] = (uint8_t)((pm & bit) ? h + 1u : 1u);
                            pm |= bit;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t b = ob[k];
                    uint32_t m = (uint32_t)(pm >> (16u * b)) & 0xFFFFu;
                    if (m) {
                        uint32_t left = 0, rowq = 0;
                        int i = ik[k];
                        double acc_b = acck[k];
                        do {
                            const bool refill = left == 0u;             // then m != 0 (loop condition)
                            const uint32_t qi = (uint32_t)__builtin_ctz(m | 0x10000u);
                            const uint32_t hcnt = hw[((b << 4) | (qi & 15u)) * 64u];
                            m = refill ? (m & (m - 1u)) : m;
                            left = refill ? hcnt : left;
                            rowq = refill ? ((uint32_t)(wtop - 4) - qi) * NC * NC + row_n : rowq;
                            acc_b += tab[rowq + (uint32_t)i];
                            ++i; --left;
                        } while (m != 0u || left != 0u);
                        ik[k] = i; acck[k] = acc_b;
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
                bs[b] = (ob[0] == (uint32_t)b) ? acck[0] : ((ob[1] == (uint32_t)b) ? acck[1] : ((ob[2] == (uint32_t)b) ? acck[2] : acck[3]));
        }
        float mx = -INFINITY;
        // (round 4) only the genotypes some evaluation of the wavefront has: alleles beyond the largest allele count among its sites are
        // never written (C3's binary sites: 3 of 5 alleles, 6 of 15 genotypes)
        const int nA_w = 1 + (__ballot(on && nA >= 2) != 0) + (__ballot(on && nA >= 3) != 0) + (__ballot(on && nA >= 4) != 0) + (__ballot(on && nA >= 5) != 0);
        if (on) {
#pragma unroll
            for (int i = 0; i < A; ++i) {
                if (i >= nA_w) continue;                                  // wave-uniform
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const int idx = i * (i + 1) / 2 + j;
                    const int b1 = nib(si.alleles2acgt, j), b2 = nib(si.alleles2acgt, i);
                    float tmp1 = 0.0f; int tmp2 = 0;
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        if (k != b1 && k != b2) { tmp1 = (float)((double)tmp1 + bs[k]); tmp2 += c[k]; }
                    float q;
                    if (b1 == b2) q = tmp2 ? tmp1 : 0.0f;
                    else {
                        const int lo = b1 < b2 ? b1 : b2, hi = b1 < b2 ? b2 : b1;
                        int clo = 0, chi = 0;                              // counts of the (subsampled) pileup; the unobserved allele has none
#pragma unroll
                        for (int k = 0; k < 4; ++k) { clo = (k == lo) ? c[k] : clo; chi = (k == hi) ? c[k] : chi; }
                        const int cjk = clo + chi;                         // <= n <= 255
                        const double lh = P.gl1_lhet[cjk << 8 | chi];
                        q = tmp2 ? (float)(-4.343 * lh + (double)tmp1) : (float)(-4.343 * lh);
                    }
                    if (q < 0.0f) q = 0.0f;
                    const float v = -div10_f32(q);                          // = (float)((-1.0 * (double)q) / 10.0), gl_methods.cpp:285 (q >= 0, finite)
                    acc[idx] = v;
                    if (i < nA) mx = (v > mx) ? v : mx;
                }
            }
        }
        if (P.error_qs == 2) __syncthreads();                           // the histograms lay over s_x: every wavefront is done with its own before the deposits
        if (on) {
#pragma unroll
            for (int i = 0; i < NG; ++i) s_x[i * WG + otid] = __float_as_uint(acc[i] - mx);     // deposit: column otid, rows = genotypes
        }
    }

    // ---- GL / PL / GP planes (vcfgl.cpp:907-970; no-reads site :297-305), FORMAT/AD, ADF, ADR (vcfgl.cpp:806-843).
    // The lanes have worked in sorted order and left their accumulators in LDS column (natural position); from here every
    // thread handles the evaluation at its own natural position, so that each store of a wavefront is one contiguous segment of
    // a plane and no tag needs a transposition of its own.
    if constexpr (FUSED) {
        // ---- k_site (see above): sixteen threads of the wavefront that took the END of the sorted order (evaluations without reads, padding)
        // work out the allele order, thread m writes row-table entry m, the first one the site's record and outputs
        const int w_light = (WPB - 1 - (int)(bx & (WPB - 1))) & (WPB - 1);
        const int m = tid - 64 * w_light;
        if (m >= 0 && m < 16) {
            const int S = fsplit;
            const int site = (int)(bx / (uint32_t)S), part = (int)bx - site * S;
            int32_t acc9[9];
#pragma unroll
            for (int k = 1; k < 5; ++k) { acc9[k] = s_lds.f_acc[k]; acc9[4 + k] = acc9[k]; }
            acc9[0] = acc9[1] + acc9[2] + acc9[3] + acc9[4];
            const VglSiteOrder o = site_order(P, acc9[0], &acc9[1]);
            s_lds.f_rowmap[m] = site_rowmap_entry(P.A, o.a2b, (uint32_t)m);
            if (m == 0) { s_lds.f_si = site_info_of(o); if (part == 0) site_outputs(P, T, site, o, acc9); }
        }
    }
    __syncthreads();
    if constexpr (FUSED) {
        si_nv = s_lds.f_si;
        if (s0 < N && dp0 >= 0) rmap = s_lds.f_rowmap[rmask];
    }
    const bool live0 = (bx * (uint32_t)WPB + (uint32_t)(tid >> 6) < nwaves) && (s0 < N);
    VglSiteInfo si_n;                                                    // wave-uniform: to scalar registers, here where it is first needed
    if (GLM == 2) {
        si_n.status = __builtin_amdgcn_readfirstlane(si_nv.status); si_n.n_alleles = __builtin_amdgcn_readfirstlane(si_nv.n_alleles);
        si_n.alleles2acgt = (uint32_t)__builtin_amdgcn_readfirstlane((int)si_nv.alleles2acgt); si_n.acgt2alleles = 0;
    } else si_n = si_nv;
    const int nA0 = si_n.n_alleles, nG0 = nA0 * (nA0 + 1) / 2;
    const bool have0 = (si_n.status == SITE_OK);
    const int dpn = (int)((a & 0xFFFF) + ((a >> 16) & 0xFFFF) + ((a >> 32) & 0xFFFF) + ((a >> 48) & 0xFFFF));
    const bool sample_ok = have0 && live0 && dpn > 0;
    if (GLM == 2) {
#pragma unroll
        for (int i = 0; i < NG; ++i) acc[i] = MISS;                      // (unused until here with GL model 2; PL / GP look at sample_ok, not at the value)
    }
    if (sample_ok) {
        const uint32_t* colx = s_x + tid;
        if (GLM == 2) {
            // genotype idx (bcf_alleles2gt order) of the site's alleles -> accumulator row: four bits of the table entry loaded at the start
            const uint32_t rlo = (uint32_t)rmap, rhi = (uint32_t)(rmap >> 32);
#pragma unroll
            for (int idx = 0; idx < NG; ++idx) {
                const uint32_t m = ((idx < 8 ? rlo : rhi) >> (4 * (idx & 7))) & 15u;
                acc[idx] = __uint_as_float(colx[m * WG]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NG; ++i) acc[i] = __uint_as_float(colx[i * WG]);
        }
    }
    // VGL_PUT(base, KMAX, nK, expr of i): the loops are spelled out here (not in a lambda) so that acc[] stays in registers.
    // VGL_LAYOUT_PLANES: one store per plane, lanes = consecutive samples.  VGL_LAYOUT_SAMPLE_MAJOR (include/vcfgl_hip.h): the
    // record's array as simRecord keeps it, x[sample * nK + k] with the site's own nK -- the wavefront's 64 x nK values are one
    // contiguous run of the slab, so they pass through the wavefront's own columns of s_x (free once every lane has read its
    // accumulators: columns 64 wv .. 64 wv + 63 of every row belong to this wavefront alone) in linear order and leave as
    // nK fully coalesced stores of 256 bytes.
    const bool sm = (FUSED && !SMB) ? false : (P.out_layout == 1);        // VGL_LAYOUT_SAMPLE_MAJOR
    const int wv = tid >> 6;
    const uint32_t wb = 64u * (uint32_t)wv;
    const int ls_w = ls0, sb_w = sb0;                                    // (scalar: the natural-order wavefront's site and first sample)
    const bool wave_ok = bx * (uint32_t)WPB + (uint32_t)__builtin_amdgcn_readfirstlane(wv) < nwaves;
    const uint32_t nv = (wave_ok && sb_w < N) ? (uint32_t)((N - sb_w) < 64 ? (N - sb_w) : 64) : 0u;   // samples of this wavefront's chunk (FUSED: a site's workgroup may have wavefronts beyond its samples)
    const uint32_t nG0u = (uint32_t)__builtin_amdgcn_readfirstlane(nG0), nA0u = (uint32_t)__builtin_amdgcn_readfirstlane(nA0);
#define VGL_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// the rows of one plane group: the row pointer is wave-uniform and is kept in scalar registers (pinned: left to itself the compiler folds the
// lane into a 64-bit vector address and pays a 64-bit vector add per row), the lane is the 32-bit offset of the store
#define VGL_ROWS(BASE_, KMAX, EXPR)                                                                      \
    do {                                                                                                 \
        vgl_gu32* rp_ = (vgl_gu32*)((BASE_) + ((size_t)ls0 * (KMAX)) * N + (size_t)sb0);                 \
        _Pragma("unroll") for (int i = 0; i < (KMAX); ++i) {                                             \
            if constexpr (GLM == 2) asm volatile("" : "+s"(rp_));                                        \
            rp_[(uint32_t)lane] = (EXPR);                                                                \
            rp_ += N;                                                                                    \
        }                                                                                                \
    } while (0)
#define VGL_PUT(BASE, KMAX, NK, EXPR)                                                                    \
    do {                                                                                                 \
        uint32_t* const base_ = (uint32_t*)(BASE);                                                       \
        if (base_ && !sm) {                                                                              \
            if (live0) VGL_ROWS(base_, KMAX, EXPR);                                                      \
        } else if (base_) {                                                                              \
            const uint32_t nk_ = (NK);                                                                   \
            _Pragma("unroll") for (int i = 0; i < (KMAX); ++i)                                           \
                if ((uint32_t)i < nk_) { const uint32_t L_ = (uint32_t)lane * nk_ + (uint32_t)i; s_x[(L_ >> 6) * WG + wb + (L_ & 63u)] = (EXPR); } \
            VGL_WAVE_LDS_SYNC();                                                                         \
            uint32_t* const dst_ = base_ + (size_t)ls_w * (KMAX) * N + (size_t)sb_w * nk_;               \
            const uint32_t total_ = nv * nk_;                                                            \
            _Pragma("unroll") for (int k = 0; k < (KMAX); ++k) {                                         \
                const uint32_t L_ = 64u * (uint32_t)k + (uint32_t)lane;                                  \
                if (L_ < total_) dst_[L_] = s_x[k * WG + wb + (uint32_t)lane];                           \
            }                                                                                            \
            VGL_WAVE_LDS_SYNC();                                                                         \
        }                                                                                                \
    } while (0)
    if (GLM == 2 && T.gl && !sm && nG0u == (uint32_t)NG) {                // every genotype of the plane group exists at this site: the registers hold
        if (live0) VGL_ROWS((uint32_t*)T.gl, NG, __float_as_uint(acc[i]));  // the value or (no reads / site not written) the missing pattern already
    } else VGL_PUT(T.gl, NG, nG0u, __float_as_uint((sample_ok && i < nG0) ? acc[i] : MISS));
    VGL_PUT(T.pl, NG, nG0u, pl_of(acc[i], sample_ok && i < nG0));
    if (T.pl_u8) {                                                       // PL in one byte (capped at 255; 255 also stands for missing)
        if (!sm) {
            if (live0) {
                typedef __attribute__((address_space(1))) uint8_t gu8;
                gu8* rp = (gu8*)(T.pl_u8 + ((size_t)ls0 * NG) * N + (size_t)sb0);         // (scalar row pointer: VGL_ROWS)
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const uint32_t v = pl_of(acc[i], sample_ok && i < nG0);
                    if constexpr (GLM == 2) asm volatile("" : "+s"(rp));
                    rp[(uint32_t)lane] = (uint8_t)(v > 255u ? 255u : v);
                    rp += N;
                }
            }
        } else {
            uint8_t* const sb8 = (uint8_t*)s_x;
#pragma unroll
            for (int i = 0; i < NG; ++i)
                if ((uint32_t)i < nG0u) {
                    const uint32_t v = pl_of(acc[i], sample_ok && i < nG0);
                    const uint32_t Lb = (uint32_t)lane * nG0u + (uint32_t)i, W = Lb >> 2;
                    sb8[((((W >> 6) * WG) + wb + (W & 63u)) << 2) | (Lb & 3u)] = (uint8_t)(v > 255u ? 255u : v);
                }
            VGL_WAVE_LDS_SYNC();
            uint8_t* const dst = T.pl_u8 + (size_t)ls_w * NG * N + (size_t)sb_w * nG0u;
            const uint32_t total = nv * nG0u;                             // bytes of this wavefront's run
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                 // 64 x 15 bytes = 240 words at most
                const uint32_t b0 = 4u * (64u * (uint32_t)k + (uint32_t)lane);
                const uint32_t word = s_x[k * WG + wb + (uint32_t)lane];
                if (b0 + 4u <= total) __builtin_memcpy(dst + b0, &word, 4);
                else {
#pragma unroll
                    for (int j = 0; j < 3; ++j) if (b0 + (uint32_t)j < total) dst[b0 + j] = (uint8_t)(word >> (8 * j));
                }
            }
            VGL_WAVE_LDS_SYNC();
        }
    }
    if ((!FUSED || GEN) && T.gp) {                                       // GP = 10^GL normalised by its float32 sum in genotype order
        float sum_gps = 0.0f;
#pragma unroll
        for (int i = 0; i < NG; ++i) {                                   // (the likelihoods are not needed any more: reuse their registers)
            const bool valid = sample_ok && i < nG0;
            acc[i] = valid ? (float)exp10_nonpos((double)acc[i]) : 0.0f;
            if (valid) sum_gps += acc[i];
        }
        VGL_PUT(T.gp, NG, nG0u, __float_as_uint((sample_ok && i < nG0) ? acc[i] / sum_gps : MISS));
    }
    if ((!FUSED || GEN) && (T.fmt_ad || T.fmt_adf || T.fmt_adr)) {
        const uint64_t adf4 = (P.need_adf && live0) ? T.adf4[(size_t)ls0 * N + (size_t)sb0 + (size_t)lane] : a;
        VGL_PUT(T.fmt_ad, A, nA0u, (uint32_t)cnt_of(a, (have0 && i < nA0) ? nib(si_n.alleles2acgt, i) : 0xF));
        VGL_PUT(T.fmt_adf, A, nA0u, (uint32_t)cnt_of(adf4, (have0 && i < nA0) ? nib(si_n.alleles2acgt, i) : 0xF));
        VGL_PUT(T.fmt_adr, A, nA0u, (uint32_t)(cnt_of(a, (have0 && i < nA0) ? nib(si_n.alleles2acgt, i) : 0xF) - cnt_of(adf4, (have0 && i < nA0) ? nib(si_n.alleles2acgt, i) : 0xF)));
    }
#undef VGL_WAVE_LDS_SYNC
#undef VGL_PUT
#undef VGL_ROWS
}
template <int A, int GLM, bool PREC, int WPB, int FUSEDW_ = 0>
__global__ __launch_bounds__(64 * WPB) void k_gl(const VglDevParams P, const VglTilePtrs T) {
    k_gl_body<A, GLM, PREC, WPB, FUSEDW_>(P, T, blockIdx.x, gridDim.x);
}
// k_gl2's workgroups that could not keep their accumulators (their bit in T.gl2_redo): k_gl2_scan lists them, k_gl_redo runs k_gl's body on
// the two 512-evaluation halves of each listed workgroup -- every tag of those evaluations is written again
__global__ __launch_bounds__(256) void k_gl2_scan(const VglTilePtrs T, const uint32_t n_words) {
    for (uint32_t w = threadIdx.x; w < n_words; w += 256u) {
        uint32_t bits = T.gl2_redo[w];
        if (bits) {
            T.gl2_redo[w] = 0u;
            while (bits) {
                const uint32_t b = (uint32_t)__builtin_ctz(bits);
                T.gl2_redo_list[atomicAdd(T.gl2_redo_count, 1u)] = 32u * w + b;
                bits &= bits - 1u;
            }
        }
    }
}
template <int A>
__global__ __launch_bounds__(512) void k_gl_redo(const VglDevParams P, const VglTilePtrs T) {
    const uint32_t n = *T.gl2_redo_count;
    for (uint32_t li = blockIdx.x; li < 2u * n; li += gridDim.x) {
        k_gl_body<A, 2, false, 8, 0>(P, T, 2u * T.gl2_redo_list[li >> 1] + (li & 1u), 0u);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// k_gl2 (round 5, second session): GL model 2 of the three-kernel path with TWO evaluations per thread.  k_gl is latency-bound in the
// workgroups a CU holds (1.24 + 3.98 / n ms at fixed-q, n = 4: docs/tried.md), and a CU's 2048 thread slots and its LDS both stop at 2048
// evaluations in flight.  Here a workgroup of 512 threads takes 1024 evaluations (sixteen natural wavefronts: thread t the positions t and
// t + 512) -- one counting sort over all of them; thread t works on sorted position q = (t + rotation) mod 512 of the heavy half and on
// 1023 - q of the light half, so that every wavefront has loop work -- and the accumulators lie in the compact array described at
// gl2_row_code(): 6 full rows + 9 rows of VGL_GL2_OVC columns = 33 KB for 1024 evaluations (k_gl: 30 KB for 512).  Column = the
// evaluation's sorted position (the natural thread knows it from its insertion).  A workgroup with more than VGL_GL2_OVC three- or
// four-base evaluations cannot keep their upper rows: it sets its bit in T.gl2_redo and k_gl (REDO instantiation) works on it again.
// Planes layout, no --precise-gl 1, sort on: everything else stays with k_gl.
template <int A>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_gl2(const VglDevParams P, const VglTilePtrs T) {
    constexpr int WPB = 8, WG = 512, WGE = 1024, NW = 16;               // threads, evaluations, natural wavefronts per workgroup
    constexpr int QL = 96;
    constexpr int NG = A * (A + 1) / 2;
    constexpr int OVC = VGL_GL2_OVC;
#ifdef VGL_TEST_HOOKS
    constexpr bool RICH = true;                                        // (VGL_GL2X=2: every tag, for the tests)
#else
    constexpr bool RICH = false;                                       // vgl_launch_gl keeps tiles with GP or FORMAT/AD* on k_gl: the shipped kernel does not carry their epilogue
#endif
    struct Lds {
        double q2gl[3 * QL];
        uint32_t x[6 * WGE + 9 * OVC];
        uint16_t perm[WGE];
        int32_t ws[2 * NW];
        int32_t n3p;
        // the epilogue reads accumulator row c of an evaluation at x + 4 pos + 512 c for every genotype of the site, also for an evaluation whose pool rows
        // (codes 48 ... 64) do not exist because its sorted position is beyond VGL_GL2_OVC -- such a workgroup has set its redo bit and k_gl_redo writes all of
        // its tags again, so the value is never used; the pad keeps even that read (at most 4092 + 512 x 64 bytes from x) inside this structure
        uint32_t guard[224];
    };
    static_assert(sizeof(Lds) - offsetof(Lds, x) >= 4092 + 512 * 64 + 4, "k_gl2: every accumulator address of the epilogue lies inside the LDS block");
    __shared__ Lds s_lds;
    double* const s_q2gl = s_lds.q2gl;
    uint32_t* const s_x = s_lds.x;
    uint16_t* const s_perm = s_lds.perm;
    int32_t* const s_ws = s_lds.ws;
    uint32_t* const s_hist = s_x;
    const int N = P.n_samples;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (P.error_qs == 2) for (int i = tid; i < 3 * QL; i += WG) s_q2gl[i] = P.q2gl[(i / QL) * 257 + (i % QL)];
    const uint32_t chunks_k = (uint32_t)P.chunks;
    const uint32_t nwaves = (uint32_t)T.n_sites * chunks_k;
    const uint32_t bx = P.xcd_map ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t wg_ls = (bx * (uint32_t)NW) / chunks_k, wg_rem = (bx * (uint32_t)NW) - wg_ls * chunks_k;
    auto wave_site = [&](const int k, int& ls_, int& s_base) {
        const int t = (int)wg_rem + k, c = (int)chunks_k;
        int add = 0;
#pragma unroll
        for (int j = 1; j < NW; ++j) add += (t >= j * c) ? 1 : 0;
        ls_ = (int)wg_ls + add;
        s_base = (t - add * c) * 64;
    };
    constexpr int NK = 4;
    const int sort_sh = P.read_cap > 511 ? 2 : (P.read_cap > 255 ? 1 : 0);
    const int nbd = (P.read_cap >> sort_sh) + 1;
    const int nb = NK * nbd + 2;
    for (int i = tid; i < nb; i += WG) s_hist[i] = 0;
    // ---- the two evaluations at this thread's natural positions
    int ls0[2] = {0, 0}, sb0[2] = {0, 0}, s0[2] = {N, N}, dp0[2] = {-1, -1}, key[2];
    uint64_t a[2] = {0, 0}, rm_lo[2] = {0, 0}, rm_hi[2] = {0, 0};
    bool wave_ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int wn = wv + WPB * h;                                   // natural wavefront of the workgroup
        const uint32_t w = bx * (uint32_t)NW + (uint32_t)wn;
        wave_ok[h] = w < nwaves;
        int k0 = 1;
        if (wave_ok[h]) {
            wave_site(wn, ls0[h], sb0[h]);
            ls0[h] = __builtin_amdgcn_readfirstlane(ls0[h]); sb0[h] = __builtin_amdgcn_readfirstlane(sb0[h]);
            if (lane == 0) { s_ws[2 * wn] = ls0[h]; s_ws[2 * wn + 1] = sb0[h]; }
            s0[h] = sb0[h] + lane;
            if (s0[h] < N) {
                a[h] = T.ad4[(size_t)ls0[h] * N + (size_t)sb0[h] + (size_t)lane];
                dp0[h] = (int)((a[h] & 0xFFFF) + ((a[h] >> 16) & 0xFFFF) + ((a[h] >> 32) & 0xFFFF) + ((a[h] >> 48) & 0xFFFF));
                if (dp0[h] > 1023) dp0[h] = 1023;
                const uint32_t q0 = (a[h] & 0xFFFFULL) != 0, q1 = ((a[h] >> 16) & 0xFFFF) != 0, q2 = ((a[h] >> 32) & 0xFFFF) != 0, q3 = (a[h] >> 48) != 0;
                k0 = (int)(q0 + q1 + q2 + q3);
                const uint64_t* const e = T.rowmap8 + ((size_t)ls0[h] * 16 + (q0 | (q1 << 1) | (q2 << 2) | (q3 << 3))) * 2;
                rm_lo[h] = e[0]; rm_hi[h] = e[1];                       // (in flight during the loops)
            }
        }
        int kk = nb - 1;
        if (dp0[h] == 0) kk = nb - 2;
        else if (dp0[h] > 0) {
            const int dq = (dp0[h] > P.read_cap ? P.read_cap : dp0[h]) >> sort_sh;
            kk = (NK - k0) * nbd + ((k0 == 2 && P.gl_flip2) ? dq : nbd - 1 - dq);
        }
        key[h] = kk;
    }
    __syncthreads();
    atomicAdd(&s_hist[key[0]], 1u);
    atomicAdd(&s_hist[key[1]], 1u);
    __syncthreads();
    if (tid < 64) {
        uint32_t run = 0;
        for (int base = 0; base < nb; base += 64) {
            const int i = base + tid;
            const uint32_t v = (i < nb) ? s_hist[i] : 0u;
            const uint32_t incl = wave_incl_scan_u32(v);
            if (i < nb) s_hist[i] = run + incl - v;
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (tid == 0) s_lds.n3p = (int32_t)s_hist[2 * nbd];            // evaluations with three or four bases: the head of the sorted order
    }
    __syncthreads();
    uint32_t pos[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { pos[h] = atomicAdd(&s_hist[key[h]], 1u); s_perm[pos[h]] = (uint16_t)(tid + WG * h); }
    __syncthreads();
    const int n3p = s_lds.n3p;
    if (n3p > (P.dbg_gl2_ovc > 0 ? P.dbg_gl2_ovc : OVC) && tid == 0) atomicOr(&T.gl2_redo[bx >> 5], 1u << (bx & 31u));     // (rare: k_gl_redo takes the workgroup's evaluations again)
    const size_t plane = (size_t)T.n_sites * N;
    const float MISS = f32_missing();
    // ---- the two evaluations this thread works on: sorted position q of the heavy half, 1023 - q of the light half
    const int rot_q = (tid + 64 * (int)(bx & (WPB - 1))) & (WG - 1);
    // (asking for both evaluations' records ahead of the first loop, in either order, measured slower: the loops then spill)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = h ? (WGE - 1 - rot_q) : rot_q;                     // its column of the compact array
        const int oe = s_perm[q];                                        // natural position (0 .. 1023) of the evaluation
        const uint32_t w = bx * (uint32_t)NW + (uint32_t)(oe >> 6);
        int ls = 0, s = N;
        if (w < nwaves) { ls = s_ws[2 * (oe >> 6)]; s = s_ws[2 * (oe >> 6) + 1] + (oe & 63); }
        const bool live = s < N;
        if (!live) { ls = 0; s = 0; }
        const size_t ev = (size_t)ls * N + s;
        const VglSiteInfo si = T.sinfo[ls];
        const uint64_t ad4 = T.ad4[ev];
        const int nA = si.n_alleles;
        const bool have = (si.status == SITE_OK);
        const int dp = (have && live) ? (int)((ad4 & 0xFFFF) + ((ad4 >> 16) & 0xFFFF) + ((ad4 >> 32) & 0xFFFF) + ((ad4 >> 48) & 0xFFFF)) : 0;
        if (dp > 0) {
            const bool per_read = (P.error_qs == 2);
            const uint32_t p0 = (ad4 & 0xFFFFULL) != 0, p1 = ((ad4 >> 16) & 0xFFFF) != 0, p2 = ((ad4 >> 32) & 0xFFFF) != 0, p3 = (ad4 >> 48) != 0;
            const int k_pres = (int)(p0 + p1 + p2 + p3);
            const uint32_t cmap = (p0 << 2) | ((p0 + p1) << 4) | ((p0 + p1 + p2) << 6);
            const bool has_abs = nA > k_pres;
            const int K = 1 + (__ballot(k_pres >= 2) != 0) + (__ballot(k_pres >= 3) != 0) + (__ballot(k_pres >= 4) != 0);
            const bool ov_ok = q < OVC;                                  // (the upper rows of a three- / four-base evaluation)
            uint32_t* const col = s_x + q;
            // row r of k_gl's numbering -> this array (gl2_row_code): full rows of 4096 bytes, then rows of 512
            auto put = [&](const int row, const float v, const int kmin) {
                const int ph = gl2_row_phys(row);
                if (k_pres >= kmin) {
                    if (ph < 6) col[ph * WGE] = __float_as_uint(v);
                    else if (ov_ok) col[6 * WGE + (ph - 6) * OVC] = __float_as_uint(v);
                }
            };
            auto read_loop = [&](auto k_tag, auto full_tag) {
                constexpr int KK = decltype(k_tag)::value;
                constexpr bool FULL = decltype(full_tag)::value;
                constexpr int NT = (KK + 1) * (KK + 2) / 2;
                float tr[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) tr[i] = -0.0f;
                double homT = P.pre_homT, het = P.pre_het, homF = P.pre_homF;
                const uint32_t* colw = (const uint32_t*)T.reads + ev;
                const int lastw = (dp - 1) >> 2;
                uint32_t w0 = colw[0], w1 = colw[(size_t)(1 < lastw ? 1 : lastw) * plane], w2 = colw[(size_t)(2 < lastw ? 2 : lastw) * plane];
                auto one_read = [&](const uint32_t rb) {
                    const int ci = (int)((cmap >> ((rb & 3) * 2)) & 3);
                    if (per_read) {
                        const int qv = (int)(rb >> 2);
                        if (__builtin_expect(qv < QL, 1)) { homT = s_q2gl[qv]; het = s_q2gl[QL + qv]; homF = s_q2gl[2 * QL + qv]; }
                        else { homT = P.q2gl[qv]; het = P.q2gl[257 + qv]; homF = P.q2gl[514 + qv]; }
                    }
                    float mx = -INFINITY;
                    auto step = [&](float& ac, const double t, const bool valid) {
                        const float v = (float)((double)ac + t);
                        ac = v;
                        if (FULL) mx = __builtin_fmaxf(mx, v);
                        else if (valid) mx = (v > mx) ? v : mx;
                    };
#pragma unroll
                    for (int b = 0; b <= KK; ++b) {
#pragma unroll
                        for (int a_ = 0; a_ <= b; ++a_) {
                            double t;
                            if (b == KK) t = (a_ == KK) ? homF : ((KK == 1 || ci == a_) ? het : homF);
                            else if (a_ == b) t = (KK == 1 || ci == a_) ? homT : homF;
                            else t = (KK == 2 || ci == a_ || ci == b) ? het : homF;
                            step(tr[b * (b + 1) / 2 + a_], t, b < k_pres || has_abs);
                        }
                    }
                    {
                        const v2f m2 = {mx, mx};
#pragma unroll
                        for (int i = 0; i + 1 < NT; i += 2) {
                            v2f t2 = {tr[i], tr[i + 1]};
                            t2 = t2 - m2;
                            tr[i] = t2.x; tr[i + 1] = t2.y;
                        }
                        if (NT & 1) tr[NT - 1] -= mx;
                    }
                };
                for (int r0 = 0; r0 < dp; r0 += 4) {
                    const uint32_t cur = w0;
                    w0 = w1; w1 = w2;
                    const int nw = (r0 >> 2) + 3;
                    w2 = colw[(size_t)(nw < lastw ? nw : lastw) * plane];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (r0 + j < dp) one_read((cur >> (8 * j)) & 0xFFu);
                }
                // deposit: the rows this evaluation's own set of bases reaches (the natural thread reads no others)
#pragma unroll
                for (int b = 0; b <= KK; ++b) {
#pragma unroll
                    for (int a_ = 0; a_ <= b; ++a_) {
                        const int row = (b < KK) ? b * (b + 1) / 2 + a_ : (a_ < KK ? 10 + a_ : 14);
                        const int kmin = (b < KK) ? b + 1 : (a_ < KK ? a_ + 1 : 1);     // the evaluation has this row iff it shows at least kmin bases
                        put(row, tr[b * (b + 1) / 2 + a_], kmin);
                    }
                }
            };
            const bool all_full = __ballot(!has_abs) == 0;
            if (K == 1 && !per_read && P.gl2_run != nullptr) {
                const float* const rw = P.gl2_run + ((size_t)(has_abs ? 0 : (P.read_cap + 1)) + (size_t)(dp < P.read_cap ? dp : P.read_cap)) * 3;
                const float t0 = rw[0], t1 = rw[1], t2 = rw[2];
                put(0, t0, 1); put(10, t1, 1); put(14, t2, 1);
            } else if (all_full) {
                if (K == 1) read_loop(std::integral_constant<int, 1>{}, std::true_type{});
                else if (K == 2) read_loop(std::integral_constant<int, 2>{}, std::true_type{});
                else if (K == 3) read_loop(std::integral_constant<int, 3>{}, std::true_type{});
                else read_loop(std::integral_constant<int, 4>{}, std::true_type{});
            } else {
                if (K == 1) read_loop(std::integral_constant<int, 1>{}, std::false_type{});
                else if (K == 2) read_loop(std::integral_constant<int, 2>{}, std::false_type{});
                else if (K == 3) read_loop(std::integral_constant<int, 3>{}, std::false_type{});
                else read_loop(std::integral_constant<int, 4>{}, std::false_type{});
            }
        }
    }
    __syncthreads();
    // ---- tags of the two evaluations at this thread's natural positions (k_gl's epilogue, planes layout)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        VglSiteInfo si_n;
        {
            const VglSiteInfo si_nv = T.sinfo[ls0[h]];
            si_n.status = __builtin_amdgcn_readfirstlane(si_nv.status); si_n.n_alleles = __builtin_amdgcn_readfirstlane(si_nv.n_alleles);
            si_n.alleles2acgt = (uint32_t)__builtin_amdgcn_readfirstlane((int)si_nv.alleles2acgt); si_n.acgt2alleles = 0;
        }
        const bool live0 = wave_ok[h] && (s0[h] < N);
        const int nA0 = si_n.n_alleles, nG0 = nA0 * (nA0 + 1) / 2;
        const bool have0 = (si_n.status == SITE_OK);
        const int dpn = (int)((a[h] & 0xFFFF) + ((a[h] >> 16) & 0xFFFF) + ((a[h] >> 32) & 0xFFFF) + ((a[h] >> 48) & 0xFFFF));
        const bool sample_ok = have0 && live0 && dpn > 0;
        float acc[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) acc[i] = MISS;
        if (sample_ok) {
            const uint8_t* const colb = (const uint8_t*)(s_x + pos[h]);
#pragma unroll
            for (int idx = 0; idx < NG; ++idx) {
                const uint32_t c = (uint32_t)(((idx < 8 ? rm_lo[h] : rm_hi[h]) >> (8 * (idx & 7))) & 0xFFu);
                acc[idx] = __uint_as_float(*(const uint32_t*)(colb + 512u * c));
            }
        }
        const uint32_t nG0u = (uint32_t)__builtin_amdgcn_readfirstlane(nG0), nA0u = (uint32_t)__builtin_amdgcn_readfirstlane(nA0);
        const int lsh = ls0[h], sbh = sb0[h];
#define VGL2_ROWS(BASE_, KMAX, EXPR)                                                                     \
    do {                                                                                                 \
        vgl_gu32* rp_ = (vgl_gu32*)((BASE_) + ((size_t)lsh * (KMAX)) * N + (size_t)sbh);                 \
        _Pragma("unroll") for (int i = 0; i < (KMAX); ++i) {                                             \
            asm volatile("" : "+s"(rp_));                                                                \
            rp_[(uint32_t)lane] = (EXPR);                                                                \
            rp_ += N;                                                                                    \
        }                                                                                                \
    } while (0)
#define VGL2_PUT(BASE, KMAX, EXPR) do { uint32_t* const base_ = (uint32_t*)(BASE); if (base_ && live0) VGL2_ROWS(base_, KMAX, EXPR); } while (0)
        if (T.gl && nG0u == (uint32_t)NG) { if (live0) VGL2_ROWS((uint32_t*)T.gl, NG, __float_as_uint(acc[i])); }
        else VGL2_PUT(T.gl, NG, __float_as_uint((sample_ok && i < nG0) ? acc[i] : MISS));
        VGL2_PUT(T.pl, NG, pl_of(acc[i], sample_ok && i < nG0));
        if (T.pl_u8 && live0) {
            typedef __attribute__((address_space(1))) uint8_t gu8;
            gu8* rp = (gu8*)(T.pl_u8 + ((size_t)lsh * NG) * N + (size_t)sbh);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const uint32_t v = pl_of(acc[i], sample_ok && i < nG0);
                asm volatile("" : "+s"(rp));
                rp[(uint32_t)lane] = (uint8_t)(v > 255u ? 255u : v);
                rp += N;
            }
        }
        if (RICH && T.gp) {
            float sum_gps = 0.0f;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const bool valid = sample_ok && i < nG0;
                acc[i] = valid ? (float)exp10_nonpos((double)acc[i]) : 0.0f;
                if (valid) sum_gps += acc[i];
            }
            VGL2_PUT(T.gp, NG, __float_as_uint((sample_ok && i < nG0) ? acc[i] / sum_gps : MISS));
        }
        if (RICH && (T.fmt_ad || T.fmt_adf || T.fmt_adr)) {
            const uint64_t adf4 = (P.need_adf && live0) ? T.adf4[(size_t)lsh * N + (size_t)sbh + (size_t)lane] : a[h];
            VGL2_PUT(T.fmt_ad, A, (uint32_t)cnt_of(a[h], (have0 && i < nA0) ? nib(si_n.alleles2acgt, i) : 0xF));
            VGL2_PUT(T.fmt_adf, A, (uint32_t)cnt_of(adf4, (have0 && i < nA0) ? nib(si_n.alleles2acgt, i) : 0xF));
            VGL2_PUT(T.fmt_adr, A, (uint32_t)(cnt_of(a[h], (have0 && i < nA0) ? nib(si_n.alleles2acgt, i) : 0xF) - cnt_of(adf4, (have0 && i < nA0) ? nib(si_n.alleles2acgt, i) : 0xF)));
        }
        (void)nA0u;
#undef VGL2_PUT
#undef VGL2_ROWS
    }
}

// ------------------------------------------------------------------------------------
// INFO/QS and INFO/I16 (vcfgl.cpp:845-898, 982-1074): float32 sums over a site's samples IN SAMPLE ORDER, which a tree reduction would not
// reproduce bit for bit.  One wavefront takes SIXTEEN sites:
//   * QS: qs[allele of b] += (float)qsum[s][b] / sum_s is an order-dependent chain per (site, base) -- 64 chains per wavefront.  The
//     wavefront walks the samples in chunks of 64: for each of its sites the lanes (lane = sample) load the chunk's four quality sums
//     (coalesced, 256 bytes per row), form the four correctly rounded quotients and leave them in LDS; then lane (site j, base b) adds
//     the 64 values of ITS chain in sample order.  A sample without reads, or a base without an allele, contributes +0 where the
//     reference skips it: x + 0 = x for these non-negative sums.  (Round 3: one lane per site walking N samples with 16 KB between the
//     lanes' addresses: 7.1 ms per 65536 x 1000 tile.)
//   * I16 fields 5-8 are float32 running sums of INTEGERS (per-sample quality sums): while the total stays at or below 2^24 every partial
//     sum is an integer the format holds exactly, so the result is the integer total whatever the order -- k_sample leaves those totals
//     in acc[16..23].  Fields 9-12 add the constant mapping quality once per read: K additions of c are exact while K c <= 2^24 2^tz(c)
//     (every partial sum is a multiple of 2^tz(c)); beyond that the remaining additions are performed one by one (no memory traffic).
//     A site whose integer totals leave the exact range is walked sample by sample by one lane, as the reference does (rare: more than
//     2^24 / q^2 reads of one base at a site).
// fields 1-4: per-base strand depths of the site (acc); fields 13-16: tail distances (T.site_tail: the scout's libc rand() draws in serial mode, k_tail's in tile mode).
#define VGL_AGG_SITES 16
#define VGL_AGG_ROW 65                                                 // 64 samples + 1 word of padding: the chain lanes' reads fall on distinct banks
// the reference's K-fold `acc += c` (acc starts at 0; c a positive integer-valued float): jump over the exact range, walk the rest
__device__ __forceinline__ float repeated_add(const float c, const long long K) {
    if (K <= 0 || c == 0.0f) return 0.0f;
    const unsigned long long ci = (unsigned long long)c;
    if ((float)ci != c || ci == 0ULL || ci >= (1ULL << 24)) { float v = 0.0f; for (long long k = 0; k < K; ++k) v += c; return v; }
    const int tz = __builtin_ctzll(ci);
    const unsigned long long lim = 1ULL << (24 + tz);                  // multiples of 2^tz up to here are float32 values
    const unsigned long long k0 = lim / ci;                            // k c <= lim for k <= k0
    if ((unsigned long long)K <= k0) return (float)((unsigned long long)K * ci);
    float v = (float)(k0 * ci);
    for (unsigned long long k = k0; k < (unsigned long long)K; ++k) v += c;
    return v;
}
__global__ __launch_bounds__(64) void k_siteagg(const VglDevParams P, const VglTilePtrs T) {
    __shared__ float s_x[VGL_AGG_SITES * 4 * VGL_AGG_ROW];
    const int lane = threadIdx.x;
    const int N = P.n_samples, A = P.A;
    const int ls_w = blockIdx.x * VGL_AGG_SITES;                        // first site of this wavefront
    const int n_w = (T.n_sites - ls_w) < VGL_AGG_SITES ? (T.n_sites - ls_w) : VGL_AGG_SITES;
    // ---- INFO/QS
    if (T.qs) {
        const int j_c = lane >> 2;                                      // this lane's chain: (site, base)
        float chain = 0.0f;
        uint64_t have_m = 0;                                            // sites of the wavefront that are written and carry quality sums
        {
            const bool hv = lane < n_w && P.need_qsum && T.sinfo[ls_w + lane].status == SITE_OK;
            have_m = __ballot(hv);
        }
        if (have_m) {
            for (int c0 = 0; c0 < N; c0 += 64) {
                const int sidx = c0 + lane;
                const bool in_n = sidx < N;
                const size_t s_ld = in_n ? (size_t)sidx : 0;             // (a valid address for every lane: the value is dropped below)
                // four sites per trip: their sixteen row loads are issued together (the kernel waits on memory, not on arithmetic)
                for (int j0 = 0; j0 < n_w; j0 += 4) {
                    uint32_t qv[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = (j0 + u < n_w) ? j0 + u : n_w - 1;
                        const uint32_t* q = T.qsum + (size_t)(ls_w + j) * 4 * N + s_ld;
#pragma unroll
                        for (int b = 0; b < 4; ++b) qv[u][b] = q[(size_t)b * N];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + u;
                        if (j < n_w) {                                   // wave-uniform
                            const bool on = in_n && ((have_m >> j) & 1);
                            const float f0 = (float)(int)qv[u][0], f1 = (float)(int)qv[u][1], f2 = (float)(int)qv[u][2], f3 = (float)(int)qv[u][3];
                            const float sum = (((0.0f + f0) + f1) + f2) + f3;                 // vcfgl.cpp:879-881, in base order
                            const double r = recip_int24(sum);                              // (float)((float)q / sum), vcfgl.cpp:893: quot_int24
                            const bool nz = on && (0.0f != sum);
                            float* row = s_x + (size_t)(j * 4) * VGL_AGG_ROW + lane;
                            row[0] = nz ? quot_int24(f0, r) : 0.0f; row[VGL_AGG_ROW] = nz ? quot_int24(f1, r) : 0.0f;
                            row[2 * VGL_AGG_ROW] = nz ? quot_int24(f2, r) : 0.0f; row[3 * VGL_AGG_ROW] = nz ? quot_int24(f3, r) : 0.0f;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (j_c < n_w) {
                    const float* mine = s_x + (size_t)lane * VGL_AGG_ROW;
#pragma unroll 16
                    for (int k = 0; k < 64; ++k) chain += mine[k];      // samples c0 .. c0 + 63 of chain (j_c, b_c), in order
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        // qs[a] of site j = the chain of base alleles2acgt[a]: the chain values go back through LDS, lane (j, a) picks its base's
        s_x[lane] = chain;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int t = lane; t < n_w * A; t += 64) {
            const int j = t / A, a = t - j * A;
            const VglSiteInfo si = T.sinfo[ls_w + j];
            float v = 0.0f;
            if (si.status == SITE_OK && P.need_qsum) {
#pragma unroll
                for (int b = 0; b < 4; ++b) if (nib(si.acgt2alleles, b) == a) v = s_x[j * 4 + b];
            }
            T.qs[(size_t)(ls_w + j) * A + a] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- INFO/I16: lane j < 16 takes site j
    if (T.i16) {
        const bool mine = lane < n_w;
        const int ls = ls_w + (mine ? lane : 0);
        const VglSiteInfo si = T.sinfo[ls];
        const bool have = mine && (si.status == SITE_OK);
        const int nA = si.n_alleles;
        const int nObs = (A == 5) ? nA - 1 : nA;
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.0f;
        const bool on = have && P.add_i16 && nA > 1;
        bool walk = false;                                              // this site's running sums leave the exact range of float32
        if (on) {
            const int32_t* acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
            const int refb = nib(si.alleles2acgt, 0);
            v[0] = (float)acc[5 + refb]; v[1] = (float)(acc[1 + refb] - acc[5 + refb]);
            const float mq = (float)P.i16_mapq, mq2 = (float)(P.i16_mapq * P.i16_mapq);
            // integer totals (modulo 2^32: trusted while the site's reads cannot have wrapped them -- 3969 = 63^2 per read at most)
            const bool tot_ok = (unsigned long long)(uint32_t)acc[0] * 3969ULL < (1ULL << 32);
            unsigned long long t4 = (uint32_t)acc[VGL_ACC_QSUM + refb], t5 = (uint32_t)acc[VGL_ACC_QSUMSQ + refb], t6 = 0, t7 = 0;
            long long k_ref = acc[1 + refb], k_non = 0;
            for (int a = 1; a < nA; ++a) {
                if (a == nObs) continue;
                const int b = nib(si.alleles2acgt, a);
                t6 += (uint32_t)acc[VGL_ACC_QSUM + b]; t7 += (uint32_t)acc[VGL_ACC_QSUMSQ + b];
                k_non += acc[1 + b];
                v[2] += (float)acc[5 + b]; v[3] += (float)(acc[1 + b] - acc[5 + b]);
            }
            const unsigned long long lim = 1ULL << 24;
            if (tot_ok && t4 <= lim && t5 <= lim && t6 <= lim && t7 <= lim) { v[4] = (float)t4; v[5] = (float)t5; v[6] = (float)t6; v[7] = (float)t7; }
            else walk = true;
            // one addition of the mapping quality (and of its square) per read, reference allele / the others (vcfgl.cpp:1003-1024)
            v[8] = repeated_add(mq, k_ref); v[9] = repeated_add(mq2, k_ref);
            v[10] = repeated_add(mq, k_non); v[11] = repeated_add(mq2, k_non);
            // tail distance (vcfgl.cpp:1029-1071): from the serial-mode scout (libc rand()) or from k_tail / k_tail_fin (tile mode)
            if (T.site_tail) {
                const VglSiteTail tl = T.site_tail[ls];
                if (tl.base == refb) { v[12] = tl.sum; v[13] = tl.sumsq; }
                for (int a = 1; a < nA; ++a) {
                    if (a == nObs) continue;
                    if (nib(si.alleles2acgt, a) == tl.base) { v[14] += tl.sum; v[15] += tl.sumsq; }
                }
            }
        }
        // Sites whose sums leave the exact range (more than 2^24 / q^2 reads of a base: wide, deep tiles): the reference's own walks
        // (vcfgl.cpp:997-1000 reference allele, 1052-1056 the other alleles one after the other), as float32 chains in sample order --
        // by the same LDS transposition as the QS chains: pass `a` feeds chain (site, 0) the quality sums and chain (site, 1) the squared
        // sums of the site's a-th allele (pass 0 into fields 5-6, the later passes into 7-8), 64 samples per coalesced row load.
        const uint64_t walk_m = __ballot(walk);
        if (walk_m) {
            const int j_c = lane >> 2, k_c = lane & 3;
            float ch = 0.0f, ch_ref = 0.0f;                             // this lane's chain (k_c < 2), and its value after pass 0
            for (int a = 0; a < A; ++a) {
                // base of the a-th allele of every walking site (wave-uniform mask; lanes j < 16 hold the sites)
                const int b_mine = (walk && a < nA && a != nObs) ? nib(si.alleles2acgt, a) : -1;
                const uint64_t pass_m = __ballot(b_mine >= 0 && b_mine < 4);
                if (a == 1) { ch_ref = ch; ch = 0.0f; }
                if (!pass_m) continue;
                for (int c0 = 0; c0 < N; c0 += 64) {
                    const int sidx = c0 + lane;
                    for (int j = 0; j < n_w; ++j) {
                        if (!((pass_m >> j) & 1)) continue;             // wave-uniform
                        const int b = __builtin_amdgcn_readlane(b_mine, j);
                        float x0 = 0.0f, x1 = 0.0f;
                        if (sidx < N) {
                            const size_t row = ((size_t)(ls_w + j) * 4 + (size_t)b) * N + (size_t)sidx;
                            x0 = (float)(int)T.qsum[row]; x1 = (float)(int)T.qsumsq[row];
                        }
                        s_x[(size_t)(j * 4) * VGL_AGG_ROW + lane] = x0;
                        s_x[(size_t)(j * 4 + 1) * VGL_AGG_ROW + lane] = x1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (k_c < 2 && ((pass_m >> j_c) & 1)) {
                        const float* mine_row = s_x + (size_t)lane * VGL_AGG_ROW;
#pragma unroll 16
                        for (int k = 0; k < 64; ++k) ch += mine_row[k];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            // chain (j, 0): fields 5 (pass 0) and 7 (later passes); chain (j, 1): fields 6 and 8 -- back to the site's lane through LDS
            s_x[lane] = ch_ref; s_x[64 + lane] = ch;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (walk) { v[4] = s_x[lane * 4]; v[5] = s_x[lane * 4 + 1]; v[6] = s_x[64 + lane * 4]; v[7] = s_x[64 + lane * 4 + 1]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (mine) {
#pragma unroll
            for (int k = 0; k < 16; ++k) T.i16[(size_t)ls * 16 + k] = v[k];
        }
    }
}

// ------------------------------------------------------------------------------------
// VGL_RNG_SERIAL, GL model 1, evaluations deeper than 255 reads: the reference's process calls errmod_cal() sample by sample,
// site by site (only for records that reach calculate_gls: status OK), and every call with n > 255 takes n - 1 draws of htslib's
// stream.  Exclusive prefix of those counts = where each evaluation's shuffle starts; the stream then advances by the total.
__global__ __launch_bounds__(1024) void k_hts_offsets(const VglDevParams P, const VglTilePtrs T, VglSerialState* S, long long* __restrict__ off, uint64_t* hts_base) {
    __shared__ long long s_w[16];
    __shared__ long long s_run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long n = (long long)T.n_sites * P.n_samples;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (long long base = 0; base < n; base += 1024) {
        const long long i = base + tid;
        long long v = 0;
        if (i < n && T.sinfo[i / P.n_samples].status == SITE_OK) {
            const uint64_t a = T.ad4[i];
            const long long d = (long long)((a & 0xFFFF) + ((a >> 16) & 0xFFFF) + ((a >> 32) & 0xFFFF) + ((a >> 48) & 0xFFFF));
            v = d > 255 ? d - 1 : 0;
        }
        long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        long long woff = 0;
        for (int k = 0; k < wv; k++) woff += s_w[k];
        const long long run = s_run;
        if (i < n) off[i] = run + woff + incl - v;
        __syncthreads();
        if (tid == 1023) s_run = run + woff + incl;
        __syncthreads();
    }
    if (tid == 0) { *hts_base = S->st_hts; S->st_hts = rand48_jump(S->st_hts, (uint64_t)s_run); }
}

extern "C" int vgl_launch_hts_offsets(const VglDevParams* p, const VglTilePtrs* t, VglSerialState* st, long long* hts_off, uint64_t* hts_base, void* stream) {
    if (t->n_sites == 0) return 0;
    hipLaunchKernelGGL(k_hts_offsets, dim3(1), dim3(1024), 0, (hipStream_t)stream, *p, *t, st, hts_off, hts_base);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_site(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    if (t->n_sites == 0) return 0;
    hipLaunchKernelGGL(k_site, dim3((unsigned)(((size_t)t->n_sites * (t->rowmap ? 16 : 1) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_gl(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    const int64_t waves = (int64_t)t->n_sites * p->chunks;
    if (waves == 0) return 0;
    if (waves + 8 >= (1LL << 31)) return (int)hipErrorInvalidValue;       // k_gl indexes the tile's wavefronts with 32 bits
    // 8 wavefronts (512 evaluations) per workgroup for GL model 2 at depth: the evaluations with three or four distinct bases, whose
    // loop is the expensive one, then fill one wavefront in 512 rather than one in 256 (P.gl_wpb: vgl_ctx_create, VGL_GL_WPB; measured
    // at C3 / fixed-q / C4: 4 -> 8 wavefronts -7 / -6 / -10 % of the kernel's time, 16 is slower again; equal at depth 5)
    // (GL model 1 with per-read scores, round 4: 8 wavefronts per workgroup measured 5.80 against 5.29 ms per C3-shaped launch -- its per-lane
    //  histograms take 4 KB of LDS per wavefront either way and the larger workgroup only adds barrier waiting; it keeps 4)
    if (p->gl2x == 2 || (p->gl2x && !t->gp && !t->fmt_ad && !t->fmt_adf && !t->fmt_adr)) {      // (vgl_ctx_create's choice; tiles with GP or FORMAT/AD* stay with k_gl -- 2: test hook, every tile)
        if (p->gl_model != 2 || p->precise_gl || !p->gl_sort || p->out_layout != 0 || !t->rowmap8 || !t->gl2_redo || !t->gl2_redo_list || !t->gl2_redo_count) return (int)hipErrorInvalidValue;
        const unsigned blocks2 = (unsigned)((waves + 15) / 16);
        hipStream_t s2 = (hipStream_t)stream;
        if (hipMemsetAsync(t->gl2_redo_count, 0, sizeof(uint32_t), s2) != hipSuccess) return (int)hipGetLastError();
        if (p->A == 5) hipLaunchKernelGGL((k_gl2<5>), dim3(blocks2), dim3(512), 0, s2, *p, *t);
        else hipLaunchKernelGGL((k_gl2<4>), dim3(blocks2), dim3(512), 0, s2, *p, *t);
        hipLaunchKernelGGL(k_gl2_scan, dim3(1), dim3(256), 0, s2, *t, (blocks2 + 31u) / 32u);
        if (p->A == 5) hipLaunchKernelGGL((k_gl_redo<5>), dim3(1024), dim3(512), 0, s2, *p, *t);
        else hipLaunchKernelGGL((k_gl_redo<4>), dim3(1024), dim3(512), 0, s2, *p, *t);
        return (int)hipGetLastError();
    }
    const int wpb = (p->gl_model == 2 && p->gl_wpb == 8) ? 8 : 4;
    const unsigned blocks = (unsigned)((waves + wpb - 1) / wpb);
    const size_t lds = 0;
    const dim3 g(blocks), b(64 * wpb);
    hipStream_t s = (hipStream_t)stream;
#define VGL_LAUNCH_GL(GLM, PREC, WPB) \
    do { if (p->A == 5) hipLaunchKernelGGL((k_gl<5, GLM, PREC, WPB>), g, b, lds, s, *p, *t); else hipLaunchKernelGGL((k_gl<4, GLM, PREC, WPB>), g, b, lds, s, *p, *t); } while (0)
    if (p->gl_model == 2 && p->precise_gl) { if (wpb == 8) VGL_LAUNCH_GL(2, true, 8); else VGL_LAUNCH_GL(2, true, 4); }
    else if (p->gl_model == 2) { if (wpb == 8) VGL_LAUNCH_GL(2, false, 8); else VGL_LAUNCH_GL(2, false, 4); }
    else VGL_LAUNCH_GL(1, false, 4);
#undef VGL_LAUNCH_GL
    return (int)hipGetLastError();
}

// the fused build (see k_gl): one workgroup of 256 / 512 threads per site (1024 threads for up to 1024 samples were measured: 2.07e10 against
// 2.34e10 evaluations/s for the three kernels at N = 1000 and C5's flags -- two workgroups per CU, sixteen wavefronts behind every barrier).
// The caller (vgl_host.cpp) checks the other conditions
extern "C" int vgl_launch_fused(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    if (t->n_sites == 0) return 0;
    const int S = p->fused_split;
    const int wg = p->n_samples <= 256 ? 256 : 512;
    if (S < 1 || (int64_t)S * wg < p->n_samples || p->read_cap > 128 || p->gl_model != 2 || p->precise_gl || p->error_qs != 0 || (p->depth_pre != 1 && p->depth_pre != 2))
        return (int)hipErrorInvalidValue;
    if ((int64_t)t->n_sites * S * 8 + 8 >= (1LL << 31)) return (int)hipErrorInvalidValue;
    const dim3 g((unsigned)((int64_t)t->n_sites * S));
    hipStream_t s = (hipStream_t)stream;
#define VGL_LAUNCH_FUSED(WPB, FW) \
    do { if (p->A == 5) hipLaunchKernelGGL((k_gl<5, 2, false, WPB, FW>), g, dim3(64 * WPB), 0, s, *p, *t); \
         else hipLaunchKernelGGL((k_gl<4, 2, false, WPB, FW>), g, dim3(64 * WPB), 0, s, *p, *t); } while (0)
    if (S == 1 && p->out_layout == 0 && !t->gp && !t->fmt_ad && !t->fmt_adf && !t->fmt_adr) {      // the plain build (k_gl_body)
        if (p->read_cap <= 64) { if (wg == 256) VGL_LAUNCH_FUSED(4, 4); else VGL_LAUNCH_FUSED(8, 4); }
        else { if (wg == 256) VGL_LAUNCH_FUSED(4, 8); else VGL_LAUNCH_FUSED(8, 8); }
    } else {                                                             // + 16: the general build
        if (p->read_cap <= 64) { if (wg == 256) VGL_LAUNCH_FUSED(4, 20); else VGL_LAUNCH_FUSED(8, 20); }
        else { if (wg == 256) VGL_LAUNCH_FUSED(4, 24); else VGL_LAUNCH_FUSED(8, 24); }
    }
#undef VGL_LAUNCH_FUSED
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// INFO/I16 fields 13-16 in VGL_RNG_TILE (vcfgl.cpp:647-663, 1029-1071): one tail distance per read of the site, min(1 + rand() / (RAND_MAX / 50 + 1), 25)
// (rng.h:12, CAP_TAIL_DIST), all credited to the base of the site's LAST simulated read (the reference's r_base is stale by then), summed and
// square-summed as float32 in (sample, read) order.  The reference's rand() is one serial never-seeded stream (VGL_RNG_SERIAL follows it, vgl_serial.hip);
// tile mode draws from a second rand48 sequence (VGL_TAIL_RAND48_X0) addressed like the first: evaluation e = H(site) N + sample owns draws
// [e block, (e + 1) block), read r takes draw r (include/vcfgl_hip.h) -- the site's state from k_sitebase, the sample's jump from P.samp_tab.
//   k_tail      one wavefront per (site, 64 samples): each lane its evaluation's draws, integer sums by DPP scans into the site's accumulators
//               (acc[9], acc[10..11], acc[12]: VGL_ACC_TAIL*), the last read's base from the staged reads.
//   k_tail_fin  one lane per site: while the sum of squares stays at or below 2^24 every partial sum of the reference's float32 chains is an integer the
//               format holds -- the chains' results ARE the integer totals; beyond that (more than ~40 000 reads at a site) the lane walks the site's reads
//               in the reference's order and adds in float32 as the reference does.
// T = the top 32 bits of the 48-bit state (lcg52_top32); the reference's 1 + rand() / (RAND_MAX / 50 + 1) on x = T >> 1, capped at CAP_TAIL_DIST
__device__ __forceinline__ uint32_t tail_dist_of(const uint32_t T) {
    const uint32_t q = (T >> 1) / 42949673u;                           // (a 31-bit dividend: one multiply-high and a shift)
    return q >= 24u ? 25u : q + 1u;
}
// the evaluation's draws: its window's state on the raw 52-bit form (x 16: lcg52_step, three instructions a step), sum and sum of squares of dp distances
__device__ __forceinline__ void tail_sums(const uint64_t st48, const int dp, uint32_t& s1, uint32_t& s2) {
    const uint64_t x = st48 << 4;
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    for (int r = 0; r < dp; ++r) {
        lcg52_step(lo, hi, lo, hi);
        const uint32_t td = tail_dist_of(lcg52_top32(lo, hi));
        s1 += td; s2 += td * td;                                       // (at most 1023 reads: 25 x 1023, 625 x 1023)
    }
}
__device__ __forceinline__ int dp_of_ad4(const uint64_t a) { return (int)((a & 0xFFFF) + ((a >> 16) & 0xFFFF) + ((a >> 32) & 0xFFFF) + ((a >> 48) & 0xFFFF)); }

__global__ __launch_bounds__(256) void k_tail(const VglDevParams P, const VglTilePtrs T) {
    const int lane = threadIdx.x & 63;
    const int N = P.n_samples;
    const int64_t w = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= (int64_t)T.n_sites * P.chunks) return;
    const int ls = (int)(w / P.chunks), s = (int)(w - (int64_t)ls * P.chunks) * 64 + lane;
    uint32_t s1 = 0, s2 = 0, key = 0;
    if (s < N) {
        const size_t ev = (size_t)ls * N + (size_t)s;
        const int dp = dp_of_ad4(T.ad4[ev]);
        if (dp > 0) {
            const size_t plane = (size_t)T.n_sites * N;
            const uint8_t last = T.reads[vgl_read_byte(dp - 1, plane, ev)];
            key = ((uint32_t)(s + 1) << 2) | (uint32_t)(last & 3u);
            tail_sums(aff(P.samp_tab[s], T.tail_base[ls]), dp, s1, s2);
        }
    }
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(s1), 63);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(s2), 63);      // (64 x 625 x 1023 < 2^32)
    uint32_t km = key;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)km, m, 64); km = o > km ? o : km; }
    if (lane == 0 && km) {
        int32_t* const acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
        atomicAdd((unsigned int*)&acc[VGL_ACC_TAIL], t1);
        atomicAdd((unsigned long long*)&acc[VGL_ACC_TAILSQ], (unsigned long long)t2);
        atomicMax((unsigned int*)&acc[VGL_ACC_TAILKEY], km);
    }
}

__global__ __launch_bounds__(64) void k_tail_fin(const VglDevParams P, const VglTilePtrs T) {
    const int ls = blockIdx.x * 64 + threadIdx.x;
    if (ls >= T.n_sites) return;
    const int N = P.n_samples;
    const int32_t* const acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
    const uint32_t t1 = (uint32_t)acc[VGL_ACC_TAIL], key = (uint32_t)acc[VGL_ACC_TAILKEY];
    const unsigned long long t2 = *(const unsigned long long*)&acc[VGL_ACC_TAILSQ];
    VglSiteTail t; t.sum = (float)t1; t.sumsq = (float)t2; t.base = key ? (int32_t)(key & 3u) : -1; t.pad = 0;
    if (t2 > (1ULL << 24)) {                                           // (t1 <= t2: every distance is at least 1)
        float f1 = 0.0f, f2 = 0.0f;
        const uint64_t xb = T.tail_base[ls];
        for (int s = 0; s < N; ++s) {
            const int dp = dp_of_ad4(T.ad4[(size_t)ls * N + (size_t)s]);
            const uint64_t x = aff(P.samp_tab[s], xb) << 4;
            uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
            for (int r = 0; r < dp; ++r) {
                lcg52_step(lo, hi, lo, hi);
                const uint32_t td = tail_dist_of(lcg52_top32(lo, hi));
                f1 += (float)td; f2 += (float)(td * td);
            }
        }
        t.sum = f1; t.sumsq = f2;
    }
    T.site_tail[ls] = t;
}

extern "C" int vgl_launch_tail(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    if (t->n_sites == 0 || !t->site_tail || !t->tail_base) return 0;
    const int64_t waves = (int64_t)t->n_sites * p->chunks;
    hipLaunchKernelGGL(k_tail, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *p, *t);
    hipLaunchKernelGGL(k_tail_fin, dim3((t->n_sites + 63) / 64), dim3(64), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_siteagg(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    if (t->n_sites == 0) return 0;
    hipLaunchKernelGGL(k_siteagg, dim3((t->n_sites + VGL_AGG_SITES - 1) / VGL_AGG_SITES), dim3(64), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}
