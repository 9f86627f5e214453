// vgl_sample.hip -- k_sample<EQS>: the per-read sampling of one tile in VGL_RNG_TILE mode
// (vcfgl.cpp:364-389, 441-640; rng.h).  See vgl_common.hip.h for the work decomposition.
//
// Reads are staged as 1 byte each in [read][site][sample] planes; per-site depth sums by wave
// reduction + one integer atomic per wave and counter.  EQS=2 (a beta deviate per read): the
// quality-score sampling of the wave's reads is an LDS-staged pool dealt dynamically to the lanes,
// each lane a select-only state machine over normal-deviate attempts.
#include <stdlib.h>
#include <type_traits>

#include "vgl_common.hip.h"

// ------------------------------------------------------------------------------------
// EQS = --error-qs.  EQS 2 (a beta deviate per read) runs the quality-score sampling as a
// wavefront-wide pool: the reads of the wave's 64 evaluations are independent work items
// (stream 3 is addressed per read), staged in LDS and dealt round-robin to the lanes, so a
// lane's work does not depend on its own evaluation's depth; each lane runs the nested
// rejection loops of the gamma sampler as one flat state machine (one normal-deviate attempt
// per iteration) so that lanes at different stages share every iteration.
// Where the windows of each site of the tile start in the rand48 sequence (VGL_RNG_TILE): one lane per site computes
// H(site) (vgl_site_hash, vgl_device.h) and the generator state J^(block N H(site)) (x0) in front of the site's N evaluation
// blocks from the 40-entry power table.  k_depth, k_sample and k_redo start from that state (one load per wavefront / lane);
// samples and reads stay additive inside a site.
#ifndef VGL_SAMPLE_SEG_TU
__global__ __launch_bounds__(256) void k_sitebase(const VglDevParams P, const VglTilePtrs T) {
    const int ls = blockIdx.x * 256 + threadIdx.x;
    if (ls >= T.n_sites) return;
    const uint64_t h = vgl_site_hash((uint64_t)(T.site0 + ls), P.site_hash_bits);
    uint64_t xb = P.x0;
#pragma unroll 1
    for (int b = 0; b < 40; ++b)
        if ((h >> b) & 1) xb = aff(P.site_pow[b], xb);
    T.site_base[ls] = xb;
    T.site_hash[ls] = h;
    if (T.tail_base) {                                                 // -addI16: the same windows of the tail distances' sequence (k_tail, vgl_gl.hip)
        uint64_t xt = VGL_TAIL_RAND48_X0;
#pragma unroll 1
        for (int b = 0; b < 40; ++b)
            if ((h >> b) & 1) xt = aff(P.site_pow[b], xt);
        T.tail_base[ls] = xt;
    }
}

// Depth draws of a tile (vcfgl.cpp:364-389; rng.h:284-351), ahead of k_sample.  The rejection sampler
// needs 2 attempts on average but about 7 for the slowest of 64 lanes, so evaluations are not tied to
// lanes here: a wavefront owns `chunk` consecutive evaluations of the tile (1024, 2048 or 4096: vgl_launch_depth) and deals them to its
// lanes as lanes finish; one rejection attempt of every busy lane per iteration (poisson_attempt: decided in float32), lane state advanced by selects
// (launched only when every sample's mean depth is >= 12; the short product-method loops stay in k_sample).  Evaluation
// (site, sample) starts its depth stream at J^(off0 + block sample) (site_base[site]): the site's state from k_sitebase, one table
// jump per evaluation.  A chunk spans sites: the site of item i is found from the chunk's first (site, sample) by one compare
// (N >= chunk) or one multiply-high (P.depth_magic).
// ZT: one mean depth for all samples, the acceptance bound's exponent from P.pois_zt
template <bool ZT>
__global__ __launch_bounds__(256) void k_depth(const VglDevParams P, const VglTilePtrs T, const int chunk) {
    const int lane = threadIdx.x & 63;
    const uint32_t N = (uint32_t)P.n_samples;
    const int64_t E = (int64_t)T.n_sites * N;
    const int64_t c0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * chunk;
    if (c0 >= E) return;
    const int cn = (int)((E - c0 < chunk) ? (E - c0) : chunk);
    const uint32_t ls_c0 = (uint32_t)(c0 / (int64_t)N);                  // site and sample of the chunk's first evaluation (wave-uniform)
    const uint32_t s_c0 = (uint32_t)(c0 - (int64_t)ls_c0 * N);
    const bool big_n = N >= (uint32_t)chunk;
    auto start_of = [&](const int i, uint32_t& s_out) -> uint64_t {      // stream start of the chunk's item i (i < cn)
        const uint32_t t = s_c0 + (uint32_t)i;                           // < N + chunk
        const uint32_t add = big_n ? (uint32_t)(t >= N) : (N == 1u ? t : __umulhi(t, P.depth_magic));   // t / N (the magic number needs N >= 2 to fit 32 bits)
        s_out = t - add * N;
        return aff(P.depth_tab[s_out], T.site_base[ls_c0 + add]);
    };

    int i = lane, next_free = 64;
    bool busy = i < cn;
    uint32_t s_i;
    uint64_t st = start_of(busy ? i : 0, s_i);
    VglPois pp = P.pois0;
    if (P.per_sample_depth) pp = P.pois[s_i];
    while (__ballot(busy)) {
        // one rejection attempt (rng.h:300-312); k_depth runs only when every sample uses this method
        const uint64_t st1 = lcg_next(st), st2 = lcg_next(st1);
        bool neg, reject; int result;
        poisson_attempt<ZT>(pp, st1, st2, busy, P.gamma_ln_tab, P.gamma_ln_n, P.pois_zt, neg, reject, result);
        st = neg ? st1 : st2;                                            // em < 0 consumes one draw, an attempt two
        const bool done = busy && !neg && !reject;
        if (done) T.dp_pre[c0 + i] = result;
        // lanes that finished take the next undealt evaluation of the chunk
        const uint64_t dm = __ballot(done);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
        const int i_n = done ? next_free + rank : i;
        next_free += __popcll(dm);
        const bool busy_n = done ? (i_n < cn) : busy;
        uint32_t s_n;
        const uint64_t st_n = start_of(busy_n ? i_n : 0, s_n);
        if (P.per_sample_depth) { const VglPois pn = P.pois[s_n]; if (done) pp = pn; }
        st = done ? st_n : st;
        i = i_n; busy = busy_n;
    }
}

#endif  // !VGL_SAMPLE_SEG_TU

// DBG: diagnostic instantiation (VGL_DEBUG_STAMPS / VGL_DEBUG_PHASE), never used in a timed run
// DM (depth mode): 1 = depths come from k_depth, 2 = every sample uses the product method (mean depth < 12), drawn
// in place by its short loop, 0 = mixed means, the general sampler in place (its float32-bounded rejection path costs
// the kernel registers, hence the specialisations)
// EQS 2 is built for 4 wavefronts per SIMD (128 VGPRs, 32 spilled into rarely executed fallback code): with 3 the VALU
// pipes were 86 % busy; the fourth wave hides the dependent f64 chains of the pool loop (+8 % on C3)
// PREC: --precise-gl 1 with EQS 2 (the exact error probability of every read is staged for k_gl)
// LEAN: EQS 2 without per-base quality sums (-addQS / -addI16), strand draws, a per-read dump or --adjust-qs: the default tag
// surface of the benchmark configurations.  The owners' state those options need (8 quality sums, the forward-strand depths, the
// dump pointers) is then not carried across the pool loop, where the 128-register build would park it in scratch.
// LEAN 2 = LEAN with the double-precision fallbacks deferred: a read whose quality score a float32 bound cannot settle (in the
// pool loop's two bounded tests or in the dense pass; about one read in 10^4) is appended to T.redo_list and drawn again in double
// by k_redo.  Without log() / pow() / the double gamma sampler the kernel needs 92 VGPRs and no scratch, and is built for 5
// wavefronts per SIMD (+6.5 % on C3).
// LEAN 3 (round 4) = the deferred build for the OPTIONAL tag surface (-addQS / -addI16, strand tags, --adjust-qs; no per-read dump): the
// owners' options are run-time flags as in LEAN 0, the fallbacks are k_redo's as in LEAN 2 -- k_redo then also adds the redrawn read's
// score to the evaluation's quality sums and to the site totals, which took a placeholder of 0.  The eight quality sums of an owner
// live only inside a segment's flush (stored, or added to what earlier segments stored, at its end) instead of across the pool loop.
// LEAN 0 keeps the inline fallbacks: per-read dumps, beta shapes below 8, VGL_NO_DEFER.
// SEG (round 6): how a wavefront whose reads do not fit one pool is served.  0: the segment loop in this kernel (every build but the float32 build
// of the default tag surface).  1 / 2: that build split in two kernels (k_sample_seg) -- 1 runs ONE segment and nothing else, a wavefront with more
// reads than the pool holds (8 sigma of the summed depth: never seen at the bench configurations) appends its index to T.seg_list and leaves; 2 is
// the segment loop, launched behind it over the listed wavefronts.  With the segment loop out of the way nothing of the owners' state (two stream
// states, thresholds, alleles, per-base depths, pointers) lives across the pool loop, which is what the 64-register build spilled: 76 -> 0 bytes of
// scratch per lane, 7.6 GB less traffic per 65536 x 1000 tile.
template <int EQS, bool DBG, int DM, bool PREC, int LEAN, int SEG = 0>
__device__ __forceinline__ void k_sample_body(const VglDevParams& P, const VglTilePtrs& T, const int64_t wave_index) {
    constexpr bool DEFER = (LEAN >= 2);
    // round 5: the deferred builds without --precise-gl 1 run their pool loop in float32 (vgl_common.hip.h, "the pool loop of k_sample<2>
    // in float32"); -DVGL_POOL_F64 keeps the float64 loop in every build (A/B timing)
#ifndef VGL_POOL_GROUPS
#define VGL_POOL_GROUPS 4
#endif
#ifndef VGL_POOL_ATTEMPTS
#define VGL_POOL_ATTEMPTS 2
#endif
#ifdef VGL_POOL_F64
    constexpr bool F32 = false;
#else
#ifdef VGL_PREC_F64
    constexpr bool F32 = DEFER && !PREC;                             // (A/B: --precise-gl 1 on the float64 loop, as until round 5)
#else
    // round 6: --precise-gl 1 too.  The DECISIONS of the rejection loops come from the float32 loop exactly as without the flag; what k_gl needs beyond
    // the score -- the read's error probability X / (X + Y) as the reference's double -- is evaluated in double for the ACCEPTED attempts only (PREC blocks
    // below: the attempt's two uniforms rebuilt from its generator states, v / u, w, a1 w^3 as rng.h:72-78,139-145 do), twice per read instead of once per attempt
    constexpr bool F32 = DEFER;
#endif
#endif
    // P16 (round 5): the float32 loop of the default tag surface keeps an item in TWO bytes -- owner << 10 | read << 2 | base until it is
    // finished, then the read's staged byte (score << 2 | base; bit 8: undecided, k_redo draws the read) written by the lane that finishes
    // it.  No base plane, no dense pass; a wavefront's pool of up to 2240 items (depth 30 in one segment) leaves LDS for eight wavefronts per SIMD
    constexpr bool P16 = F32;                                         // (LEAN 2 and LEAN 3 without --precise-gl 1)
    constexpr int ISZ = P16 ? 2 : 4;                                  // bytes of an item's slot
    constexpr bool SLIM = (LEAN == 1 || LEAN == 2);                  // default tag surface: none of the optional per-read state
    constexpr bool DUMP = (LEAN == 0);                               // a per-read dump (reads_out) may be asked for
#ifdef VGL_TEST_HOOKS
    const int dbg_redo_every = P.dbg_redo_every, dbg_qs_exact = P.dbg_qs_exact;      // test hooks (VGL_DEBUG_REDO_EVERY / VGL_DEBUG_QS_EXACT)
#else
    constexpr int dbg_redo_every = 0, dbg_qs_exact = 0;              // (the shipped library carries neither)
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const WavePos wp = wave_pos_of(P, T, wave_index);
    if (!wp.valid) return;
    const int lane = threadIdx.x & 63;
    const int N = P.n_samples;
    const int ls = wp.ls;
    const int s = wp.chunk * 64 + lane;
    const bool active = s < N;
    const size_t ev0 = (size_t)ls * N + (size_t)wp.chunk * 64;      // evaluation of lane 0
    size_t ev = ev0 + (active ? lane : 0);                          // (SEG 1 forms it again behind the pool loop instead of carrying two registers across it)
    const size_t plane = (size_t)T.n_sites * N;

    int dp = 0, a0 = 0, a1 = 0;
    int wave_total_reads = 0;                                        // EQS 2: reads of the wavefront's evaluations (wave-uniform)
    uint64_t ad4 = 0, adf4 = 0;
    uint32_t qs0 = 0, qs1 = 0, qs2 = 0, qs3 = 0, qq0 = 0, qq1 = 0, qq2 = 0, qq3 = 0;
    uint64_t st_hap = 0, st_base = 0, st_qs = 0;
    // DBG only: per-phase cycle stamps
    unsigned long long c_t0 = 0, c_pois = 0, c_owner = 0, c_pool = 0, c_flush = 0, c_iter = 0, c_items = 0, c_tmp = 0;
    unsigned long long c_ntest = 0, c_gtest = 0, c_finb = 0, c_lhave = 0, c_lfin = 0, c_lhold = 0;   // float32 pool loop: block executions and lane counts per iteration
    if (DBG) c_t0 = clock64();
    uint64_t err_thresh = P.err_thresh;
    // LEAN 4 (round 5, second session) = LEAN 3 for the common optional-tag surface -- strand draws AND quality sums, no --adjust-qs (-addQS / -addI16 with
    // the strand tags: alltags, qsi16) -- with the three options as constants: 9.85 -> 9.40 ms per 65536 x 1000 tile (vgl_launch_sample picks it)
    const bool k_strand = SLIM ? false : (LEAN == 4 ? true : (P.sample_strand != 0));
    const bool k_qsum = SLIM ? false : (LEAN == 4 ? true : (P.need_qsum != 0));
    const int k_adj = (SLIM || LEAN == 4) ? 0 : P.adjust_qs;

    // ---- stream states of this evaluation: J^(off_k) . J^(block*s) . J^(block*N*H(site)) (x0); the site factor comes from k_sitebase
    const uint64_t xb = T.site_base[ls];

    if (active) {
        const VglAffine ms = P.samp_tab[s];
        const uint64_t xe = aff(ms, xb);
        st_hap = aff(P.off[1], xe);
        st_base = aff(P.off[2], xe);
        st_qs = aff(P.off[3], xe);

        // ---- depth (vcfgl.cpp:364-389): drawn even when the genotype is missing; by k_depth where the
        //      rejection method applies, here for the short loops of the product method
        int n;
        if (DM == 1) n = T.dp_pre[ev];
        else {
            uint64_t st_depth = aff(P.off[0], xe);
            VglPois pc = P.pois0;
            if (P.per_sample_depth) pc = P.pois[s];
            if (DM == 2) {                                               // rng.h:289-299
                // the product method's loop on a state carried shifted by 4 (lcg_next52r / bits_1xxx_52r: no mask and no 64-bit shift
                // per uniform, the same 48 bits in the mantissa as u01())
                double t = 1.0;
                int em = -1;
                uint64_t s52 = st_depth << 4;
                uint32_t k3ff = 0x3FF00000u;
                asm volatile("" : "+v"(k3ff));
                do { ++em; s52 = lcg_next52r(s52); t *= bits_1xxx_52r(s52, k3ff) - 1.0; } while (t > pc.g);
                n = em;
            } else n = poisson_draw_fast(pc, st_depth, P.gamma_ln_tab, P.gamma_ln_n);
        }
        const uint32_t g = T.gt[ev];
        a0 = g & 0xF; a1 = (g >> 4) & 0xF;
        dp = (a0 == 0xF || a1 == 0xF) ? 0 : n;
        if (dp > P.read_cap) { atomicOr(T.errflag, VGL_DEVERR_CAPACITY); dp = P.read_cap; }
#ifdef VGL_EXP_FIXED_DP
        if (EQS != 2 && dp > 0) dp = VGL_EXP_FIXED_DP;                // (timing experiment, round 6: no spread of the 64 depths of a wavefront = the most lane pairing could give)
#endif
    }

    if (DBG) c_pois = clock64() - c_t0;
    if (DBG && P.dbg_phase == 1) return;
    if (EQS == 1) {
        // one beta deviate per site: stream 3 of sample 0, read 0 (vcfgl.cpp:425-437); lane 0 draws it
        uint32_t lo = 0, hi = 0;
        if (lane == 0) {
            uint64_t st_site = aff(P.off[3], aff(P.samp_tab[0], xb));
            const double pe = beta_draw(P, st_site);
            if (T.site_pick_err && wp.chunk == 0) T.site_pick_err[ls] = pe;
            const uint64_t th = (uint64_t)ceil(ldexp(pe, 48));
            lo = (uint32_t)th; hi = (uint32_t)(th >> 32);
        }
        lo = __shfl(lo, 0, 64); hi = __shfl(hi, 0, 64);
        err_thresh = ((uint64_t)hi << 32) | lo;
    }

    if (EQS != 2) {
        // ---- read loop (vcfgl.cpp:469-613), fixed quality score
        const int q_i = P.pre_q, aq_i = P.pre_adjq;
        const uint32_t q_gl = (uint32_t)((P.adjust_qs & 1) ? aq_i : q_i);
        const uint32_t qq = (uint32_t)((P.adjust_qs & 2) ? aq_i : q_i);
        const uint32_t q2 = (uint32_t)qs_to_qssq((int)qq);
        const bool stage = P.stage_fixed != 0;                // GL model 1 with one fixed qScore needs only the per-base depths, unless an evaluation may exceed
                                                              // 255 reads (k_gl then subsamples the staged reads) or k_tail wants the last read's base (I16, tile mode)
        uint64_t st_hap16 = st_hap << 16, st_base16 = st_base << 16;      // sample_read_base16: states carried shifted by 16
        const uint64_t err_thresh16 = sample_thresh16(err_thresh);
        // four reads per trip: one 32-bit store of the staged word (vgl_read_byte) instead of four byte stores
        uint32_t* const reads_w = (uint32_t*)T.reads;
        if (SLIM) {
            // default tag surface (round 4): sample_reads_fixed -- the trip's haplotype picks as four bits, errors patched in the rare branch,
            // per-base depths from the pick count; a wavefront of homozygous evaluations does not step the haplotype stream at all
            const uint32_t qrep = (q_gl << 2) * 0x01010101u;
            uint32_t* const col = reads_w + ev;
            auto emit = [&](const int trip, const uint32_t bases) { if (stage) col[(size_t)trip * plane] = staged_word_of(bases, qrep); };
#ifdef VGL_FIXED_READS_V1
            const bool narrow = false;
#else
            const bool narrow = P.read_cap <= 255;                       // (8-bit depth fields: sample_reads_fixed; deeper runs keep the 16-bit form)
#endif
            if (narrow) {
                if (__ballot(active && dp > 0 && a0 != a1) == 0) ad4 = sample_reads_fixed<true>(st_hap16, st_base16, a0, a1, dp, err_thresh16, emit);
                else ad4 = sample_reads_fixed<false>(st_hap16, st_base16, a0, a1, dp, err_thresh16, emit);
            } else {
                if (__ballot(active && dp > 0 && a0 != a1) == 0) ad4 = sample_reads_fixed_wide<true>(st_hap16, st_base16, a0, a1, dp, err_thresh16, emit);
                else ad4 = sample_reads_fixed_wide<false>(st_hap16, st_base16, a0, a1, dp, err_thresh16, emit);
            }
        } else
        for (int r0 = 0; r0 < dp; r0 += 4) {
            uint32_t rw = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + j;
                if (r < dp) {
                    bool fwd;
                    const int r_base = sample_read_base16(st_hap16, st_base16, a0, a1, err_thresh16, k_strand, fwd);
                    rw |= ((q_gl << 2) | (uint32_t)r_base) << (8 * j);
                    if (DUMP) { if (T.reads_out && r < T.reads_out_cap) T.reads_out[(size_t)r * plane + ev] = (uint8_t)((q_i << 2) | r_base); }
                    const uint64_t one = 1ULL << (16 * r_base);
                    ad4 += one;
                    if (k_strand) { if (fwd) adf4 += one; }               // without strand draws adf4 = ad4 (set after the loops)
                }
            }
            if (stage) reads_w[(size_t)(r0 >> 2) * plane + ev] = rw;
        }
        if (P.need_qsum) {
            qs0 = qq * (uint32_t)(ad4 & 0xFFFF); qs1 = qq * (uint32_t)((ad4 >> 16) & 0xFFFF);
            qs2 = qq * (uint32_t)((ad4 >> 32) & 0xFFFF); qs3 = qq * (uint32_t)((ad4 >> 48) & 0xFFFF);
            qq0 = q2 * (uint32_t)(ad4 & 0xFFFF); qq1 = q2 * (uint32_t)((ad4 >> 16) & 0xFFFF);
            qq2 = q2 * (uint32_t)((ad4 >> 32) & 0xFFFF); qq3 = q2 * (uint32_t)((ad4 >> 48) & 0xFFFF);
        }
    } else {
        // ---- LDS of this wave: [64] u64 qscore-stream bases | [2][4] f64 gamma constants | [cap + 2] u32 item slot | [cap] u8 base.
        //      An item's slot holds (8 owner) << 16 | 16 read -- the LDS address of the owner's stream base and the byte offset of the read's
        //      jump in qs_read_tab, each one 16-bit LDS read away (round 3: owner << 26 | read << 4, a shift and a mask per iteration) --
        //      until the item is finished, then the float32 error
        //      probability of its read, then (dense pass) qScore | adjusted qScore << 8.  Slot [segT] of a segment is 0: the
        //      prefetch of a lane that has no next item reads it.  Slot [cap + 1]: the pool loop's item counter.
        const int cap = P.pool_cap;
        uint8_t* wl = lds_raw;                         // one wavefront per workgroup (vgl_launch_sample): LDS offsets are compile-time
        uint64_t* l_stq = (uint64_t*)wl;
        // a1, a2, 0.15 a2^2 of the two gamma samplers and the sure-accept margin, looked up by a lane's stage in the pool loop (LDS
        // reads instead of selects and multiplications per iteration, and fewer register pairs to carry)
        double* l_gc = (double*)(wl + 512);
        uint32_t* l_it = (uint32_t*)(wl + 576);
        uint8_t* l_pb = wl + 576 + 4 * ((size_t)cap + 2);
        uint32_t* l_ctr = l_it + cap + 1;             // pool loop: first unclaimed item
        // LEAN 3, P.qsum_lds: the owners' per-base quality sums are gathered by the dense pass with one LDS atomic per read, into 32-bit
        // words (sum of squares << 13 | sum: at most 130 reads of a score <= 63 each, 130 x 63 = 8190 < 2^13) -- bases A, C of owner o at l_stq[o] (free between
        // the pool loop and the next segment), bases G, T in 512 bytes behind the pool
        const bool qfast = (LEAN >= 3) && (P.qsum_lds != 0);
        uint32_t* l_accB = (uint32_t*)(wl + ((576 + 4 * ((size_t)cap + 2) + (size_t)cap + 7) & ~(size_t)7));
        // the pool loop reads l_stq by LDS byte offsets taken from the item slots: the dynamic LDS block must start at 0
        // (this kernel has no static LDS)
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds_raw != 0u) {      // (vgl_launch_sample checks the same on the host)
            if (lane == 0) atomicOr(T.errflag, VGL_DEVERR_INTERNAL);
            return;
        }

        // exclusive prefix sum of the depths = first pool index of each owner
        const int incl = (int)wave_incl_scan_u32((uint32_t)dp);
        const int offs = incl - dp;
        const int total = __builtin_amdgcn_readlane(incl, 63);          // wave-uniform, and known to the compiler as such
        wave_total_reads = total;
        l_stq[lane] = st_qs << 4;                      // the pool loop works on states scaled by 16 (lcg_next52)
        // P16 with --qs-bins: score -> binned score (vcfgl.cpp:57-64) as a 256-byte table behind the items (0xFF: in no bin); the lane that
        // finishes a read looks its score up there -- the bins' search loop ran per finishing lane and iteration (depth 30, rta3 bins: 44.8 ms
        // against 35.6 for the dense-pass build; with the table 2 x 2 instructions)
        // P16 with the LDS quality sums (qfast): 64 owners x 4 bases x one 32-bit word (sum of squares << 13 | sum) behind the items, added to by
        // the lane that finishes a read, read and cleared by the owner when it stages its reads
        uint8_t* const l_lut = wl + ((576 + 2 * ((size_t)cap + 2) + 7) & ~(size_t)7);
        uint32_t* const l_qs = (uint32_t*)(l_lut + (P.n_qs_bins != 0 ? 256 : 0));
        // PREC with the float32 loop: a1, a2 of the first and of the second gamma sampler as doubles, 32 bytes behind the sum words (vgl_launch_sample adds them)
        double* const l_gcd = (double*)((uint8_t*)l_qs + ((LEAN >= 3 && P.qsum_lds) ? 1024 : 0));
        if (F32 && PREC) { if (lane == 0) { l_gcd[0] = P.gx.a1; l_gcd[1] = P.gx.a2; l_gcd[2] = P.gy.a1; l_gcd[3] = P.gy.a2; } }
        if (P16 && P.n_qs_bins != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qv = lane * 4 + j;
                int b = 0xFF;
                for (int i = 0; i < P.n_qs_bins; ++i) if (b == 0xFF && qv >= P.qs_bins[3 * i] && qv <= P.qs_bins[3 * i + 1]) b = P.qs_bins[3 * i + 2] & 0xFF;
                l_lut[qv] = (uint8_t)b;
            }
        }
        if (P16 && LEAN >= 3) { if (P.qsum_lds) { ((uint64_t*)l_qs)[2 * lane] = 0ULL; ((uint64_t*)l_qs)[2 * lane + 1] = 0ULL; } }
        if (lane == 0) {
            if (!P16) l_it[cap] = 0u;
            if (F32) {
                // a1, a2, the sure-accept bound's coefficient and margin in units of 2^-32, with the float32 error of either side
                // added (vgl_common.hip.h): 1 - u2 >= 0.15 a2^2 x^4 + margin holds whenever the float32 comparison says so
                float* const gf = (float*)l_gc;
                gf[0] = (float)P.gx.a1; gf[1] = (float)P.gx.a2; gf[2] = (float)((((P.gx.a2 * P.gx.a2) * 0.15) * (1.0 + 1e-5) + 1e-8) * 4294967296.0); gf[3] = (float)((P.sure_margin + 3e-7) * 4294967296.0);
                gf[4] = (float)P.gy.a1; gf[5] = (float)P.gy.a2; gf[6] = (float)((((P.gy.a2 * P.gy.a2) * 0.15) * (1.0 + 1e-5) + 1e-8) * 4294967296.0); gf[7] = (float)((P.sure_margin + 3e-7) * 4294967296.0);
            } else {
            l_gc[0] = P.gx.a1; l_gc[1] = P.gx.a2; l_gc[2] = (P.gx.a2 * P.gx.a2) * 0.15; l_gc[3] = P.sure_margin;
            l_gc[4] = P.gy.a1; l_gc[5] = P.gy.a2; l_gc[6] = (P.gy.a2 * P.gy.a2) * 0.15; l_gc[7] = P.sure_margin;
            }
        }
        int rdone = 0;
        uint32_t carry_w = 0;                          // the staged word the last segment ended in
        uint64_t st_hap16 = st_hap << 16, st_base16 = st_base << 16;      // sample_read_base16
        const uint64_t err_thresh16 = sample_thresh16(err_thresh);
        // Kernel arguments arrive in 16-dword scalar tuples that the register allocator spills and
        // reloads as a whole (v_readlane, VALU work) inside the loops below; the three values the flush
        // loop needs are therefore pinned to vector registers.
        uint8_t* reads_v = T.reads;
        uint8_t* reads_out_v = DUMP ? T.reads_out : nullptr;
        int reads_out_cap_v = (DUMP && T.reads_out) ? T.reads_out_cap : 0;
        if (!DUMP) { if constexpr (SEG != 1) asm volatile("" : "+v"(reads_v)); }
        else asm volatile("" : "+v"(reads_v), "+v"(reads_out_v), "+v"(reads_out_cap_v));

        const bool homw = __ballot(active && dp > 0 && a0 != a1) == 0;   // every evaluation of the wavefront with reads is homozygous
        // the pool holds `cap` items; a wavefront with more reads works through segments of equal length
        const int nseg = (total + cap - 1) / (cap > 0 ? cap : 1);
        const int seglen = nseg > 1 ? (total + nseg - 1) / nseg : cap;
        // One segment is the rule (a pool holds 2240 items: depth 30).  For LEAN 3 the body is instantiated twice -- SINGLE: nothing of the
        // owners' state (stream states, thresholds, alleles) lives across the pool loop, which is what the register allocator spilled (round 5)
        auto run_segment = [&](auto single_tag, const int seg0) -> bool {     // true: a diagnostic phase ends the kernel here
            constexpr bool SINGLE = decltype(single_tag)::value;
            (void)SINGLE;
            const int segT = (total - seg0 < seglen) ? (total - seg0) : seglen;
            // -- owners: bases of their reads that fall into this segment
            if (DBG) c_tmp = clock64();
            int r_end = seg0 + segT - offs; r_end = r_end > dp ? dp : r_end; r_end = r_end < rdone ? rdone : r_end;
            {
                // slot value ((8 owner) << 16 | 16 read), slot address and base address advance by constants per read
                typedef __attribute__((address_space(3))) uint32_t lds_u32o;
                typedef __attribute__((address_space(3))) uint8_t lds_u8o;
                typedef __attribute__((address_space(3))) uint16_t lds_u16o;
                const uint32_t lane4 = (uint32_t)lane << 2;
                // (P16: owner << 10 | read << 2, advancing by 4; the base goes into bits 0-1)
                uint32_t sv = P16 ? (((uint32_t)lane << 10) | ((uint32_t)rdone << 2)) : (((uint32_t)lane << 19) | ((uint32_t)rdone << 4));
                const uint32_t sv_end = P16 ? (((uint32_t)lane << 10) | ((uint32_t)r_end << 2)) : (((uint32_t)lane << 19) | ((uint32_t)r_end << 4));
                uint32_t ka = 576u + (uint32_t)ISZ * (uint32_t)(offs + rdone - seg0);            // l_it[k]  (the dynamic LDS block starts at 0)
                uint32_t pa = 576u + 4u * ((uint32_t)cap + 2u) + (uint32_t)(offs + rdone - seg0);   // l_pb[k]
                auto owner_reads = [&](auto hom_tag) {
                    constexpr bool HOMW = decltype(hom_tag)::value;
                    if constexpr (SLIM) {
                        // default tag surface: the per-base depths from the number of second-haplotype picks and the errors' corrections
                        // (as sample_reads_fixed) instead of a 64-bit shift-and-add per read
                        uint32_t n1 = 0;
                        const uint32_t nrd = (sv_end > sv) ? ((sv_end - sv) >> (P16 ? 2 : 4)) : 0u;
#ifdef VGL_OWNER_READS_V2     // (built and measured in round 6: +1.5 % on C3 / C4 -- the loop is 5 % of the kernel and the change costs the pool loop its register allocation; not used)
                        // the two states on their raw 52-bit form for the loop (lcg52_step: three instructions a step; pick = bit 19 of the high word, a
                        // wrong base its bits 18-19, the error test first on the high 20 bits and exactly inside the rare block -- as sample_reads_fixed)
                        uint32_t hl = (uint32_t)(st_hap16 >> 12), hh = (uint32_t)(st_hap16 >> 44), bl = (uint32_t)(st_base16 >> 12), bh = (uint32_t)(st_base16 >> 44);
                        const uint32_t t_lo = (uint32_t)(err_thresh16 >> 12), t_hi = (uint32_t)(err_thresh16 >> 44);
                        while (sv < sv_end) {
                            uint32_t rb = (uint32_t)a0;
                            if (!HOMW) {
                                lcg52_step(hl, hh, hl, hh);
                                const uint32_t h = (hh >> 19) & 1u;                       // u >= 0.5: the second allele (vcfgl.cpp:473)
                                n1 += h;
                                rb = h ? (uint32_t)a1 : (uint32_t)a0;
                            }
                            lcg52_step(bl, bh, bl, bh);
                            if (__builtin_expect((bh & 0xFFFFFu) <= t_hi, 0)) {
                                if ((bh & 0xFFFFFu) < t_hi || bl < t_lo) {                  // vcfgl.cpp:486-488
                                    const uint32_t tb = rb;
                                    do { lcg52_step(bl, bh, bl, bh); rb = (bh >> 18) & 3u; } while (rb == tb);
                                    ad4 += (1ULL << (16 * rb)) - (1ULL << (16 * tb));
                                }
                            }
                            if (P16) { *(lds_u16o*)(uintptr_t)ka = (uint16_t)(sv | rb); sv += 4u; ka += 2u; }
                            else { *(lds_u32o*)(uintptr_t)ka = sv; *(lds_u8o*)(uintptr_t)pa = (uint8_t)rb; sv += 16u; ka += 4u; pa += 1u; }
                        }
                        if constexpr (SEG != 1) {                                         // (another segment may follow: the states back on the form the rest of the kernel carries)
                            st_hap16 = ((((uint64_t)(hh & 0xFFFFFu)) << 32) | hl) << 12;
                            st_base16 = ((((uint64_t)(bh & 0xFFFFFu)) << 32) | bl) << 12;
                        }
#else
                        while (sv < sv_end) {
                            uint32_t rb = (uint32_t)a0;
                            if (!HOMW) {
                                st_hap16 = lcg_next16(st_hap16);
                                const uint32_t h = (uint32_t)(st_hap16 >> 63);          // u >= 0.5: the second allele (vcfgl.cpp:473)
                                n1 += h;
                                rb = h ? (uint32_t)a1 : (uint32_t)a0;
                            }
                            st_base16 = lcg_next16(st_base16);
                            if (st_base16 < err_thresh16) {                               // vcfgl.cpp:486-488
                                const uint32_t tb = rb;
                                do { st_base16 = lcg_next16(st_base16); rb = (uint32_t)(st_base16 >> 62); } while (rb == tb);
                                ad4 += (1ULL << (16 * rb)) - (1ULL << (16 * tb));
                            }
                            if (P16) { *(lds_u16o*)(uintptr_t)ka = (uint16_t)(sv | rb); sv += 4u; ka += 2u; }
                            else { *(lds_u32o*)(uintptr_t)ka = sv; *(lds_u8o*)(uintptr_t)pa = (uint8_t)rb; sv += 16u; ka += 4u; pa += 1u; }
                        }
#endif
                        ad4 += (((uint64_t)(nrd - n1)) << (16 * (a0 & 3))) + (((uint64_t)n1) << (16 * (a1 & 3)));
                    } else if constexpr (P16) {
                        // LEAN 3, float32 build: the same counting with the strand draws (vcfgl.cpp:581-586: one more output of the base
                        // stream per read, after the wrong-base draws) -- forward reads and forward second-haplotype picks counted, the
                        // errors' corrections applied to both depth words
                        uint32_t n1 = 0, nf = 0, nf1 = 0;
                        const uint32_t nrd = (sv_end > sv) ? ((sv_end - sv) >> 2) : 0u;
                        while (sv < sv_end) {
                            uint32_t rb = (uint32_t)a0, h = 0;
                            if (!HOMW) {
                                st_hap16 = lcg_next16(st_hap16);
                                h = (uint32_t)(st_hap16 >> 63);                         // u >= 0.5: the second allele (vcfgl.cpp:473)
                                n1 += h;
                                rb = h ? (uint32_t)a1 : (uint32_t)a0;
                            }
                            const uint32_t tb = rb;
                            st_base16 = lcg_next16(st_base16);
                            const bool err = st_base16 < err_thresh16;                    // vcfgl.cpp:486-488
                            if (err) { do { st_base16 = lcg_next16(st_base16); rb = (uint32_t)(st_base16 >> 62); } while (rb == tb); }
                            uint32_t f = 1u;
                            if (k_strand) { st_base16 = lcg_next16(st_base16); f = (uint32_t)(~st_base16 >> 63); }   // forward: u < 0.5
                            nf += f; nf1 += f & h;
                            if (err) {
                                const uint64_t d = (1ULL << (16 * rb)) - (1ULL << (16 * tb));
                                ad4 += d;
                                if (f) adf4 += d;
                            }
                            *(lds_u16o*)(uintptr_t)ka = (uint16_t)(sv | rb); sv += 4u; ka += 2u;
                        }
                        ad4 += (((uint64_t)(nrd - n1)) << (16 * (a0 & 3))) + (((uint64_t)n1) << (16 * (a1 & 3)));
                        adf4 += (((uint64_t)(nf - nf1)) << (16 * (a0 & 3))) + (((uint64_t)nf1) << (16 * (a1 & 3)));
                    } else
                    while (sv < sv_end) {
                        bool fwd;
                        const int r_base = sample_read_base16<HOMW>(st_hap16, st_base16, a0, a1, err_thresh16, k_strand, fwd);
                        const uint64_t one = 1ULL << (16 * r_base);
                        ad4 += one;
                        if (!SLIM) { if (fwd) adf4 += one; }
                        if (P16) { *(lds_u16o*)(uintptr_t)ka = (uint16_t)(sv | (uint32_t)r_base); sv += 4u; ka += 2u; }
                        else {
                            *(lds_u32o*)(uintptr_t)ka = sv;
                            *(lds_u8o*)(uintptr_t)pa = (uint8_t)((LEAN >= 3) ? (lane4 | (uint32_t)r_base) : (uint32_t)r_base);   // (LEAN 3: the dense pass finds the item's owner here)
                            sv += 16u; ka += 4u; pa += 1u;
                        }
                    }
                };
                // a wavefront of homozygous evaluations (most wavefronts of a rare variant's site) does not step the haplotype stream
                if (homw) owner_reads(std::true_type{}); else owner_reads(std::false_type{});
            }
            if (lane == 0) {                                          // the "no item" slot of this segment's prefetches; first unclaimed item
                if (P16) ((uint16_t*)l_it)[segT] = 0; else { l_it[segT] = 0u; *l_ctr = 128u * 4u; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // -- pool: item k -> lane k % 64.  One iteration = one normal-deviate attempt
            //    (rng.h:72-78) + the gamma step it feeds (rng.h:139-145), computed for every lane
            //    without branches; lane state (stream, stage, first gamma) advances by selects, so
            //    lanes in different stages of different reads share every instruction.  Real
            //    branches remain only around the rare bounded-log tests and the per-read epilogue.
            if (DBG) { const unsigned long long c = clock64(); c_owner += c - c_tmp; c_tmp = c; c_items += segT; }
            if (DBG && P.dbg_phase == 2) return true;
            {
                // Items are dealt dynamically: a lane owns its current item k and an item kn claimed one
                // item ahead (so that kn's operands are in flight while k is worked on); a lane that
                // finishes an item adopts kn and claims the next unclaimed one.  Each item's stream is
                // addressed by its read, so the result does not depend on who works on it.
                // k, kn: the lane's item and the one claimed ahead, as BYTE offsets of their slots (4 x item index: the counter is
                // bumped by 4, so a claim is an LDS address without a shift)
                typedef __attribute__((address_space(3))) uint32_t lds_u32;
                typedef __attribute__((address_space(3))) uint16_t lds_u16;
                // float32 loop (round 5): FOUR counters instead of one -- lanes 16 g .. 16 g + 15 deal the g-th quarter of the segment's items
                // among themselves.  Every finishing lane's ds_add_rtn went to ONE LDS word, and the LDS unit serialises same-address
                // atomics: with the loop's vector work cut by a third that queue, not instruction issue, set the kernel's time.  Four words
                // in four banks run side by side; the quarters end within an iteration or two of one another (320 items over 16 lanes each)
                constexpr int NGRP = F32 ? VGL_POOL_GROUPS : 1;
                const int grpQ = (segT + NGRP - 1) / NGRP;
                const int grp = (NGRP > 1) ? (lane / (64 / NGRP)) : 0, gl = (NGRP > 1) ? (lane % (64 / NGRP)) : lane;
                const int beg = grp * grpQ;
                const int endg = (beg + grpQ < segT) ? (beg + grpQ) : segT;
                const int segT4 = (NGRP > 1 || P16) ? endg * ISZ : segT * ISZ;      // this lane's limit (wave-uniform with one group); byte offsets: ISZ per item
                int k = (beg + gl) * ISZ, kn = (beg + gl + 64 / NGRP) * ISZ;
                bool have = k < segT4;                       // == (k < segT4) throughout: the loop tests that compare
                bool stage1 = false;                         // false: first gamma deviate (x), true: second (y)
                uint64_t st = 0; double gx = 0.0; uint32_t it_m = 0;
                uint32_t sk = 0;                             // P16: the slot of the lane's current item (its base is wanted when the item is finished)
                if (have) {
                    if (P16) { sk = ((const uint16_t*)l_it)[beg + gl]; st = aff52(P.qs_read_tab[(sk >> 2) & 0xFFu], l_stq[sk >> 10]); }
                    else { it_m = l_it[beg + gl]; st = aff52(P.qs_read_tab[(it_m & 0xFFFFu) >> 4], l_stq[it_m >> 19]); }
                }
                if (NGRP > 1 || P16) {
                    // the groups' counters: LDS bytes 544 .. 575 (behind the float32 loop's eight constants), first unclaimed item of each
                    if (gl == 0) *(lds_u32*)(uintptr_t)(544u + 4u * (uint32_t)grp) = (uint32_t)(beg + 2 * (64 / NGRP)) * (uint32_t)ISZ;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                const bool any_changed = (P.gx.changed | P.gy.changed) != 0;
                // The bounded-log tests are needed by a few lanes per iteration but cost every lane of the
                // wave; they run only every P.slow_period-th iteration.  In between, a lane that needs one
                // holds: its state is left untouched, so the later iteration recomputes the same attempt.
                int slow_cnt = P.slow_period, slow_cnt_n = P.slow_period_n;
                uint32_t k3ff = 0x3FF00000u;                 // the exponent word of 1.xxx, in a vector register for v_and_or_b32
                // a lane that finishes an item claims the next one with an LDS atomic whose result is first looked at in the next
                // iteration, so no wait stands behind the atomic; the atomic (issued with only the finishing lanes enabled) returns into
                // kn's own register -- the other lanes keep theirs without a select
                // address of the counter and the increment live in vector registers across the loop (the compiler would otherwise
                // rebuild both with two moves in front of every atomic)
                __attribute__((address_space(3))) uint32_t* ctr_p = (NGRP > 1 || P16) ? (__attribute__((address_space(3))) uint32_t*)(uintptr_t)(544u + 4u * (uint32_t)grp)
                                                                                : (__attribute__((address_space(3))) uint32_t*)l_ctr;
                uint32_t four_v = (uint32_t)ISZ;
                asm volatile("" : "+v"(ctr_p), "+v"(four_v));
                if constexpr (F32) {
                    // ---- float32 loop (vgl_common.hip.h): same dealing, same hold / period logic; every quantity a float32 built from the
                    // integer state, every comparison against a threshold moved by the float32 error bound, `redo` where a bound cannot tell.
                    // TWO normal-deviate attempts per iteration (VGL_POOL_ATTEMPTS 2): the ratio-of-uniforms sampler rejects 27 % of its
                    // attempts, and an iteration whose attempt is rejected has paid for the gamma step, the dealing and the loop's scalar
                    // bookkeeping for nothing -- measured (round 5): extra vector instructions in this loop cost next to nothing, an
                    // iteration does.  Attempt B (the generator's next two outputs) is looked at only when attempt A is rejected outright; the
                    // gamma step takes whichever was accepted: 1.08 iterations per gamma deviate instead of 1.37.
                    typedef __attribute__((address_space(3))) float lds_f32;
                    uint32_t s_lo = (uint32_t)st, s_hi = (uint32_t)(st >> 32);
                    float gxf = 0.0f;
                    double gxd = 0.0;                                           // PREC: the first gamma deviate of the lane's read, in double
                    do {
                        if (DBG) { c_iter++; c_lhave += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(have)); }
                        // ONE period for both bounded tests: a lane held for the normal attempt's logarithm test AND for the gamma test advances only
                        // in an iteration that runs both; with two counters (VGL_SLOW_PERIOD != VGL_SLOW_PERIOD_N under the hooks) that was every lcm of the periods,
                        // all 64 lanes spinning meanwhile.  P.slow_period_n is the float64 loop's alone now
                        const bool full = (--slow_cnt == 0);
                        if (full) slow_cnt = P.slow_period;
                        const bool full_n = full;
                        uint32_t goff = stage1 ? 528u : 512u;                       // this stage's four constants: LDS bytes 512 / 528 (the dynamic LDS block starts at 0)
                        asm volatile("" : "+v"(goff));
                        const lds_f32* const gc = (const lds_f32*)(uintptr_t)goff;
                        const float ga1 = gc[0], ga2 = gc[1], c015s = gc[2], sure_ms = gc[3];
                        // normal attempt A (rng.h:72-78): the stream's next two outputs
                        uint32_t l1, h1, l2, h2, l3, h3;
                        lcg52_step(s_lo, s_hi, l1, h1);
                        lcg52_step(l1, h1, l2, h2);
                        lcg52_step(l2, h2, l3, h3);
                        const float ufA = pool32_u(lcg52_top32(l1, h1));
                        const float svA = pool32_sv(lcg52_top32(l2, h2));
                        const float qA = pool32_q(ufA, svA);
                        const bool q_lo = qA > VGL_P32_QLO - VGL_P32_QBAND, q_hi = qA > VGL_P32_QHI + VGL_P32_QBAND;
                        bool slow_n = false;
                        const bool n_amb = have && q_lo && !q_hi;                   // the reference may look at the logarithm test (1.2 % of the attempts)
                        bool hold = n_amb && !full_n;
                        bool redo = false;
                        if (full_n && __builtin_amdgcn_ballot_w64(n_amb)) {
                            bool und;
                            if (DBG) c_ntest++;
                            slow_n = pool32_normal_slow(svA * __builtin_amdgcn_rcpf(ufA), ufA, qA, n_amb, und);
                            redo = und || (n_amb && dbg_redo_every && (l1 >> 8) % (uint32_t)(dbg_redo_every | (dbg_redo_every == 0)) == 0u);
                        }
                        const bool accA = !(q_lo && (q_hi || slow_n));
#if VGL_POOL_ATTEMPTS == 2
                        // attempt B: outputs three and four; used when A is rejected (by q alone or by its logarithm test) and B is accepted by q
                        // alone.  B inside the logarithm test's band: the state moves past A only, and B is the next iteration's attempt A
                        uint32_t l4, h4, l5, h5;
                        lcg52_step(l3, h3, l4, h4);
                        lcg52_step(l4, h4, l5, h5);
                        const float ufB = pool32_u(lcg52_top32(l3, h3));            // (also attempt A's third uniform)
                        const float svB = pool32_sv(lcg52_top32(l4, h4));
                        const float qB = pool32_q(ufB, svB);
                        const bool rejA = have && !hold && !accA && !redo;
                        const bool useB = rejA && !(qB > VGL_P32_QLO - VGL_P32_QBAND);
                        const bool rejB = rejA && (qB > VGL_P32_QHI + VGL_P32_QBAND);           // both rejected: the state moves past four outputs
                        const float uf = useB ? ufB : ufA, sv = useB ? svB : svA;
                        const float u2f = useB ? pool32_u(lcg52_top32(l5, h5)) : ufB;
                        const bool g_try0 = (have && accA && !hold) || useB;
#else
                        const float uf = ufA, sv = svA;
                        const float u2f = pool32_u(lcg52_top32(l3, h3));
                        const bool g_try0 = have && accA && !hold;
#endif
                        // operands of this lane's next item (as in the float64 loop)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn));
#ifdef VGL_POOL_CLAMP_NEXT
                        const uint32_t kc = (uint32_t)(kn < segT4 ? kn : segT * ISZ);   // l_it[segT] = 0 stands for "none" (a quarter's own limit is another quarter's live item)
#else
                        // (round 6: no clamp.  A lane whose claim lies past its items reads whatever 16 bits are there -- another quarter's slot, the table, or
                        //  beyond the block, which LDS answers with zeros -- and forms a table address of at most 255 x 16 bytes (the table has 256 entries) and an
                        //  LDS address below 512: nothing of it is used, the lane has no item any more (`have`).  Three vector instructions per iteration less.)
                        const uint32_t kc = P16 ? (uint32_t)kn : (uint32_t)(kn < segT4 ? kn : segT * ISZ);
#endif
                        uint32_t rd16, ow8, skn = 0;
                        if (P16) {
                            skn = *(const lds_u16*)(uintptr_t)(576u + kc);                                // owner << 10 | read << 2 | base
                            asm volatile("" : "+v"(skn));
                            rd16 = (skn << 2) & 0xFF0u;                                                   // 16 x read: byte offset of its jump
                            ow8 = (skn >> 7) & 0x1F8u;                                                    // 8 x owner: LDS address of l_stq[owner]
                        } else {
                            rd16 = *(const lds_u16*)(uintptr_t)(576u + kc);
                            ow8 = *(const lds_u16*)(uintptr_t)(578u + kc);
                            asm volatile("" : "+v"(rd16), "+v"(ow8));
                        }
                        const VglAffine tab_n = *(const VglAffine*)((const uint8_t*)P.qs_read_tab + rd16);
                        const uint64_t base_n = *(__attribute__((address_space(3))) const uint64_t*)(uintptr_t)ow8;
                        // gamma step (rng.h:139-145) on the accepted deviate
                        const float xn = sv * __builtin_amdgcn_rcpf(uf);             // v / u
                        const float w = __builtin_fmaf(ga2, xn, 1.0f);
                        const float vv = (w * w) * w;
                        const float xsq = xn * xn;
                        const float x4 = xsq * xsq;
                        const bool sure = (0x1p32f - u2f) >= __builtin_fmaf(x4, c015s, sure_ms);
                        const bool in_range = (w >= 0.5f) && (uf >= VGL_P32_UMIN);   // where the value bound (and the sure-accept bound) is stated
                        redo = redo || (g_try0 && !in_range);
                        const bool g_try = g_try0 && in_range;
                        const bool g_amb = g_try && !sure;                          // squeeze and logarithm test in the bounded block (0.2 % of the lanes)
                        hold = hold || (g_amb && !full);
                        bool slow_g = false;
                        if (full && __builtin_amdgcn_ballot_w64(g_amb)) {
                            bool und;
                            if (DBG) c_gtest++;
                            slow_g = pool32_gamma_slow(u2f, ga2 * xn, ga1, x4, g_amb, und);
                            redo = redo || und || (g_amb && dbg_redo_every && (l3 >> 8) % (uint32_t)(dbg_redo_every | (dbg_redo_every == 0)) == 1u);
                        }
                        const bool acc_g = g_try && !(g_amb && slow_g) && !hold;
                        // the stream moves past the outputs this iteration used: u2 is drawn only when w > 0 (rng.h:140-142; w <= 0 is out of range here)
#if VGL_POOL_ATTEMPTS == 2
                        {
                            const uint32_t t_lo = useB ? l5 : l3, t_hi = useB ? h5 : h3;           // a gamma step was tried on A / on B
                            const uint32_t r_lo = rejB ? l4 : l2, r_hi = rejB ? h4 : h2;           // none: past A and B / past A only
                            s_lo = hold ? s_lo : (g_try0 ? t_lo : r_lo);
                            s_hi = hold ? s_hi : (g_try0 ? t_hi : r_hi);
                        }
#else
                        s_lo = hold ? s_lo : (g_try0 ? l3 : l2);
                        s_hi = hold ? s_hi : (g_try0 ? h3 : h2);
#endif
                        const float val = ga1 * vv;
                        const bool fin = (acc_g && stage1) || redo;
                        const float gx_prev = gxf;
                        if constexpr (PREC) {
                            // the accepted attempt again, in double (rng.h:72-78: u, v = 1.7156 (u' - 0.5), x = v / u; :139-145: w = 1 + a2 x, a1 w^3): its two
                            // uniforms are the generator states the float32 attempt was formed from (A: outputs one and two, B: three and four); every decision
                            // has been taken, only the value is wanted.  For the second deviate of a read the error probability X / (X + Y) (rng.h:438) goes to
                            // the staging plane k_gl reads (a read the float32 loop hands to k_redo gets its probability there).
#ifdef VGL_EXP_NO_CHAIN
                            if (false) {
#else
                            if (acc_g) {
#endif
#if VGL_POOL_ATTEMPTS == 2
                                const uint32_t ulo = useB ? l3 : l1, uhi = useB ? h3 : h1, vlo = useB ? l4 : l2, vhi = useB ? h4 : h2;
#else
                                const uint32_t ulo = l1, uhi = h1, vlo = l2, vhi = h2;
#endif
                                const double ud = bits_1xxx_52r(((uint64_t)uhi << 32) | ulo, k3ff) - 1.0;
                                const double vd = 1.7156 * (bits_1xxx_52r(((uint64_t)vhi << 32) | vlo, k3ff) - 1.5);
                                const __attribute__((address_space(3))) double* const gd = (const __attribute__((address_space(3))) double*)(uintptr_t)((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)l_gcd + (stage1 ? 16u : 0u));
                                const double a1d = gd[0], a2d = gd[1];
                                const double xd = div_inrange(vd, ud);
                                const double wd = 1.0 + a2d * xd;
                                const double vald = a1d * (wd * wd * wd);
                                if (stage1) {
                                    if (!redo) {
                                        const size_t ei = vgl_errp_index((int)((sk >> 2) & 0xFFu), ev0 + (sk >> 10), P.read_cap);
#ifdef VGL_EXP_NO_ERRP_STORE
                                        const double epx = div_inrange(gxd, gxd + vald); if (epx > 2.0) T.errp[ei] = epx;      // (timing experiment: the value computed, never stored)
#else
                                        T.errp[ei] = div_inrange(gxd, gxd + vald);
#endif
                                    }
                                } else gxd = vald;
                            }
                        }
                        gxf = (acc_g && !stage1) ? val : gxf;
                        stage1 = (stage1 != acc_g) && !redo;
                        if (DBG) {
                            const uint64_t fm = __builtin_amdgcn_ballot_w64(fin);
                            c_finb += (fm != 0); c_lfin += (unsigned)__popcll(fm); c_lhold += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(hold));
                        }
                        if (fin) {
                            const float pf = gx_prev * __builtin_amdgcn_rcpf(gx_prev + val);   // the read's error probability X / (X + Y) (rng.h:438)
                            if (P16) {
                                // the read's quality score (vcfgl.cpp:500-523) here, by the lane that finished it: the staged byte into the item's
                                // slot; bit 8 where the float32 value cannot decide it (the owner hands those reads to k_redo when it stages them)
                                int q_i, aq_i;
                                bool ok = qs_decide_fix(P, pf, q_i, aq_i, k_adj) && !redo;
                                if (dbg_redo_every) ok = ok && ((uint32_t)(seg0 + (k >> 1)) % (uint32_t)(dbg_redo_every | (dbg_redo_every == 0))) != 2u;    // test hook
                                uint32_t qv, aqv = 0;
                                if (P.n_qs_bins != 0) {                                    // (wave-uniform)
                                    const uint32_t b = l_lut[q_i < 255 ? q_i : 255];       // host: every bin ends below 255 (vgl_ctx_create)
                                    uint32_t b2 = 0;
                                    if (!SLIM) { if (k_adj) { const int a_ = aq_i < 0 ? 0 : aq_i; b2 = l_lut[a_ < 255 ? a_ : 255]; } }
                                    const bool nobin = ok && (b == 0xFFu || b2 == 0xFFu);
                                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(nobin) != 0, 0)) { if (nobin) atomicOr(T.errflag, VGL_DEVERR_QSBIN); }
                                    qv = (b == 0xFFu) ? 0u : b; aqv = (b2 == 0xFFu) ? 0u : b2;
                                } else {
                                    qv = (uint32_t)((q_i > CAP_BASEQ) ? CAP_BASEQ : q_i);
                                    if (!SLIM) aqv = (uint32_t)((aq_i > CAP_BASEQ) ? CAP_BASEQ : (aq_i < 0 ? 0 : aq_i));
                                }
                                // the score the likelihoods take (vcfgl.cpp:525-531) in the staged byte; the score the quality sums take (:557-564) in
                                // bits 9-14 (the owner's flush adds them up), or added to the owner's word of that base right here (qfast)
                                const uint32_t q_gl = SLIM ? qv : ((k_adj & 1) ? aqv : qv), q_sum = SLIM ? 0u : ((k_adj & 2) ? aqv : qv);
                                uint32_t hi6 = ok ? ((q_gl << 2) | (q_sum << 9)) : 0x100u;   // (one select: no divergent paths around the slot's store)
                                asm volatile("" : "+v"(hi6));
                                *(lds_u16*)(uintptr_t)(576u + (uint32_t)k) = (uint16_t)(hi6 | (sk & 3u));
                                if (!SLIM) {
                                    if (qfast && k_qsum && ok)
                                        __hip_atomic_fetch_add(l_qs + ((sk >> 10) * 4u + (sk & 3u)), q_sum + ((q_sum * q_sum) << 13), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                }
                                sk = skn;
                            } else
                            // ... as a float32 in the item's slot; NaN: undecided (k_redo draws the read)
                            *(lds_u32*)(uintptr_t)(576u + (uint32_t)k) = redo ? 0x7FC00000u : __float_as_uint(pf);
                            aff52_step(tab_n, (uint32_t)base_n, (uint32_t)(base_n >> 32), s_lo, s_hi);     // the next item's stream: its read's jump on its owner's base
                            k = kn;
                            asm volatile("" : "+v"(k));
                            asm volatile("ds_add_rtn_u32 %0, %1, %2" : "+v"(kn) : "v"(ctr_p), "v"(four_v) : "memory");
                        }
                        have = k < segT4;
                    } while (__builtin_amdgcn_ballot_w64(have));
                } else
                do {                                         // segT >= 1: lane 0 has an item
                    if (DBG) c_iter++;
                    const bool full = (--slow_cnt == 0);     // bounded gamma test (needed by ~0.2 % of the lanes of an iteration)
                    if (full) slow_cnt = P.slow_period;
                    const bool full_n = (--slow_cnt_n == 0); // bounded normal-deviate test (~1.2 %)
                    if (full_n) slow_cnt_n = P.slow_period_n;
                    uint32_t goff = stage1 ? 32u : 0u;                         // l_gc[stage] = LDS byte 512 + goff (the dynamic LDS block starts at 0)
                    asm volatile("" : "+v"(goff));                              // one select, not one per element
                    const __attribute__((address_space(3))) double* gc = (const __attribute__((address_space(3))) double*)(uintptr_t)goff + 64;
                    const double ga1 = gc[0], ga2 = gc[1], ga2sq015 = gc[2], sure_margin = gc[3];
                    // normal attempt
                    uint64_t st1 = lcg_next52r(st);              // raw: bits 52-63 are masked where a uniform is built
                    asm volatile("" : "+v"(st1));                // (st2 from st1, not from a second product of st: one multiply-add fewer)
                    const uint64_t st2 = lcg_next52r(st1);
                    const uint64_t st3 = lcg_next52r(st2);
                    const double u = bits_1xxx_52r(st1, k3ff) - 1.0;
                    const double v = 1.7156 * (bits_1xxx_52r(st2, k3ff) - 1.5);
                    const double x = u - 0.449871;
                    const double y = fabs(v) + 0.386595;
                    const double q = (x * x) + y * (0.19600 * y - 0.25472 * x);
                    const bool q_lo = q > 0.27597, q_hi = q > 0.27846;
                    bool slow_n = false;
                    const bool n_amb = have && q_lo && !q_hi;
                    bool hold = n_amb && !full_n;
                    bool redo = false;                                              // DEFER: this item goes to k_redo
                    if (full_n && __builtin_amdgcn_ballot_w64(n_amb)) {
                        bool und;
                        slow_n = normal_slow_test_t<DEFER>(v, u, n_amb, und);
                        if (DEFER) redo = und || (n_amb && dbg_redo_every && ((uint32_t)(st1 >> 8) % (uint32_t)(dbg_redo_every | (dbg_redo_every == 0))) == 0u);
                    }
                    const bool acc_n = !(q_lo && (q_hi || slow_n));
                    // operands of this lane's next item, fetched here -- far enough behind the claim of the previous iteration and
                    // ahead of their use at the bottom (unconditional, clamped index: no divergent control flow in the loop)
                    // the claim a finishing lane made at the end of the previous iteration lands in kn itself (below): wait for it here
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn));
                    const uint32_t kc = (uint32_t)(kn < segT4 ? kn : segT4);                                     // l_it[segT] = 0 (a valid slot) stands for "none"
                    uint32_t rd16 = *(const lds_u16*)(uintptr_t)(576u + kc);                                     // 16 x read: byte offset of its jump
                    uint32_t ow8 = *(const lds_u16*)(uintptr_t)(578u + kc);                                      // 8 x owner: LDS address of l_stq[owner]
                    asm volatile("" : "+v"(rd16), "+v"(ow8));                                                    // (plain 32-bit values: ds_read_u16 has zero-extended them)
                    const uint32_t m_n = PREC ? (rd16 | (ow8 << 16)) : 0u;
                    const VglAffine tab_n = *(const VglAffine*)((const uint8_t*)P.qs_read_tab + rd16);
                    const uint64_t base_n = *(__attribute__((address_space(3))) const uint64_t*)(uintptr_t)ow8;   // l_stq[owner]
                    // gamma step on the accepted deviate
                    const double xn = div_inrange(v, u);
                    const double w = 1.0 + ga2 * xn;
                    const bool w_pos = w > 0.0;
                    const double vv = w * w * w;
                    const double u2 = bits_1xxx_52r(st3, k3ff) - 1.0;
                    const double xsq = xn * xn;
                    const double x4 = xsq * xsq;
                    const bool sq_fail = u2 > 1.0 - 0.0331 * x4;
                    // Where the squeeze fails the reference accepts iff log(u) <= 0.5 x^2 + a1 (1 - v + log v) = -x^2 s^2 P(s) / 3,
                    // s = a2 x, P(s) = 1/4 - s/5 + s^2/6 - ... (gamma_slow_test).  -log(1 - t) >= t, and P < 0.45 for s >= -1/2
                    // (P <= 1/4 for s >= 0; 1/4 + |s|/5 + s^2 / (6 (1 - |s|)) <= 0.434 for -1/2 <= s < 0), so
                    // 1 - u >= 0.15 a2^2 x^4 (+ P.sure_margin, far above the rounding of either side) is a sure accept: it settles 97 %
                    // of these cases (the sampler's rejection rate is 0.3 % for alpha ~ 10 and 0.003 % for alpha ~ 1000), and the
                    // bounded test below is left with ~0.2 % of the lanes of an iteration.
                    const bool sure = (1.0 - u2 >= __builtin_fma(ga2sq015, x4, sure_margin)) && (w >= 0.5);   // (the bound is this kernel's own: one fused step)
                    const bool g_try = have && acc_n && w_pos && !hold;
                    const bool g_amb = g_try && sq_fail && !sure;
                    hold = hold || (g_amb && !full);
                    bool slow_g = false;
                    if (full && __builtin_amdgcn_ballot_w64(g_amb)) {
                        bool und;
                        slow_g = gamma_slow_test_t<DEFER>(u2, xsq, ga1, vv, ga2 * xn, g_amb, und);
                        if (DEFER) redo = redo || und || (g_amb && dbg_redo_every && ((uint32_t)(st3 >> 8) % (uint32_t)(dbg_redo_every | (dbg_redo_every == 0))) == 1u);
                    }
                    const bool acc_g = g_try && !(g_amb && slow_g) && !hold;     // slow_g is meaningful on the lanes that asked for it
                    st = hold ? st : (g_try ? st3 : st2);    // u2 is drawn only when w > 0 (rng.h:140-142)
                    double val = ga1 * vv;
                    if (!DEFER && __builtin_expect(any_changed, 0)) {  // alpha < 1 (rng.h:146-148); wave-uniform guard (never with DEFER)
                        asm volatile("" ::: "memory");       // (keeps the per-lane part of the test out of the common path)
                        if (acc_g && (stage1 ? P.gy.changed : P.gx.changed)) {
                            double u3;
                            do { st = lcg_next52r(st); u3 = bits_1xxx_52r(st, k3ff) - 1.0; } while (u3 == 0.0);
                            val = pow(u3, 1.0 / (stage1 ? P.gy.alpha0 : P.gx.alpha0)) * ga1 * vv;
                        }
                    }
                    const bool fin = (acc_g && stage1) || redo;               // redo (DEFER): the item is dropped here, k_redo draws its read
                    const uint64_t st_n = tab_n.a * base_n + tab_n.c;     // raw, like st
                    const double gx_prev = gx;
                    gx = (acc_g && !stage1) ? val : gx;
                    stage1 = (stage1 != acc_g) && !redo;
                    if (fin) {
                        // a finished read leaves its error probability (rng.h:438) as a float32 in the item's slot; the
                        // quality scores are taken from it by the dense pass after the loop
                        *(lds_u32*)(uintptr_t)(576u + (uint32_t)k) = __float_as_uint(qs_stage_pf(gx_prev, val));   // l_it[item]
                        if (DEFER) { if (redo) *(lds_u32*)(uintptr_t)(576u + (uint32_t)k) = 0x7FC00000u; }        // NaN: undecided for the dense pass
                        if (PREC) {
                            // read index x plane as ONE 32 x 32 -> 64-bit multiply-add (the tile has fewer than 2^32 evaluations: vgl_launch_sample)
                            const size_t ei = vgl_errp_index((int)((it_m & 0xFFFFu) >> 4), ev0 + (it_m >> 19), P.read_cap);
                            T.errp[ei] = (DEFER ? div_inrange(gx_prev, gx_prev + val) : gx_prev / (gx_prev + val));   // DEFER: both shape parameters >= 8, the operands are far from the exponent limits
                            it_m = m_n;
                        }
                        // the lane adopts kn and claims the next unclaimed item from the wave's counter (any assignment of
                        // items to lanes gives the same result)
                        st = st_n;
                        k = kn;
                        asm volatile("" : "+v"(k));                                  // (the copy first: the atomic overwrites kn's register)
                        asm volatile("ds_add_rtn_u32 %0, %1, %2" : "+v"(kn) : "v"(ctr_p), "v"(four_v) : "memory");
                    }
                    have = k < segT4;
                } while (__builtin_amdgcn_ballot_w64(have));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn) : : "memory");       // the last claims have landed before kn's register is anyone else's
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            if constexpr (SEG == 1) { uint32_t l2 = (uint32_t)lane; asm volatile("" : "+v"(l2)); ev = ev0 + (active ? (size_t)l2 : 0); }
            if (DBG) { const unsigned long long c = clock64(); c_pool += c - c_tmp; c_tmp = c; }
            if (DBG && P.dbg_phase == 3) return true;
            if (!P16 && qfast && k_qsum) {                              // the pool loop is done with l_stq: zero the 64 x 4 sum words
                l_stq[lane] = 0ULL;
                ((uint64_t*)l_accB)[lane] = 0ULL;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // -- quality scores of the segment's reads (vcfgl.cpp:500-523): pass j takes item 64 j + lane per lane.  LEAN 3 with the LDS sums
            // (qfast): lane l takes items l dB + j instead, dB = the passes, made odd -- the items of one owner are neighbours in the pool, so
            // with neighbouring items in neighbouring lanes about twenty lanes of every pass added to the SAME LDS word (one evaluation, one
            // base), and the LDS serialises same-address atomics: 0.53 ms of the 14.3 ms launch; a stride of about twenty items puts the
            // lanes of a pass on different owners, the odd stride keeps their slot reads on different banks
            const int n_pass = P16 ? 0 : ((segT + 63) >> 6);          // (P16: the scores were taken in the pool loop)
            const int dB = (qfast && !P16) ? (n_pass | 1) : 0;
            for (int jp = 0; jp < (qfast ? dB : n_pass); ++jp) {
                const int kb = qfast ? jp : 64 * jp;                    // item of lane b in this pass: kb + b * kstep
                const int kstep = qfast ? dB : 1;
                const int kk = kb + lane * kstep;
                const bool inb = kk < segT;
                const float pf = __uint_as_float(l_it[inb ? kk : cap]);
                int q_i, aq_i;
                bool ok = qs_decide_pf(P, pf, q_i, aq_i, k_adj, F32 ? VGL_P32_TF_EXTRA : 0.0f) && !dbg_qs_exact;
                if (DEFER) { if (dbg_redo_every) ok = ok && ((uint32_t)(seg0 + kk) % (uint32_t)(dbg_redo_every | (dbg_redo_every == 0))) != 2u; }    // test hook
                uint64_t amb = __ballot(inb && !ok);
                if (__builtin_expect(amb != 0, 0)) {
                    // undecided in float32: the owner of the read draws its deviate again in double (rng.h:433-444) -- here, or (DEFER)
                    // in k_redo, which patches the staged read: the owner appends (evaluation, read) to the tile's list
                    while (amb) {
                        const int b = __builtin_ctzll(amb);
                        amb &= amb - 1;
                        const int Kb = seg0 + kb + b * kstep;           // index of the read in the wave's pool
                        if (active && Kb >= offs && Kb < offs + dp) {
                            if (DEFER) {
                                const uint32_t part = (uint32_t)wave_index & (VGL_REDO_PARTS - 1);      // neighbouring wavefronts append to different counters
                                const uint32_t idx = atomicAdd(T.redo_count + VGL_REDO_STRIDE * part, 1u);
                                if (idx < T.redo_cap) T.redo_list[(size_t)part * T.redo_cap + idx] = ((unsigned long long)ev << 10) | (unsigned long long)(Kb - offs);
                                else { const size_t bit = ev * (size_t)P.read_cap + (size_t)(Kb - offs); atomicOr(&T.redo_bits[bit >> 5], 1u << (bit & 31)); }   // list full: the bitmap
                            } else {
                                VglAffine jr = P.qs_read_tab[Kb - offs]; jr.c >>= 4;     // the table carries 16 c (aff52)
                                uint64_t st_x = aff(jr, l_stq[lane] >> 4);
                                const double ep = beta_draw(P, st_x);
                                int qe, aqe;
                                errprob_raw(P, ep, qe, aqe);
                                if (aqe < 0 && (P.adjust_qs & 3)) atomicOr(T.errflag, VGL_DEVERR_ADJQ);   // vcfgl.cpp:558, gl_methods.cpp:101
                                l_it[kb + b * kstep] = (uint32_t)(uint16_t)qe | ((uint32_t)(uint16_t)aqe << 16);
                            }
                        }
                    }
                    if (!DEFER) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (inb && !ok) { const uint32_t e = l_it[kk]; q_i = (int)(int16_t)(e & 0xFFFF); aq_i = (int)(int16_t)(e >> 16); }
                    } else if (inb && !ok) { q_i = 0; aq_i = SLIM ? -1 : 0; }   // placeholder: k_redo writes the read's score (and adds it to the quality sums, which take 0 here)
                }
                qs_finish(P, q_i, aq_i, T.errflag, DEFER ? (inb && ok) : inb);
                if (SLIM) { if (inb) l_pb[kk] = (uint8_t)((q_i << 2) | l_pb[kk]); }        // the staged byte itself: score << 2 | base
                else if (qfast) {
                    // --adjust-qs 0 or 3: one score serves the likelihoods and the quality sums (vcfgl.cpp:525-531, 557-564)
                    if (inb) {
                        const uint32_t pb = l_pb[kk];                   // owner << 2 | base
                        const uint32_t q_s = (uint32_t)((k_adj & 1) ? aq_i : q_i);
                        l_pb[kk] = (uint8_t)((q_s << 2) | (pb & 3u));
                        if (k_qsum) {
                            uint32_t* const slot = ((pb & 2u) ? l_accB : (uint32_t*)l_stq) + ((pb >> 2) * 2u + (pb & 1u));
                            __hip_atomic_fetch_add(slot, q_s + ((q_s * q_s) << 13), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                    }
                }
                else if (inb) l_it[kk] = (uint32_t)(q_i & 0xFF) | ((uint32_t)(aq_i & 0xFF) << 8);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // -- owners: combine base + quality score, stage the read, quality sums (vcfgl.cpp:525-564)
            // four reads per trip, one 32-bit store of the staged word (vgl_read_byte); a word cut by a segment boundary is
            // stored again, complete, by the next segment (carry_w)
            uint32_t sq0 = 0, sq1 = 0, sq2 = 0, sq3 = 0, sqq0 = 0, sqq1 = 0, sqq2 = 0, sqq3 = 0;   // this segment's share of the owner's quality sums
            if (P16) {
                // the pool loop has left every read's staged byte in the low byte of its 16-bit slot (bit 8: undecided): two (unaligned)
                // 32-bit LDS reads per word of four reads, the four bytes picked by one v_perm_b32, masked to the reads of this segment
                const uint8_t* const it8 = (const uint8_t*)l_it;
                // (SEG 1: the pointer is taken from the kernel arguments HERE -- pinned in vector registers ahead of the segment it was the one
                //  value the build still carried across the pool loop in scratch)
                uint8_t* reads_f = reads_v;
                if constexpr (SEG == 1) { reads_f = T.reads; asm volatile("" : "+v"(reads_f)); }
                for (int r0 = rdone & ~3; r0 < r_end; r0 += 4) {
                    uint32_t w_lo, w_hi;
                    __builtin_memcpy(&w_lo, it8 + 2 * (offs + r0 - seg0), 4);   // slots before / after the lane's reads are masked below
                    __builtin_memcpy(&w_hi, it8 + 2 * (offs + r0 - seg0) + 4, 4);
                    const uint32_t w4 = __builtin_amdgcn_perm(w_hi, w_lo, 0x06040200u);           // the low bytes of the four slots
                    const int lo = rdone > r0 ? rdone - r0 : 0, hi = r_end - r0 < 4 ? r_end - r0 : 4;
                    const uint32_t mask = (0xFFFFFFFFu << (8 * lo)) & (0xFFFFFFFFu >> (8 * (4 - hi)));
                    const uint32_t rw = (w4 & mask) | ((r0 < rdone) ? carry_w : 0u);
                    ((uint32_t*)reads_f)[(size_t)(r0 >> 2) * plane + ev] = rw;
                    carry_w = rw;
                    uint32_t und4 = __builtin_amdgcn_perm(w_hi, w_lo, 0x07050301u) & mask & 0x01010101u;   // the slots' high bytes: bit 0 = undecided
                    if (__builtin_expect(und4 != 0u, 0)) {
                        // this lane owns the read: (evaluation, read) to the tile's list for k_redo, which patches the staged byte
                        while (und4) {
                            const int j = __builtin_ctz(und4) >> 3;
                            und4 &= und4 - 1;
                            const uint32_t part = (uint32_t)wave_index & (VGL_REDO_PARTS - 1);      // neighbouring wavefronts append to different counters
                            const uint32_t idx = atomicAdd(T.redo_count + VGL_REDO_STRIDE * part, 1u);
                            if (idx < T.redo_cap) T.redo_list[(size_t)part * T.redo_cap + idx] = ((unsigned long long)ev << 10) | (unsigned long long)(r0 + j);
                            else { const size_t bit = ev * (size_t)P.read_cap + (size_t)(r0 + j); atomicOr(&T.redo_bits[bit >> 5], 1u << (bit & 31)); }   // list full: the bitmap
                        }
                    }
                    if (!SLIM) {
                        if (k_qsum && !qfast) {
                            // --adjust-qs 1 / 2 or deep staging: the score the sums take sits in bits 9-14 of each slot (0 for an undecided read:
                            // k_redo adds its score later), vcfgl.cpp:557-564
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t sl = ((j < 2 ? w_lo : w_hi) >> (16 * (j & 1))) & 0xFFFFu;
                                const bool mine = (r0 + j >= rdone) && (r0 + j < r_end) && !(sl & 0x100u);
                                const uint32_t qq = mine ? ((sl >> 9) & 63u) : 0u, q2 = (uint32_t)qs_to_qssq((int)qq), rb = sl & 3u;
                                sq0 += (rb == 0) ? qq : 0u; sq1 += (rb == 1) ? qq : 0u; sq2 += (rb == 2) ? qq : 0u; sq3 += (rb == 3) ? qq : 0u;
                                sqq0 += (rb == 0) ? q2 : 0u; sqq1 += (rb == 1) ? q2 : 0u; sqq2 += (rb == 2) ? q2 : 0u; sqq3 += (rb == 3) ? q2 : 0u;
                            }
                        }
                    }
                }
            } else if (SLIM || qfast) {
                // the dense pass has left the staged bytes in l_pb: one (unaligned) 32-bit LDS read per word, masked to the reads of
                // this segment
                for (int r0 = rdone & ~3; r0 < r_end; r0 += 4) {
                    uint32_t w4;
                    __builtin_memcpy(&w4, l_pb + (offs + r0 - seg0), 4);       // bytes before / after the lane's reads are masked below
                    const int lo = rdone > r0 ? rdone - r0 : 0, hi = r_end - r0 < 4 ? r_end - r0 : 4;
                    const uint32_t mask = (0xFFFFFFFFu << (8 * lo)) & (0xFFFFFFFFu >> (8 * (4 - hi)));
                    const uint32_t rw = (w4 & mask) | ((r0 < rdone) ? carry_w : 0u);
                    ((uint32_t*)reads_v)[(size_t)(r0 >> 2) * plane + ev] = rw;
                    carry_w = rw;
                }
            } else
            for (int r0 = rdone & ~3; r0 < r_end; r0 += 4) {
                uint32_t rw = (r0 < rdone) ? carry_w : 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = r0 + j;
                    if (r >= rdone && r < r_end) {
                        const int k = offs + r - seg0;
                        const int r_base = l_pb[k] & 3;                 // (LEAN 3 keeps the owner above the base)
                        const uint32_t qe = l_it[k];
                        const int q_i = (int)(qe & 0xFF);
                        const int aq_i = k_adj ? (int)((qe >> 8) & 0xFF) : -1;
                        const int q_gl = (k_adj & 1) ? aq_i : q_i;
                        rw |= (uint32_t)((q_gl << 2) | r_base) << (8 * j);
                        if (DUMP) { if (r < reads_out_cap_v) reads_out_v[(size_t)r * plane + ev] = (uint8_t)((q_i << 2) | r_base); }
                        if (k_qsum) {
                            const uint32_t qq = (uint32_t)((k_adj & 2) ? aq_i : q_i);
                            const uint32_t q2 = (uint32_t)qs_to_qssq((int)qq);
                            sq0 += (r_base == 0) ? qq : 0u; sq1 += (r_base == 1) ? qq : 0u;
                            sq2 += (r_base == 2) ? qq : 0u; sq3 += (r_base == 3) ? qq : 0u;
                            sqq0 += (r_base == 0) ? q2 : 0u; sqq1 += (r_base == 1) ? q2 : 0u;
                            sqq2 += (r_base == 2) ? q2 : 0u; sqq3 += (r_base == 3) ? q2 : 0u;
                        }
                    }
                }
                ((uint32_t*)reads_v)[(size_t)(r0 >> 2) * plane + ev] = rw;
                carry_w = rw;
            }
            rdone = r_end;
            if (!SLIM) {
                if (P16 && qfast && k_qsum) {                           // this owner's sums of the segment, from the words the finishing lanes added to
                    const uint64_t v01 = ((const uint64_t*)l_qs)[2 * lane], v23 = ((const uint64_t*)l_qs)[2 * lane + 1];
                    sq0 = (uint32_t)v01 & 0x1FFFu; sqq0 = (uint32_t)v01 >> 13; sq1 = (uint32_t)(v01 >> 32) & 0x1FFFu; sqq1 = (uint32_t)(v01 >> 32) >> 13;
                    sq2 = (uint32_t)v23 & 0x1FFFu; sqq2 = (uint32_t)v23 >> 13; sq3 = (uint32_t)(v23 >> 32) & 0x1FFFu; sqq3 = (uint32_t)(v23 >> 32) >> 13;
                    ((uint64_t*)l_qs)[2 * lane] = 0ULL; ((uint64_t*)l_qs)[2 * lane + 1] = 0ULL;     // (for the next segment)
                } else if (qfast && k_qsum) {                           // this owner's sums of the segment, from the dense pass's LDS words
                    const uint64_t v01 = l_stq[lane], v23 = ((const uint64_t*)l_accB)[lane];
                    sq0 = (uint32_t)v01 & 0x1FFFu; sqq0 = (uint32_t)v01 >> 13; sq1 = (uint32_t)(v01 >> 32) & 0x1FFFu; sqq1 = (uint32_t)(v01 >> 32) >> 13;
                    sq2 = (uint32_t)v23 & 0x1FFFu; sqq2 = (uint32_t)v23 >> 13; sq3 = (uint32_t)(v23 >> 32) & 0x1FFFu; sqq3 = (uint32_t)(v23 >> 32) >> 13;
                    if (seg0 + seglen < total) {                        // another segment follows: its pool loop needs the stream bases back
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                        if (active) l_stq[lane] = aff(P.off[3], aff(P.samp_tab[s], xb)) << 4;
                        else l_stq[lane] = 0ULL;
                    }
                }
                if (k_qsum) {
                    // the owner's quality sums of this segment: stored by the first segment (every evaluation's slots, zeros included), added by
                    // later ones; the site's integer totals for k_siteagg.  Nothing of them is carried across the pool loop.
                    if (active) {
                        uint32_t* q = T.qsum + (size_t)ls * 4 * N + s;
                        if (seg0 == 0) { q[0] = sq0; q[(size_t)N] = sq1; q[(size_t)2 * N] = sq2; q[(size_t)3 * N] = sq3; }
                        else { q[0] += sq0; q[(size_t)N] += sq1; q[(size_t)2 * N] += sq2; q[(size_t)3 * N] += sq3; }
                        if (P.need_qsumsq) {
                            uint32_t* qq = T.qsumsq + (size_t)ls * 4 * N + s;
                            if (seg0 == 0) { qq[0] = sqq0; qq[(size_t)N] = sqq1; qq[(size_t)2 * N] = sqq2; qq[(size_t)3 * N] = sqq3; }
                            else { qq[0] += sqq0; qq[(size_t)N] += sqq1; qq[(size_t)2 * N] += sqq2; qq[(size_t)3 * N] += sqq3; }
                        }
                    }
                    const uint32_t q4[4] = {sq0, sq1, sq2, sq3}, qq4[4] = {sqq0, sqq1, sqq2, sqq3};
                    wave_add_qsum_totals(T.acc + (size_t)ls * VGL_ACC_STRIDE, lane, q4, qq4, P.need_qsumsq != 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (DBG) c_flush += clock64() - c_tmp;
            if (DBG && P.dbg_phase == 4) return true;
            return false;
        };
        // (measured, same box: the two-instance form -3.6 % for LEAN 3 -- 10.25 -> 9.89 ms at qsi16 -- and +2.7 % for LEAN 2, whose single instance
        //  at 64 VGPRs the allocator already serves best: profiles/r05_ab/ab_single.txt)
        if constexpr (SEG == 1) {
            if (total > P.seg_limit) {                                  // (P.seg_limit = the pool's capacity; a test hook lowers it)
                if (lane == 0) T.seg_list[atomicAdd(T.redo_count + 1, 1u)] = (uint32_t)wave_index;     // redo_count[1]: zeroed with the redo counters before every tile
                return;
            }
            if (total > 0) { if (run_segment(std::true_type{}, 0)) return; }
        } else if constexpr (SEG == 2) {
            for (int seg0 = 0; seg0 < total; seg0 += seglen) { if (run_segment(std::false_type{}, seg0)) return; }
        } else {
        if (LEAN >= 3 && nseg <= 1) { if (total > 0) { if (run_segment(std::true_type{}, 0)) return; } }
        else for (int seg0 = 0; seg0 < total; seg0 += seglen) { if (run_segment(std::false_type{}, seg0)) return; }
        }
    }

    if constexpr (SEG == 1) { uint32_t l3 = (uint32_t)lane; asm volatile("" : "+v"(l3)); ev = ev0 + (active ? (size_t)l3 : 0); }    // (as behind the pool loop)
    if (active) {
        if (!k_strand) adf4 = ad4;
        if (T.fmt_dp) T.fmt_dp[ev] = dp;
        T.ad4[ev] = ad4;
        if (!SLIM && P.need_adf) T.adf4[ev] = adf4;
        if (!SLIM && P.need_qsum && (EQS != 2 || wave_total_reads == 0)) {       // (EQS 2: the segments have stored theirs -- unless the wavefront has no read at all)
            uint32_t* q = T.qsum + (size_t)ls * 4 * N + s;
            q[0] = qs0; q[(size_t)N] = qs1; q[(size_t)2 * N] = qs2; q[(size_t)3 * N] = qs3;
            if (P.need_qsumsq) {
                uint32_t* qq = T.qsumsq + (size_t)ls * 4 * N + s;
                qq[0] = qq0; qq[(size_t)N] = qq1; qq[(size_t)2 * N] = qq2; qq[(size_t)3 * N] = qq3;
            }
        }
        if (DUMP && T.reads_out) for (int r = dp; r < T.reads_out_cap; ++r) T.reads_out[(size_t)r * plane + ev] = 0xFF;
    }

    // ---- per-site sums: wave reduction, one atomic per wave and counter
    // every read of an evaluation shows one base: INFO/DP is the sum of the four per-base totals; without strand draws every read is "forward"
    int v[9];
    wave_sum_ad4(ad4, &v[1]);
    if (k_strand) wave_sum_ad4(adf4, &v[5]);
    else { v[5] = v[1]; v[6] = v[2]; v[7] = v[3]; v[8] = v[4]; }
    v[0] = v[1] + v[2] + v[3] + v[4];
    if (!SLIM && EQS != 2) {
        if (P.need_qsum) {                                   // the site's integer totals of the quality sums, for k_siteagg (EQS 2: added per segment)
            const uint32_t q4[4] = {qs0, qs1, qs2, qs3}, qq4[4] = {qq0, qq1, qq2, qq3};
            wave_add_qsum_totals(T.acc + (size_t)ls * VGL_ACC_STRIDE, lane, q4, qq4, P.need_qsumsq != 0);
        }
    }
    if (lane == 0) {
        int32_t* acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
#pragma unroll
        for (int k = 0; k < 9; ++k) if (v[k]) atomicAdd(&acc[k], v[k]);
        if (DBG) {
            atomicAdd(&T.dbg[0], 1ULL); atomicAdd(&T.dbg[1], clock64() - c_t0); atomicAdd(&T.dbg[2], c_pois);
            atomicAdd(&T.dbg[3], c_owner); atomicAdd(&T.dbg[4], c_pool); atomicAdd(&T.dbg[5], c_flush);
            atomicAdd(&T.dbg[6], c_iter); atomicAdd(&T.dbg[7], c_items);
            atomicAdd(&T.dbg[8], c_ntest); atomicAdd(&T.dbg[9], c_gtest); atomicAdd(&T.dbg[10], c_finb); atomicAdd(&T.dbg[11], c_lhave);
            atomicAdd(&T.dbg[12], c_lfin); atomicAdd(&T.dbg[13], c_lhold);
        }
    }
}

// wavefronts per SIMD the register allocator is asked for.  The float32 pool loop of the default tag surface (LEAN 2 without --precise-gl 1) runs at
// EIGHT (64 VGPRs; two-byte items, pools of 2240: vgl_host.cpp): with the redo list's counter out of the way more resident wavefronts pay (round 5,
// A/B on one box: 5 -> 8 wavefronts at depth 20 9.9 -> 9.4 ms with five-byte items in two segments, 8.5 with two-byte items in one); the
// optional-tag float32 build (LEAN 3) at SEVEN (72 VGPRs: 11.4 -> 10.5 ms; 8 wavefronts 10.7); the --precise-gl 1 builds at five, the
// inline-fallback builds at four
#ifndef VGL_SAMPLE_WAVES_F32
#define VGL_SAMPLE_WAVES_F32 8
#endif
#ifdef VGL_POOL_F64
#define VGL_SAMPLE_WAVES(EQS, PREC, LEAN) ((EQS) == 2 ? ((LEAN) >= 2 ? 5 : 4) : 8)
#else
#ifndef VGL_SAMPLE_WAVES_L3
#define VGL_SAMPLE_WAVES_L3 7
#endif
#ifndef VGL_SAMPLE_WAVES_PREC
#define VGL_SAMPLE_WAVES_PREC 5
#endif
#define VGL_SAMPLE_WAVES(EQS, PREC, LEAN) ((EQS) == 2 ? ((LEAN) == 2 && !(PREC) ? VGL_SAMPLE_WAVES_F32 : ((LEAN) >= 3 && !(PREC) ? VGL_SAMPLE_WAVES_L3 : ((LEAN) >= 2 ? VGL_SAMPLE_WAVES_PREC : 4))) : 8)
#endif
#ifndef VGL_SAMPLE_SEG_TU
template <int EQS, bool DBG, int DM, bool PREC, int LEAN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EQS == 2 ? VGL_SAMPLE_WAVES(EQS, PREC, LEAN) : 1, VGL_SAMPLE_WAVES(EQS, PREC, LEAN)))) void k_sample(const VglDevParams P, const VglTilePtrs T) {
    // (one 64-sample chunk per launched wavefront: several chunks per wavefront, one after the other, measured slower -- docs/tried.md)
    k_sample_body<EQS, DBG, DM, PREC, LEAN>(P, T, (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
}
#endif

// Translation units (round 6, second session): the k_sample_seg kernels are built from vgl_sample_seg.hip (= this file with VGL_SAMPLE_SEG_TU: the body
// template, the kernels below and their launcher, nothing else), so that they can be compiled with their own scheduler options (Makefile: SEGFLAGS) --
// an LLVM scheduling strategy is a property of the whole module, and no setting held across ALL kernels of this file (docs/tried.md, round 6).
extern "C" int vgl_sample_seg_launch(int dm, int lean, unsigned grid1, unsigned grid2, size_t lds, void* stream, const VglDevParams* p, const VglTilePtrs* t);
extern "C" size_t vgl_sample_seg_static_lds(void);           // the largest static LDS of the seg kernels (must be 0), (size_t)-1 on a query error
#ifdef VGL_SAMPLE_SEG_TU
// the float32 build of the default tag surface (k_sample<2, false, DM, false, 2>) as two kernels: SEG 1 = one pool segment per wavefront, SEG 2 = the
// segment loop over the wavefronts SEG 1 listed (k_sample_body, SEG)
// (eight wavefronts per SIMD for the optional-tag builds too: without the segment loop LEAN 3 / 4 need 68-69 registers, and at 64 they measure 2.5-3 % faster
//  than at 72 -- qsi16 9.40 -> 9.15 ms, alltags 9.58 -> 9.31, same box: profiles/r06_ab/ab_seg_split.txt)
template <int DM, int SEG, int LEAN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VGL_SAMPLE_WAVES_F32, VGL_SAMPLE_WAVES_F32))) void k_sample_seg(const VglDevParams P, const VglTilePtrs T) {
    if constexpr (SEG == 1) k_sample_body<2, false, DM, false, LEAN, 1>(P, T, (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
    else {
        const uint32_t n = __builtin_amdgcn_readfirstlane(T.redo_count[1]);
        for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {          // (one wavefront per workgroup)
            k_sample_body<2, false, DM, false, LEAN, 2>(P, T, (int64_t)T.seg_list[i]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

extern "C" int vgl_sample_seg_launch(const int dm, const int lean, const unsigned grid1, const unsigned grid2, const size_t lds, void* stream, const VglDevParams* p, const VglTilePtrs* t) {
    const dim3 g(grid1), g2(grid2), b(64);
    hipStream_t s = (hipStream_t)stream;
#define VGL_LAUNCH_SEG_DM(DM, LEAN) do { hipLaunchKernelGGL((k_sample_seg<DM, 1, LEAN>), g, b, lds, s, *p, *t); hipLaunchKernelGGL((k_sample_seg<DM, 2, LEAN>), g2, b, lds, s, *p, *t); } while (0)
#define VGL_LAUNCH_SEG(LEAN) do { if (dm == 1) VGL_LAUNCH_SEG_DM(1, LEAN); else if (dm == 2) VGL_LAUNCH_SEG_DM(2, LEAN); else VGL_LAUNCH_SEG_DM(0, LEAN); } while (0)
    if (lean == 2) VGL_LAUNCH_SEG(2); else if (lean == 3) VGL_LAUNCH_SEG(3); else if (lean == 4) VGL_LAUNCH_SEG(4); else return (int)hipErrorInvalidValue;
#undef VGL_LAUNCH_SEG
#undef VGL_LAUNCH_SEG_DM
    return (int)hipGetLastError();
}
extern "C" size_t vgl_sample_seg_static_lds(void) {
    size_t worst = 0;
    hipFuncAttributes a;
#define VGL_STATIC_LDS_SEG(DM, SEG, LEAN) \
    if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sample_seg<DM, SEG, LEAN>)) != hipSuccess) return (size_t)-1; \
    worst = a.sharedSizeBytes > worst ? a.sharedSizeBytes : worst;
#define VGL_STATIC_LDS_SEG3(DM, SEG) VGL_STATIC_LDS_SEG(DM, SEG, 2) VGL_STATIC_LDS_SEG(DM, SEG, 3) VGL_STATIC_LDS_SEG(DM, SEG, 4)
    VGL_STATIC_LDS_SEG3(0, 1) VGL_STATIC_LDS_SEG3(1, 1) VGL_STATIC_LDS_SEG3(2, 1) VGL_STATIC_LDS_SEG3(0, 2) VGL_STATIC_LDS_SEG3(1, 2) VGL_STATIC_LDS_SEG3(2, 2)
#undef VGL_STATIC_LDS_SEG3
#undef VGL_STATIC_LDS_SEG
    return worst;
}
#else  // !VGL_SAMPLE_SEG_TU: everything else of this file

// the deviate dump (vgl_tile_out.read_errp, [read][site][N] as the ABI documents it) from the evaluation-major staging planes: one lane per evaluation
__global__ __launch_bounds__(256) void k_errp_dump(const double* __restrict__ errp, double* __restrict__ out, const size_t n_eval, const int rows, const int read_cap) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_eval) return;
    for (int r = 0; r < rows; ++r) out[(size_t)r * n_eval + e] = errp[vgl_errp_index(r, e, read_cap)];
}
extern "C" int vgl_launch_errp_dump(const VglDevParams* p, const double* errp, double* out, size_t n_eval, int rows, void* stream) {
    if (n_eval == 0 || rows <= 0) return 0;
    hipLaunchKernelGGL(k_errp_dump, dim3((unsigned)((n_eval + 255) / 256)), dim3(256), 0, (hipStream_t)stream, errp, out, n_eval, rows, (int)p->read_cap);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
extern "C" int vgl_launch_sitebase(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    if (t->n_sites == 0) return 0;
    hipLaunchKernelGGL(k_sitebase, dim3((unsigned)((t->n_sites + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_depth(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    const int64_t E = (int64_t)t->n_sites * p->n_samples;
    if (E == 0) return 0;
    // measured at 65536 x 1000 / x 2000 evaluations, depth 20 / 30: 0.478 / 0.904 ms with chunks of 1024, 0.446 / 0.841 with 2048, 0.438 / 0.820
    // with 4096; a small tile keeps the short chunks (a wavefront works its chunk off alone: 35 us per 1024 evaluations)
    int chunk = E >= ((int64_t)1 << 24) ? VGL_DEPTH_CHUNK_MAX : (E >= ((int64_t)1 << 23) ? 2048 : VGL_DEPTH_CHUNK);
    if (p->dbg_depth_chunk == 1024 || p->dbg_depth_chunk == 2048 || p->dbg_depth_chunk == 4096) chunk = p->dbg_depth_chunk;
    const int64_t waves = (E + chunk - 1) / chunk;
    if (!p->per_sample_depth && p->pois_zt) hipLaunchKernelGGL(k_depth<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *p, *t, chunk);
    else hipLaunchKernelGGL(k_depth<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *p, *t, chunk);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// The reads k_sample<2, ., ., ., 2> could not settle in float32: one lane per list entry draws the read's error probability again
// in double from the read's own stream (rng.h:433-444: the stream of a read does not depend on who works on it), takes the
// quality score the exact way (vcfgl.cpp:500-523) and writes it into the staged read (the base bits stay); with -addQS / -addI16 it also
// adds the score to the evaluation's per-base quality sums and the site totals (the kernel counted 0 for the read).  No per-read dump
// (reads_out) with the deferred builds: nothing else depends on the score.
__device__ __forceinline__ void redo_read(const VglDevParams& P, const VglTilePtrs& T, const size_t ev, const int r) {
    const size_t N = (size_t)P.n_samples, plane = (size_t)T.n_sites * N;
    const size_t ls = ev / N, s = ev - ls * N;
    const uint64_t xe = aff(P.samp_tab[s], T.site_base[ls]);
    const uint64_t st_qs = aff(P.off[3], xe);
    VglAffine jr = P.qs_read_tab[r]; jr.c >>= 4;                         // the table carries 16 c (aff52)
    uint64_t st_x = aff(jr, st_qs);
    const double ep = beta_draw(P, st_x);
    int q, aq;
    errprob_raw(P, ep, q, aq);
    if (aq < 0 && (P.adjust_qs & 3)) atomicOr(T.errflag, VGL_DEVERR_ADJQ);   // vcfgl.cpp:558, gl_methods.cpp:101
    qs_finish(P, q, aq, T.errflag, true);
    if (T.errp) T.errp[vgl_errp_index(r, ev, P.read_cap)] = ep;                     // --precise-gl 1 / the deviate dump: the read's exact error probability
    uint8_t* const p = T.reads + vgl_read_byte(r, plane, ev);
    const uint32_t base = *p & 3u;
    const int q_gl = (P.adjust_qs & 1) ? aq : q;                         // vcfgl.cpp:525-531
    *p = (uint8_t)(((uint32_t)q_gl << 2) | base);
    if (P.need_qsum) {
        // LEAN 3: the evaluation's quality sums and the site totals took 0 for this read (several reads of one evaluation may be here: atomics)
        const uint32_t qq = (uint32_t)((P.adjust_qs & 2) ? aq : q), q2 = (uint32_t)qs_to_qssq((int)qq);
        int32_t* acc = T.acc + ls * VGL_ACC_STRIDE;
        atomicAdd(&T.qsum[(ls * 4 + base) * N + s], qq);
        atomicAdd((unsigned int*)&acc[VGL_ACC_QSUM + base], qq);
        if (P.need_qsumsq) { atomicAdd(&T.qsumsq[(ls * 4 + base) * N + s], q2); atomicAdd((unsigned int*)&acc[VGL_ACC_QSUMSQ + base], q2); }
    }
}
__global__ __launch_bounds__(64) void k_redo(const VglDevParams P, const VglTilePtrs T) {
    // workgroup b works on partition b mod VGL_REDO_PARTS of the list, together with the gridDim / VGL_REDO_PARTS - 1 others of that partition
    const uint32_t part = blockIdx.x & (VGL_REDO_PARTS - 1), sub = blockIdx.x / VGL_REDO_PARTS, nsub = gridDim.x / VGL_REDO_PARTS;
    const uint32_t cnt_p = T.redo_count[VGL_REDO_STRIDE * part];
    const uint32_t n = cnt_p < T.redo_cap ? cnt_p : T.redo_cap;
    for (uint32_t i = sub * 64u + threadIdx.x; i < n; i += nsub * 64u) {
        const unsigned long long e = T.redo_list[(size_t)part * T.redo_cap + i];
        redo_read(P, T, (size_t)(e >> 10), (int)(e & 1023u));
    }
    bool over = false;                                                   // some partition took more than it holds (wave-uniform)
    for (int q = 0; q < VGL_REDO_PARTS; ++q) over = over || (T.redo_count[VGL_REDO_STRIDE * q] > T.redo_cap);
    // More undecided reads than the list holds (never seen with the flag sets this build is chosen for, but then the result must
    // still be right): the owners marked the rest in a bitmap over (evaluation, read), which is walked and cleared here.
    if (over) {
        const size_t words = ((size_t)T.n_sites * (size_t)P.n_samples * (size_t)P.read_cap + 31) >> 5;
        for (size_t w = (size_t)blockIdx.x * 64u + threadIdx.x; w < words; w += (size_t)gridDim.x * 64u) {
            uint32_t m = T.redo_bits[w];
            if (!m) continue;
            T.redo_bits[w] = 0u;
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                const size_t bit = (w << 5) + (size_t)b;
                redo_read(P, T, bit / (size_t)P.read_cap, (int)(bit % (size_t)P.read_cap));
            }
        }
    }
}

// a deferred build of k_sample<2> (LEAN 2: default tag surface, LEAN 3: optional tags) serves this tile: vgl_launch_sample runs it,
// vgl_launch_redo runs k_redo behind it
static bool sample_deferred(const VglDevParams* p, const VglTilePtrs* t) {
    return !p->serial && p->error_qs == 2 && !t->reads_out && p->defer_ok && t->redo_list && !(t->dbg != nullptr && !t->errp && p->dbg_stamps != 2);
}
extern "C" int vgl_launch_redo(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    if ((int64_t)t->n_sites * p->chunks == 0 || !sample_deferred(p, t)) return 0;
    hipLaunchKernelGGL(k_redo, dim3(32 * VGL_REDO_PARTS), dim3(64), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_sample(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    const int64_t waves = (int64_t)t->n_sites * p->chunks;
    if (waves == 0) return 0;
    if ((int64_t)t->n_sites * p->n_samples >= (1LL << 32)) return (int)hipErrorInvalidValue;     // evaluation indices of a tile are 32-bit in places
    if (p->serial) return vgl_launch_sample_serial(p, t, stream);
    const bool dbg = t->dbg != nullptr;                         // VGL_DEBUG_STAMPS / VGL_DEBUG_PHASE (a -DVGL_TEST_HOOKS library only)
    (void)dbg;
    // wavefronts never cooperate here, and a workgroup's wave slots and LDS are only handed on when its last
    // wavefront retires: one wavefront per workgroup keeps every SIMD at its full complement of waves
    const int wpb = 1;
    const dim3 g((unsigned)((waves + wpb - 1) / wpb)), b(64 * wpb);
    // LDS per wavefront of the build that is launched: five bytes per item (slot + base plane; + 512 B of quality-sum words with qsum_lds), or two
    // (k_sample<2, LEAN 2> without --precise-gl 1: P16)
    const size_t lds5 = (size_t)wpb * ((((size_t)576 + 4 * ((size_t)p->pool_cap + 2) + (size_t)p->pool_cap + 7) & ~(size_t)7) + (p->qsum_lds ? 512 : 0));
#ifdef VGL_POOL_F64
    const size_t lds16 = lds5, lds16x = lds5;
#else
    const size_t lds16 = (size_t)wpb * ((((size_t)576 + 2 * ((size_t)p->pool_cap + 2) + 7) & ~(size_t)7) + (p->n_qs_bins ? 256 : 0));
    const size_t lds16x = lds16 + (size_t)wpb * (p->qsum_lds ? 1024 : 0);          // LEAN 3: + the quality-sum words
#endif
    const size_t lds = lds5;
    hipStream_t s = (hipStream_t)stream;
    const int dm = p->depth_pre;                                  // 0 mixed / 1 k_depth / 2 product method only
#define VGL_LAUNCH_SAMPLE(EQS, DBG, PREC, LEAN, LDS) \
    do { if (dm == 1) hipLaunchKernelGGL((k_sample<EQS, DBG, 1, PREC, LEAN>), g, b, LDS, s, *p, *t); \
         else if (dm == 2) hipLaunchKernelGGL((k_sample<EQS, DBG, 2, PREC, LEAN>), g, b, LDS, s, *p, *t); \
         else hipLaunchKernelGGL((k_sample<EQS, DBG, 0, PREC, LEAN>), g, b, LDS, s, *p, *t); } while (0)
    const bool lean = p->lean_ok && !t->reads_out;
    if (p->error_qs == 2) {
        // the pool loop addresses LDS by integer offsets from 0: none of its instantiations may own static LDS
        static const bool static_lds_free = [] {
            size_t worst = 0;
            hipFuncAttributes a;
#define VGL_STATIC_LDS(DBG, DM, PREC, LEAN) \
            if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sample<2, DBG, DM, PREC, LEAN>)) != hipSuccess) return false; \
            worst = a.sharedSizeBytes > worst ? a.sharedSizeBytes : worst;
#ifdef VGL_TEST_HOOKS
            VGL_STATIC_LDS(true, 0, false, 0) VGL_STATIC_LDS(true, 1, false, 0) VGL_STATIC_LDS(true, 2, false, 0)
            VGL_STATIC_LDS(true, 0, false, 2) VGL_STATIC_LDS(true, 1, false, 2) VGL_STATIC_LDS(true, 2, false, 2)
#endif
            VGL_STATIC_LDS(false, 0, true, 1) VGL_STATIC_LDS(false, 1, true, 1) VGL_STATIC_LDS(false, 2, true, 1)
            VGL_STATIC_LDS(false, 0, true, 0) VGL_STATIC_LDS(false, 1, true, 0) VGL_STATIC_LDS(false, 2, true, 0)
            VGL_STATIC_LDS(false, 0, false, 1) VGL_STATIC_LDS(false, 1, false, 1) VGL_STATIC_LDS(false, 2, false, 1)
            VGL_STATIC_LDS(false, 0, false, 2) VGL_STATIC_LDS(false, 1, false, 2) VGL_STATIC_LDS(false, 2, false, 2)
            VGL_STATIC_LDS(false, 0, true, 2) VGL_STATIC_LDS(false, 1, true, 2) VGL_STATIC_LDS(false, 2, true, 2)
            VGL_STATIC_LDS(false, 0, false, 0) VGL_STATIC_LDS(false, 1, false, 0) VGL_STATIC_LDS(false, 2, false, 0)
            VGL_STATIC_LDS(false, 0, false, 3) VGL_STATIC_LDS(false, 1, false, 3) VGL_STATIC_LDS(false, 2, false, 3)
            VGL_STATIC_LDS(false, 0, true, 3) VGL_STATIC_LDS(false, 1, true, 3) VGL_STATIC_LDS(false, 2, true, 3)
            VGL_STATIC_LDS(false, 0, false, 4) VGL_STATIC_LDS(false, 1, false, 4) VGL_STATIC_LDS(false, 2, false, 4)
#undef VGL_STATIC_LDS
            { const size_t ws = vgl_sample_seg_static_lds(); if (ws == (size_t)-1) return false; worst = ws > worst ? ws : worst; }
            return worst == 0;
        }();
        if (!static_lds_free) return (int)hipErrorInvalidConfiguration;
#ifdef VGL_TEST_HOOKS
        if (dbg && !t->errp && p->dbg_stamps == 2 && lean && sample_deferred(p, t)) { VGL_LAUNCH_SAMPLE(2, true, false, 2, lds16); } else   // VGL_DEBUG_STAMPS=2: the stamped float32 build
        if (dbg && !t->errp) { VGL_LAUNCH_SAMPLE(2, true, false, 0, lds); } else   // diagnostic build (VGL_DEBUG_STAMPS / VGL_DEBUG_PHASE): --precise-gl 0 only
#endif
#if defined(VGL_POOL_F64) || defined(VGL_PREC_F64)
        if (t->errp && sample_deferred(p, t)) { if (lean) VGL_LAUNCH_SAMPLE(2, false, true, 2, lds); else VGL_LAUNCH_SAMPLE(2, false, true, 3, lds); }   // --precise-gl 1: the deferred builds, k_redo (vgl_launch_redo) also rewrites errp
#else
        if (t->errp && sample_deferred(p, t)) { if (lean) VGL_LAUNCH_SAMPLE(2, false, true, 2, lds16 + 32); else VGL_LAUNCH_SAMPLE(2, false, true, 3, lds16x + 32); }   // --precise-gl 1: the float32 loop + the double value chain (two-byte items)
#endif
        else if (t->errp) { if (lean) VGL_LAUNCH_SAMPLE(2, false, true, 1, lds); else VGL_LAUNCH_SAMPLE(2, false, true, 0, lds); }   // --precise-gl 1, or the deviates were asked for
        else if (sample_deferred(p, t)) {
            // one segment per wavefront, then the segment loop over what that kernel listed (nothing, at the bench configurations)
            const dim3 g2((unsigned)(waves < 1024 ? waves : 1024));
#define VGL_LAUNCH_SEG(LEAN, LDS) do { const int rc_ = vgl_sample_seg_launch(dm, LEAN, g.x, g2.x, LDS, stream, p, t); if (rc_) return rc_; } while (0)
            const bool split = p->seg_split && t->seg_list;
            if (lean && split) VGL_LAUNCH_SEG(2, lds16);
            else if (!lean && split && p->adjust_qs == 0 && p->sample_strand && p->need_qsum) VGL_LAUNCH_SEG(4, lds16x);
            else if (!lean && split) VGL_LAUNCH_SEG(3, lds16x);
            else if (lean) VGL_LAUNCH_SAMPLE(2, false, false, 2, lds16);
            else if (p->adjust_qs == 0 && p->sample_strand && p->need_qsum) VGL_LAUNCH_SAMPLE(2, false, false, 4, lds16x);      // LEAN 3 with its three options fixed
            else VGL_LAUNCH_SAMPLE(2, false, false, 3, lds16x);
        }
        else { if (lean) VGL_LAUNCH_SAMPLE(2, false, false, 1, lds); else VGL_LAUNCH_SAMPLE(2, false, false, 0, lds); }
    }
    // fixed quality score: the LEAN build (no strand draws, forward-strand depths, quality sums or per-read dump) keeps those
    // options' per-read instructions out of the read loop
    else if (p->error_qs == 1) { if (lean) VGL_LAUNCH_SAMPLE(1, false, false, 1, 0); else VGL_LAUNCH_SAMPLE(1, false, false, 0, 0); }
    else { if (lean) VGL_LAUNCH_SAMPLE(0, false, false, 1, 0); else VGL_LAUNCH_SAMPLE(0, false, false, 0, 0); }
#undef VGL_LAUNCH_SAMPLE
    return (int)hipGetLastError();
}

#ifdef VGL_TEST_HOOKS
// ------------------------------------------------------------------------------------
// debug hook (not part of the C ABI): the raw hardware / ocml float32 functions over a buffer, for the exploratory scripts
// tools/vlogcheck.py and tools/sincos_check.py.  The bounds the kernels rely on are asserted by tests/test_gpu_bounds.py
// through vgl_dbg_bound_sweep (vgl_bounds.hip), over every float32 argument.
__global__ void k_dbg_vlog(const float* in, float* out, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) out[i] = __builtin_amdgcn_logf(in[i]);          // v_log_f32
    else if (mode == 1) out[i] = tanf(in[i]);                      // ocml tanf
    else if (mode == 2) out[i] = __builtin_amdgcn_exp2f(in[i]);    // v_exp_f32
    else if (mode == 3) out[i] = __builtin_amdgcn_sinf(in[i]);     // v_sin_f32: sin(2 pi x)
    else out[i] = __builtin_amdgcn_cosf(in[i]);                    // v_cos_f32: cos(2 pi x)
}
extern "C" __attribute__((visibility("default"))) int vgl_dbg_vlog(const float* d_in, float* d_out, int n, int mode) {
    hipLaunchKernelGGL(k_dbg_vlog, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, d_out, n, mode);
    return (int)hipDeviceSynchronize();
}
#endif
#endif  // !VGL_SAMPLE_SEG_TU
