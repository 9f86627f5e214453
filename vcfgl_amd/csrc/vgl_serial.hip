// vgl_serial.hip -- VGL_RNG_SERIAL: the reference's own random-stream order, bit for bit
// (SURVEY Appendix B).  A sequential scout finds every evaluation's stream states; the evaluations
// are then computed in parallel from the recorded states.
#include "vgl_common.hip.h"

// ------------------------------------------------------------------------------------
// VGL_RNG_SERIAL: the reference's own stream order.  The three rand48 streams
// and the mt19937 of the default beta sampler are consumed in a data-dependent serial order, so a
// sequential scout (one lane) walks the tile once and records, per evaluation, the state of each
// stream at the point the reference reaches it; the per-read beta deviates (one global mt19937 /
// rng2 stream) are recorded too.  All the remaining work (Poisson evaluation, reads, likelihoods)
// then runs in parallel from those recorded states and reproduces the serial program exactly.

// std::mt19937 (libstdc++) -- default BetaSampler, rng.h:353-421
__device__ uint32_t mt_next(VglSerialState* S) {
    if (S->mt_idx >= 624) {
        for (int i = 0; i < 624; i++) {
            const uint32_t y = (S->mt[i] & 0x80000000u) | (S->mt[(i + 1) % 624] & 0x7fffffffu);
            S->mt[i] = S->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        S->mt_idx = 0;
    }
    uint32_t y = S->mt[S->mt_idx++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}
// std::generate_canonical<double,53>
__device__ double mt_canonical(VglSerialState* S) {
    double sum = 0.0, tmp = 1.0;
    sum += (double)mt_next(S) * tmp; tmp *= 4294967296.0;
    sum += (double)mt_next(S) * tmp; tmp *= 4294967296.0;
    double r = sum / tmp;
    if (r >= 1.0) r = 0x1.fffffffffffffp-1;
    return r;
}
// std::gamma_distribution<double>(alpha,1)(gen) on a fresh distribution object (rng.h:409-412)
__device__ double std_gamma_fresh(VglSerialState* S, const double alpha) {
    const double malpha = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double a1 = malpha - 1.0 / 3.0;
    const double a2 = 1.0 / sqrt(9.0 * a1);
    bool saved_avail = false; double saved = 0.0;
    double u, v, n;
    do {
        do {
            if (saved_avail) { saved_avail = false; n = saved; }
            else {
                double x, y, r2;
                do {
                    x = 2.0 * mt_canonical(S) - 1.0;
                    y = 2.0 * mt_canonical(S) - 1.0;
                    r2 = x * x + y * y;
                } while (r2 > 1.0 || r2 == 0.0);
                const double mult = sqrt(-2 * log(r2) / r2);
                saved = x * mult; saved_avail = true;
                n = y * mult;
            }
            v = 1.0 + a2 * n;
        } while (v <= 0.0);
        v = v * v * v;
        u = mt_canonical(S);
    } while (u > 1.0 - 0.0331 * n * n * n * n && (log(u) > (0.5 * n * n + a1 * (1.0 - v + log(v)))));
    if (alpha == malpha) return a1 * v;
    do u = mt_canonical(S); while (u == 0.0);
    return pow(u, 1.0 / alpha) * a1 * v;
}
__device__ double serial_beta(const VglDevParams& P, VglSerialState* S) {
    if (P.beta_std) {
        const double x = std_gamma_fresh(S, P.beta_a);
        const double y = std_gamma_fresh(S, P.beta_b);
        return x / (x + y);
    }
    return beta_draw(P, S->st2);
}

// glibc rand(): random_r() TYPE_3
__device__ int glibc_rand(VglSerialState* S) {
    const uint32_t val = (S->rand_state[S->rand_f] += S->rand_state[S->rand_r]);
    if (++S->rand_f >= 31) { S->rand_f = 0; ++S->rand_r; }
    else if (++S->rand_r >= 31) S->rand_r = 0;
    return (int)(val >> 1);
}

__global__ __launch_bounds__(64) void k_scout(const VglDevParams P, const VglTilePtrs T, VglSerialState* S) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int N = P.n_samples;
    uint64_t st0 = S->st0, st1 = S->st1;
    for (int ls = 0; ls < T.n_sites; ++ls) {
        const size_t e0 = (size_t)ls * N;
        // depths of all samples first (vcfgl.cpp:364-389)
        long long info_dp = 0;
        for (int s = 0; s < N; ++s) {
            int n;
            if (P.per_sample_depth) { const VglPois pc = P.pois[s]; n = poisson_draw_fast(pc, st1, P.gamma_ln_tab, P.gamma_ln_n); }
            else n = poisson_draw_fast(P.pois0, st1, P.gamma_ln_tab, P.gamma_ln_n);
            const uint32_t g = T.gt[e0 + s];
            if ((g & 0xF) == 0xF || ((g >> 4) & 0xF) == 0xF) n = 0;
            if (n > P.read_cap) { atomicOr(T.errflag, VGL_DEVERR_CAPACITY); n = P.read_cap; }
            T.sdp[e0 + s] = n;
            info_dp += n;
        }
        uint64_t thresh = P.err_thresh;
        int last_base = -1;
        if (T.site_tail) { VglSiteTail z; z.sum = 0.0f; z.sumsq = 0.0f; z.base = -1; z.pad = 0; T.site_tail[ls] = z; }
        if (info_dp == 0) { if (P.error_qs == 1) T.site_thresh[ls] = thresh; continue; }   // nothing else is drawn (vcfgl.cpp:396-404)
        if (P.error_qs == 1) {                                     // vcfgl.cpp:425-437
            const double pe = serial_beta(P, S);
            if (T.site_pick_err) T.site_pick_err[ls] = pe;
            thresh = (uint64_t)ceil(ldexp(pe, 48));
            T.site_thresh[ls] = thresh;
        }
        for (int s = 0; s < N; ++s) {
            const int dp = T.sdp[e0 + s];
            T.sst_hap[e0 + s] = st1;
            T.sst_base[e0 + s] = st0;
            if (dp == 0) continue;
            const uint32_t g = T.gt[e0 + s];
            const int a0 = g & 0xF, a1 = (g >> 4) & 0xF;
            for (int r = 0; r < dp; ++r) {                          // vcfgl.cpp:469-613
                bool fwd;
                last_base = sample_read_base(st1, st0, a0, a1, thresh, P.sample_strand != 0, fwd);
                if (P.error_qs == 2) T.errp[vgl_errp_index(r, (size_t)e0 + s, P.read_cap)] = serial_beta(P, S);
            }
        }
        if (P.add_i16 && T.site_tail) {                             // vcfgl.cpp:647-663
            VglSiteTail t; t.sum = 0.0f; t.sumsq = 0.0f; t.base = last_base; t.pad = 0;
            for (long long i = 0; i < info_dp; ++i) {
                int td = 1 + glibc_rand(S) / (2147483647 / (50 - 1 + 1) + 1);       // sample_from_range_rng_rand(1,50)
                if (td > 25) td = 25;                                            // CAP_TAIL_DIST
                t.sum += td; t.sumsq += (td * td);
            }
            T.site_tail[ls] = t;
        }
    }
    S->st0 = st0; S->st1 = st1;
}

// Wave-parallel scout for runs without per-read beta deviates (--error-qs 0/1).  The chains stay
// sequential, but the expensive work on them is done 64 stream positions at a time:
//   depth stream   every lane evaluates the rejection attempt that would START at its position; the
//                  chain (1 draw if em<0, else 2; accept ends a sample) is then walked with bit tests
//   base stream    the error tests of 64 consecutive reads are 64 independent compares on jumped
//                  states; only a read whose test fires (rate = error rate) is stepped sequentially
//   haplotype      one draw per read: the state of a sample's first read is a table jump
// Per-sample start states come from J^k jump tables relative to the current block.
// ordering between the lanes of the scout's single wavefront: LDS needs only program order,
// global scratch needs the writes to be visible to the other lanes' loads
__device__ __forceinline__ void scout_sync(const bool in_lds) {
    if (in_lds) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    else __threadfence();
}

__global__ __launch_bounds__(64) void k_scout_wave(const VglDevParams P, const VglTilePtrs T, VglSerialState* S) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    // jump table J^k in LDS: it sits on the critical path of every chain step
    VglAffine* stp = (VglAffine*)lds_all;
    { const VglAffine* gtab = P.step_tab; for (int i = threadIdx.x; i < 192; i += 64) stp[i] = gtab[i]; }
    uint8_t* lds_raw = lds_all + 192 * sizeof(VglAffine);
    scout_sync(true);
    const int lane = threadIdx.x;
    const int N = P.n_samples;
    const int stride = P.sample_strand ? 2 : 1;
    // per-site working set (depths, first-read indices, genotypes) lives in LDS when it fits: every
    // step of the sequential chains then waits on LDS latency, not on global memory
    const bool in_lds = (size_t)N * 9 <= (size_t)P.scout_lds_bytes;
    int32_t* dpv = in_lds ? (int32_t*)lds_raw : nullptr;
    int32_t* offv = in_lds ? (int32_t*)lds_raw + N : T.scout_off;
    uint8_t* gtl = in_lds ? (lds_raw + (size_t)8 * N) : nullptr;
    uint64_t st0 = S->st0, st1 = S->st1;                           // identical in every lane
    for (int ls = 0; ls < T.n_sites; ++ls) {
        const size_t e0 = (size_t)ls * N;
        int32_t* dps = in_lds ? dpv : (T.sdp + e0);
        const uint8_t* gts = T.gt + e0;
        if (in_lds) {
            for (int i = lane; i < N; i += 64) gtl[i] = gts[i];
            gts = gtl;
        }
        scout_sync(in_lds);
        // ---------------- depths of all samples (vcfgl.cpp:364-389)
        long long info_dp = 0;
        if (!P.per_sample_depth && !P.pois0.st12) {
            int s = 0, sw = -64;
            uint64_t missm = 0;
            while (s < N) {
                const uint64_t x1 = aff(stp[lane + 1], st1);
                const uint64_t x2 = lcg_next(x1);
                bool neg, rej; int emi;
                poisson_attempt<false>(P.pois0, x1, x2, true, P.gamma_ln_tab, P.gamma_ln_n, nullptr, neg, rej, emi);
                const uint64_t negm = __ballot(neg), rejm = __ballot(rej & !neg);
                int pos = 0;
                while (s < N && pos <= 63) {                       // scalar walk of the attempt chain
                    if ((negm >> pos) & 1) { pos += 1; continue; }
                    if ((rejm >> pos) & 1) { pos += 2; continue; }
                    if (s >= sw + 64) {                            // missing-genotype flags of the next 64 samples
                        sw = s;
                        const int sl = s + lane;
                        const uint32_t g = (sl < N) ? gts[sl] : 0u;
                        missm = __ballot((g & 0xF) == 0xF || ((g >> 4) & 0xF) == 0xF);
                    }
                    int n = __builtin_amdgcn_readlane(emi, __builtin_amdgcn_readfirstlane(pos));
                    if ((missm >> (s - sw)) & 1) n = 0;
                    if (n > P.read_cap) { if (lane == 0) atomicOr(T.errflag, VGL_DEVERR_CAPACITY); n = P.read_cap; }
                    if (lane == 0) { dps[s] = n; if (in_lds) T.sdp[e0 + s] = n; }
                    info_dp += n; ++s; pos += 2;
                }
                st1 = aff(stp[pos], st1);                   // pos <= 65 draws consumed
            }
        } else if (!P.per_sample_depth) {
            // product method (rng.h:289-297): 64 uniforms per block from jumped states, then the
            // multiplication chain t *= u walked with scalar lane reads
            int s = 0, sw = -64;
            uint64_t missm = 0;
            double t = 1.0; int em = -1;
            while (s < N) {
                const double u = u01(aff(stp[lane + 1], st1));
                const int ulo = __double2loint(u), uhi = __double2hiint(u);
                int pos = 0;
                while (s < N && pos < 64) {
                    const int p1 = __builtin_amdgcn_readfirstlane(pos);
                    const double uk = __hiloint2double(__builtin_amdgcn_readlane(uhi, p1), __builtin_amdgcn_readlane(ulo, p1));
                    ++em; t *= uk; ++pos;
                    if (!(t > P.pois0.g)) {
                        if (s >= sw + 64) {
                            sw = s;
                            const int sl = s + lane;
                            const uint32_t g = (sl < N) ? gts[sl] : 0u;
                            missm = __ballot((g & 0xF) == 0xF || ((g >> 4) & 0xF) == 0xF);
                        }
                        int n = em;
                        if ((missm >> (s - sw)) & 1) n = 0;
                        if (n > P.read_cap) { if (lane == 0) atomicOr(T.errflag, VGL_DEVERR_CAPACITY); n = P.read_cap; }
                        if (lane == 0) { dps[s] = n; if (in_lds) T.sdp[e0 + s] = n; }
                        info_dp += n; ++s; em = -1; t = 1.0;
                    }
                }
                st1 = aff(stp[pos], st1);
            }
        } else {
            for (int s = 0; s < N; ++s) {                          // per-sample means: lane-uniform
                int n;
                if (P.per_sample_depth) { const VglPois pc = P.pois[s]; n = poisson_draw_fast(pc, st1, P.gamma_ln_tab, P.gamma_ln_n); }
                else n = poisson_draw_fast(P.pois0, st1, P.gamma_ln_tab, P.gamma_ln_n);
                const uint32_t g = gts[s];
                if ((g & 0xF) == 0xF || ((g >> 4) & 0xF) == 0xF) n = 0;
                if (n > P.read_cap) { if (lane == 0) atomicOr(T.errflag, VGL_DEVERR_CAPACITY); n = P.read_cap; }
                if (lane == 0) { dps[s] = n; if (in_lds) T.sdp[e0 + s] = n; }
                info_dp += n;
            }
        }
        uint64_t thresh = P.err_thresh;
        if (T.site_tail && lane == 0) { VglSiteTail z; z.sum = 0.0f; z.sumsq = 0.0f; z.base = -1; z.pad = 0; T.site_tail[ls] = z; }
        if (info_dp == 0) { if (P.error_qs == 1 && lane == 0) T.site_thresh[ls] = thresh; continue; }
        if (P.error_qs == 1) {                                     // one beta deviate per site (vcfgl.cpp:425-437)
            uint32_t lo = 0, hi = 0;
            if (lane == 0) {
                const double pe = serial_beta(P, S);
                if (T.site_pick_err) T.site_pick_err[ls] = pe;
                const uint64_t th = (uint64_t)ceil(ldexp(pe, 48));
                T.site_thresh[ls] = th; lo = (uint32_t)th; hi = (uint32_t)(th >> 32);
            }
            lo = __shfl(lo, 0, 64); hi = __shfl(hi, 0, 64);
            thresh = ((uint64_t)hi << 32) | lo;
        }
        scout_sync(in_lds);                                        // depths written by lane 0, read by all lanes below
        // ---------------- exclusive prefix sums of the depths = first read index of each sample
        {
            int run = 0;
            for (int c0 = 0; c0 < N; c0 += 64) {
                const int s = c0 + lane;
                const int d = (s < N) ? dps[s] : 0;
                int incl = d;
#pragma unroll
                for (int k = 1; k < 64; k <<= 1) { const int t = __shfl_up(incl, k, 64); if (lane >= k) incl += t; }
                if (s < N) offv[s] = run + incl - d;
                run += __shfl(incl, 63, 64);
            }
        }
        scout_sync(in_lds);
        // ---------------- reads of the site as one sequence of R = INFO/DP reads (vcfgl.cpp:469-613)
        const int R = (int)info_dp;
        int q0 = 0, sn = 0;                                        // reads done / samples whose start state is recorded
        uint64_t hb = st1, bb = st0;                               // haplotype / base stream states before read q0
        int last_err_read = -1, last_err_base = -1;
        while (sn < N || q0 < R) {                                 // every sample recorded and every read consumed
            const int nb = (R - q0 < 64) ? (R - q0) : 64;
            bool cand = false;
            if (lane < nb) cand = aff(stp[lane * stride + 1], bb) < thresh;      // error test of read q0+lane (u < e)
            const uint64_t m = __ballot(cand);
            const int nclean = m ? (__ffsll((unsigned long long)m) - 1) : nb;
            int cnt;
            do {                                                   // samples that start inside the clean range
                const int s = sn + lane;
                const int o = (s < N) ? offv[s] : 0x7fffffff;
                const bool ok = (s < N) & (o <= q0 + nclean);
                if (ok) {
                    T.sst_hap[e0 + s] = aff(stp[o - q0], hb);
                    T.sst_base[e0 + s] = aff(stp[(o - q0) * stride], bb);
                }
                cnt = __popcll(__ballot(ok));
                sn += cnt;
            } while (cnt == 64);
            if (m == 0) {
                hb = aff(stp[nb], hb); bb = aff(stp[nb * stride], bb); q0 += nb;
            } else {
                // read q0+nclean is miscalled: its sample is the last recorded one with reads
                const int qe = q0 + nclean;
                int se = -1;
                for (int back = 0; se < 0; back += 64) {
                    const int s = sn - 1 - back - lane;
                    const bool has = (s >= 0) && (dps[s] > 0);
                    const uint64_t hm = __ballot(has);
                    if (hm) se = sn - 1 - back - (__ffsll((unsigned long long)hm) - 1);
                    else if (sn - 1 - back - 64 < 0) break;
                }
                const uint32_t g = gts[se < 0 ? 0 : se];
                const uint64_t hx = aff(stp[nclean + 1], hb);
                const int true_base = (hx < (1ULL << 47)) ? (int)(g & 0xF) : (int)((g >> 4) & 0xF);
                uint64_t x = aff(stp[nclean * stride + 1], bb);
                int rb;
                do { x = lcg_next(x); rb = (int)(x >> 46); } while (rb == true_base);      // vcfgl.cpp:487
                if (P.sample_strand) x = lcg_next(x);
                bb = x; hb = hx; q0 = qe + 1;
                last_err_read = qe; last_err_base = rb;
            }
        }
        st1 = hb; st0 = bb;
        if (P.add_i16 && T.site_tail && lane == 0) {               // vcfgl.cpp:647-663 (stale r_base = base of the last read)
            int last_base = last_err_base;
            if (last_err_read != R - 1) {
                int se = N - 1;
                while (se > 0 && dps[se] == 0) --se;
                const uint32_t g = gts[se];
                last_base = (hb < (1ULL << 47)) ? (int)(g & 0xF) : (int)((g >> 4) & 0xF);
            }
            VglSiteTail t; t.sum = 0.0f; t.sumsq = 0.0f; t.base = last_base; t.pad = 0;
            for (int i = 0; i < R; ++i) {
                int td = 1 + glibc_rand(S) / (2147483647 / (50 - 1 + 1) + 1);
                if (td > 25) td = 25;
                t.sum += td; t.sumsq += (td * td);
            }
            T.site_tail[ls] = t;
        }
    }
    if (lane == 0) { S->st0 = st0; S->st1 = st1; }
}

// parallel evaluation from the recorded states: same per-read code as k_sample
__global__ __launch_bounds__(256) void k_sample_serial(const VglDevParams P, const VglTilePtrs T) {
    const WavePos wp = wave_pos(P, T);
    if (!wp.valid) return;
    const int lane = threadIdx.x & 63;
    const int N = P.n_samples;
    const int ls = wp.ls;
    const int s = wp.chunk * 64 + lane;
    const bool active = s < N;
    const size_t ev = (size_t)ls * N + (size_t)wp.chunk * 64 + (active ? lane : 0);
    const size_t plane = (size_t)T.n_sites * N;
    int dp = 0;
    uint64_t ad4 = 0, adf4 = 0;
    uint32_t qs[4] = {0, 0, 0, 0}, qq[4] = {0, 0, 0, 0};
    if (active) {
        uint64_t st_hap = T.sst_hap[ev], st_base = T.sst_base[ev];
        const uint32_t g = T.gt[ev];
        const int a0 = g & 0xF, a1 = (g >> 4) & 0xF;
        dp = T.sdp[ev];                                            // the scout's depth draw (0 for a missing genotype)
        if (dp > P.read_cap) { atomicOr(T.errflag, VGL_DEVERR_CAPACITY); dp = P.read_cap; }   // (the scouts clamp their draws already; wave_sum_ad4's 16-bit fields rely on dp <= read_cap <= 1023)
        const uint64_t err_thresh = (P.error_qs == 1) ? T.site_thresh[ls] : P.err_thresh;
        for (int r = 0; r < dp; ++r) {
            bool fwd;
            const int r_base = sample_read_base(st_hap, st_base, a0, a1, err_thresh, P.sample_strand != 0, fwd);
            int q_i = P.pre_q, aq_i = P.pre_adjq;
            if (P.error_qs == 2) {
                double e;
                if (T.errp_lin) { e = T.errp_lin[T.roff[ev] + r]; if (T.errp) T.errp[vgl_errp_index(r, ev, P.read_cap)] = e; }   // k_gl reads the planes
                else e = T.errp[vgl_errp_index(r, ev, P.read_cap)];
                errprob_to_qs(P, e, q_i, aq_i, T.errflag);
                if (aq_i < 0 && (P.adjust_qs & 3)) atomicOr(T.errflag, VGL_DEVERR_ADJQ);               // vcfgl.cpp:558, gl_methods.cpp:101
            }
            const int q_gl = (P.adjust_qs & 1) ? aq_i : q_i;
            T.reads[vgl_read_byte(r, plane, ev)] = (uint8_t)((q_gl << 2) | r_base);
            if (T.reads_out && r < T.reads_out_cap) T.reads_out[(size_t)r * plane + ev] = (uint8_t)((q_i << 2) | r_base);
            if (P.need_qsum) {
                const uint32_t qv = (uint32_t)((P.adjust_qs & 2) ? aq_i : q_i);
                const uint32_t q2 = (uint32_t)qs_to_qssq((int)qv);
#pragma unroll
                for (int b = 0; b < 4; ++b) { qs[b] += (r_base == b) ? qv : 0u; qq[b] += (r_base == b) ? q2 : 0u; }
            }
            const uint64_t one = 1ULL << (16 * r_base);
            ad4 += one;
            if (fwd) adf4 += one;
        }
        if (!P.sample_strand) adf4 = ad4;
        if (T.fmt_dp) T.fmt_dp[ev] = dp;
        T.ad4[ev] = ad4;
        if (P.need_adf) T.adf4[ev] = adf4;
        if (P.need_qsum) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                T.qsum[((size_t)ls * 4 + b) * N + s] = qs[b];
                if (P.need_qsumsq) T.qsumsq[((size_t)ls * 4 + b) * N + s] = qq[b];
            }
        }
        if (T.reads_out) for (int r = dp; r < T.reads_out_cap; ++r) T.reads_out[(size_t)r * plane + ev] = 0xFF;
    }
    // every read of an evaluation shows one base: INFO/DP is the sum of the four per-base totals; without strand draws every read is "forward"
    int v[9];
    wave_sum_ad4(ad4, &v[1]);
    if ((P.sample_strand != 0)) wave_sum_ad4(adf4, &v[5]);
    else { v[5] = v[1]; v[6] = v[2]; v[7] = v[3]; v[8] = v[4]; }
    v[0] = v[1] + v[2] + v[3] + v[4];
    if (lane == 0) {
        int32_t* acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
#pragma unroll
        for (int k = 0; k < 9; ++k) if (v[k]) atomicAdd(&acc[k], v[k]);
    }
    if (P.need_qsum) wave_add_qsum_totals(T.acc + (size_t)ls * VGL_ACC_STRIDE, lane, qs, qq, P.need_qsumsq != 0);
}

// ------------------------------------------------------------------------------------
extern "C" int vgl_launch_scout(const VglDevParams* p, const VglTilePtrs* t, VglSerialState* st, void* stream) {
    if (t->n_sites == 0) return 0;
    if (p->error_qs != 2 || p->beta_chain) hipLaunchKernelGGL(k_scout_wave, dim3(1), dim3(64), (size_t)p->scout_lds_bytes + 192 * sizeof(VglAffine), (hipStream_t)stream, *p, *t, st);
    else hipLaunchKernelGGL(k_scout, dim3(1), dim3(64), 0, (hipStream_t)stream, *p, *t, st);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_sample_serial(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    const int64_t waves = (int64_t)t->n_sites * p->chunks;
    if (waves == 0) return 0;
    hipLaunchKernelGGL(k_sample_serial, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}
