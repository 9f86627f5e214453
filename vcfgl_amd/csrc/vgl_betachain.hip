// vgl_betachain.hip -- VGL_RNG_SERIAL with --error-qs 2 and the default (std::mt19937) beta sampler.
//
// The reference draws one beta deviate per read from ONE std::mt19937 (rng.h:353-421): read j gets the
// j-th deviate, and a deviate consumes a data-dependent number of generator words (libstdc++
// gamma_distribution: polar normal pairs + rejection).  The rand48 streams do not depend on these
// deviates (the error test uses --error-rate, vcfgl.cpp:486), so the wave scout resolves them as for
// --error-qs 0 and this file resolves the beta chain, per chunk of the generator's output:
//   k_mt_fill      the generator's next words into W (one workgroup: the 624-word twist is parallel
//                  inside a block, blocks are sequential); state snapshots every VGL_MT_SNAP blocks
//   k_beta_cons    for EVERY even word position p: the number of words a deviate starting at p consumes
//   k_chain_walk   position p -> p + cons(p) is a chain; a wavefront per segment of VGL_SEG positions
//                  follows it from each of the 64 possible entry positions (lane = entry): exit, count
//   k_chain_stitch one lane composes the segments' (entry -> exit, count) maps in order
//   k_chain_walk   again, from each segment's true entry: the start position of every read's deviate
//   k_beta_emit    the deviates at those positions, one lane per read -> errp_lin[read]
//   k_mt_advance   the persistent generator state moved behind the last consumed word
// Every step but the stitch is parallel over the chunk; the results equal the serial program's.
#include "vgl_common.hip.h"

#define VGL_MT_SNAP 256            // generator blocks between state snapshots
#define VGL_SEG 32768              // chain positions (= 2 generator words each) per segment
#define VGL_BETA_MAXW 126          // words one deviate may consume in this scheme (a 64-entry window); more -> error

// ---- std::generate_canonical<double,53> over a window of generator words ---------------------------
struct WordSrc {
    const uint32_t* w; uint32_t pos, end; bool ovf;
    __device__ double canonical() {
        if (pos + 2 > end) { ovf = true; return 0.25; }
        const double lo = (double)w[pos], hi = (double)w[pos + 1];
        pos += 2;
        double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
        if (r >= 1.0) r = 0x1.fffffffffffffp-1;
        return r;
    }
};
// std::gamma_distribution<double>(alpha,1) on a fresh object (rng.h:409-412; vgl_serial.hip: std_gamma_fresh),
// written as one flat loop over candidate normal deviates: the nested rejection loops of the library code,
// evaluated by 64 lanes at once, came out of the compiler with a lane's consumption depending on its
// neighbours (a single lane gave the right answer); a flat loop with one back edge does not.
__device__ double gamma_from_words(WordSrc& S, const double alpha) {
    const double malpha = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double a1 = malpha - 1.0 / 3.0;
    const double a2 = 1.0 / sqrt(9.0 * a1);
    bool saved_avail = false; double saved = 0.0;
    double v = 0.0;
    bool done = false;
    while (!done && !S.ovf) {
        // one candidate normal deviate: the saved half of the last polar pair, or a new pair
        double n; bool have_n;
        if (saved_avail) { saved_avail = false; n = saved; have_n = true; }
        else {
            const double x = 2.0 * S.canonical() - 1.0;
            const double y = 2.0 * S.canonical() - 1.0;
            const double r2 = x * x + y * y;
            have_n = !(r2 > 1.0 || r2 == 0.0);
            const double mult = have_n ? sqrt(-2 * log(r2) / r2) : 0.0;
            saved = x * mult; saved_avail = have_n;
            n = y * mult;
        }
        const double w = 1.0 + a2 * n;
        if (have_n && w > 0.0) {
            v = w * w * w;
            const double u = S.canonical();
            const bool reject = u > 1.0 - 0.0331 * n * n * n * n && (log(u) > (0.5 * n * n + a1 * (1.0 - v + log(v))));
            done = !reject;
        }
    }
    if (alpha == malpha) return a1 * v;
    double u;
    do u = S.canonical(); while (!S.ovf && u == 0.0);
    return pow(u, 1.0 / alpha) * a1 * v;
}
__device__ double beta_from_words(const VglDevParams& P, WordSrc& S) {
    const double x = gamma_from_words(S, P.beta_a);
    const double y = gamma_from_words(S, P.beta_b);
    return x / (x + y);
}

// ---- read offsets: exclusive prefix sum of the scout's depths in (site, sample) order ------------------
__global__ __launch_bounds__(1024) void k_read_offsets(const int32_t* __restrict__ sdp, long long n, long long* __restrict__ roff, long long* total) {
    __shared__ long long s_w[16];
    __shared__ long long s_run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (long long base = 0; base < n; base += 1024) {
        const long long i = base + tid;
        const long long v = (i < n) ? (long long)sdp[i] : 0;
        long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        long long woff = 0;
        for (int k = 0; k < wv; k++) woff += s_w[k];
        const long long run = s_run;
        if (i < n) roff[i] = run + woff + incl - v;
        __syncthreads();
        if (tid == 1023) s_run = run + woff + incl;
        __syncthreads();
    }
    if (tid == 0) *total = s_run;
}

// ---- mt19937: block-parallel twist, tempering -------------------------------------------------------
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ uint32_t mt_mix(const uint32_t a, const uint32_t b, const uint32_t m) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// mt[i] <- mt[(i+397)%624] ^ f(mt[i], mt[(i+1)%624]) for i = 0..623 in order: elements 0..226 read only old
// values, 227..453 read new values of 0..226, 454..622 new values of 227..395, 623 new values of 0 and 396
__device__ void mt_twist_block(uint32_t* mt, const int tid, const int nthr) {
    uint32_t nv[2]; int cnt;
    // every phase first reads (neighbours that the same phase rewrites are still old), then writes
    cnt = 0; for (int i = tid; i < 227; i += nthr) nv[cnt++] = mt_mix(mt[i], mt[i + 1], mt[i + 397]);
    __syncthreads();
    cnt = 0; for (int i = tid; i < 227; i += nthr) mt[i] = nv[cnt++];
    __syncthreads();
    cnt = 0; for (int i = 227 + tid; i < 454; i += nthr) nv[cnt++] = mt_mix(mt[i], mt[i + 1], mt[i - 227]);
    __syncthreads();
    cnt = 0; for (int i = 227 + tid; i < 454; i += nthr) mt[i] = nv[cnt++];
    __syncthreads();
    cnt = 0; for (int i = 454 + tid; i < 623; i += nthr) nv[cnt++] = mt_mix(mt[i], mt[i + 1], mt[i - 227]);
    __syncthreads();
    cnt = 0; for (int i = 454 + tid; i < 623; i += nthr) mt[i] = nv[cnt++];
    __syncthreads();
    if (tid == 0) mt[623] = mt_mix(mt[623], mt[0], mt[396]);
    __syncthreads();
}

// W[0..n_words) = the next n_words outputs after the persistent state (which is left untouched);
// snap[k] = state array before the (k * VGL_MT_SNAP)-th twist of this call, snap_words[k] = words produced before it
__global__ __launch_bounds__(256) void k_mt_fill(const VglSerialState* S, uint32_t* __restrict__ W, const long long n_words,
                                                 uint32_t* __restrict__ snap, long long* __restrict__ snap_words, long long* n_snap_out) {
    __shared__ uint32_t mt[624];
    const int tid = threadIdx.x;
    for (int i = tid; i < 624; i += 256) mt[i] = S->mt[i];
    __syncthreads();
    int idx = S->mt_idx;
    long long produced = 0, blk = 0;
    while (true) {
        const long long left = n_words - produced;
        const int n = (int)((624 - idx < left) ? (624 - idx) : left);
        for (int t = tid; t < n; t += 256) W[produced + t] = mt_temper(mt[idx + t]);
        produced += n; idx += n;
        if (produced >= n_words) break;
        if (blk % VGL_MT_SNAP == 0) {
            const long long k = blk / VGL_MT_SNAP;
            for (int i = tid; i < 624; i += 256) snap[k * 624 + i] = mt[i];
            if (tid == 0) snap_words[k] = produced;
        }
        __syncthreads();
        mt_twist_block(mt, tid, 256);
        idx = 0; blk++;
    }
    if (tid == 0) *n_snap_out = (blk + VGL_MT_SNAP - 1) / VGL_MT_SNAP;
}

// the persistent state moved forward by exactly endw words
__global__ __launch_bounds__(256) void k_mt_advance(VglSerialState* S, const VglChainCtl* ctl, const uint32_t* __restrict__ snap,
                                                    const long long* __restrict__ snap_words, const long long* n_snap_in) {
    __shared__ uint32_t mt[624];
    const int tid = threadIdx.x;
    const long long endw = ctl->endw, n_snap = *n_snap_in;
    long long k = -1;
    for (long long j = 0; j < n_snap; j++) if (snap_words[j] <= endw) k = j; else break;
    int idx; long long at;
    if (k < 0) { for (int i = tid; i < 624; i += 256) mt[i] = S->mt[i]; idx = S->mt_idx; at = 0; }
    else { for (int i = tid; i < 624; i += 256) mt[i] = snap[k * 624 + i]; idx = 624; at = snap_words[k]; }
    __syncthreads();
    while (at < endw) {
        if (idx >= 624) { mt_twist_block(mt, tid, 256); idx = 0; }
        const long long left = endw - at;
        const int n = (int)((624 - idx < left) ? (624 - idx) : left);
        idx += n; at += n;
    }
    __syncthreads();
    for (int i = tid; i < 624; i += 256) S->mt[i] = mt[i];
    if (tid == 0) S->mt_idx = idx;
}

// ---- words consumed by a deviate starting at every even word position ----------------------------------
__global__ __launch_bounds__(256) void k_beta_cons(const VglDevParams P, const uint32_t* __restrict__ W, const long long n_pos,
                                                   uint8_t* __restrict__ cons) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pos) return;
    WordSrc s; s.w = W + 2 * p; s.pos = 0; s.end = VGL_BETA_MAXW; s.ovf = false;
    (void)beta_from_words(P, s);
    cons[p] = s.ovf ? (uint8_t)255 : (uint8_t)(s.pos >> 1);
}

// ---- the chain inside one segment ---------------------------------------------------------------------
// MODE 0: lane = entry position 0..63 of the segment -> exit position relative to the next segment, reads started
// MODE 1: lane 0 from the true entry: pos_out[base + k] = global position of the k-th read that starts here
template <int MODE>
__global__ __launch_bounds__(64) void k_chain_walk(const uint8_t* __restrict__ cons, VglChainCtl* ctl, uint8_t* __restrict__ seg_exit,
                                                   int32_t* __restrict__ seg_cnt, const uint8_t* __restrict__ seg_entry,
                                                   const long long* __restrict__ seg_base, uint32_t* __restrict__ pos_out) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const long long p0 = (long long)g * VGL_SEG;
    const long long n_pos = ctl->n_pos;
    const int len = (int)((n_pos - p0 < VGL_SEG) ? (n_pos - p0) : VGL_SEG);
    if (MODE == 0) {
        int p = lane, cnt = 0; bool bad = false;
        while (p < len) {
            const int c = cons[p0 + p];
            if (c == 255) { bad = true; break; }
            p += c; cnt++;
        }
        const int ex = p - len;
        if (bad || ex > 63) { atomicExch(&ctl->err, 1); }
        seg_exit[(size_t)g * 64 + lane] = (uint8_t)(ex > 63 ? 63 : (ex < 0 ? 0 : ex));
        seg_cnt[(size_t)g * 64 + lane] = cnt;
    } else {
        if (lane != 0 || g > ctl->last_seg) return;
        const long long base = seg_base[g], want = ctl->n_chunk;
        int p = seg_entry[g]; long long k = base;
        while (p < len && k < want) {
            pos_out[k] = (uint32_t)(p0 + p);
            const int c = cons[p0 + p];
            p += c; k++;
            if (k == want) ctl->endw = 2 * (p0 + p);          // start of the first deviate this chunk does not serve
        }
    }
}

__global__ void k_chain_stitch(VglChainCtl* ctl, const uint8_t* __restrict__ seg_exit, const int32_t* __restrict__ seg_cnt,
                               uint8_t* __restrict__ seg_entry, long long* __restrict__ seg_base) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int e = 0; long long cum = 0;
    const long long remaining = ctl->remaining;
    int last = ctl->n_seg - 1;
    for (int g = 0; g < ctl->n_seg; g++) {
        seg_entry[g] = (uint8_t)e; seg_base[g] = cum;
        const long long c = seg_cnt[(size_t)g * 64 + e];
        if (cum + c >= remaining) { cum = remaining; last = g; break; }
        cum += c;
        e = seg_exit[(size_t)g * 64 + e];
    }
    ctl->n_chunk = cum < remaining ? cum : remaining;
    ctl->last_seg = last;
    if (cum < remaining) {                                                // the chunk ends before the reads do: next chunk
        const long long full = (long long)ctl->n_seg * VGL_SEG;           // starts where the chain leaves this one
        ctl->endw = 2 * ((ctl->n_pos < full ? ctl->n_pos : full) + e);
    }
}

// ---- the deviates at the chain positions -----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_beta_emit(const VglDevParams P, const uint32_t* __restrict__ W, const uint32_t* __restrict__ pos,
                                                   const long long n, double* __restrict__ out) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    WordSrc s; s.w = W + 2 * (long long)pos[k]; s.pos = 0; s.end = VGL_BETA_MAXW; s.ovf = false;
    out[k] = beta_from_words(P, s);
}

// ---- host entry points (one stream; the caller synchronises where it reads ctl back) -------------------
extern "C" int vgl_chain_read_offsets(const int32_t* sdp, long long n, long long* roff, long long* total, void* stream) {
    hipLaunchKernelGGL(k_read_offsets, dim3(1), dim3(1024), 0, (hipStream_t)stream, sdp, n, roff, total);
    return (int)hipGetLastError();
}
extern "C" long long vgl_chain_snapshots_needed(long long n_words) { return n_words / 624 / VGL_MT_SNAP + 2; }
extern "C" int vgl_chain_seg() { return VGL_SEG; }
extern "C" int vgl_chain_margin_words() { return VGL_BETA_MAXW + 2; }
extern "C" int vgl_chain_chunk(const VglDevParams* p, VglSerialState* S, VglChainCtl* ctl, uint32_t* W, long long n_words, uint8_t* cons,
                               uint8_t* seg_exit, int32_t* seg_cnt, uint8_t* seg_entry, long long* seg_base, uint32_t* pos,
                               uint32_t* snap, long long* snap_words, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const long long n_pos = (n_words - (VGL_BETA_MAXW + 2)) / 2;           // positions whose deviate surely fits inside W
    const int n_seg = (int)((n_pos + VGL_SEG - 1) / VGL_SEG);
    hipLaunchKernelGGL(k_mt_fill, dim3(1), dim3(256), 0, st, (const VglSerialState*)S, W, n_words, snap, snap_words, snap_words + vgl_chain_snapshots_needed(n_words));
    hipLaunchKernelGGL(k_beta_cons, dim3((unsigned)((n_pos + 255) / 256)), dim3(256), 0, st, *p, (const uint32_t*)W, n_pos, cons);
    hipLaunchKernelGGL(k_chain_walk<0>, dim3(n_seg), dim3(64), 0, st, (const uint8_t*)cons, ctl, seg_exit, seg_cnt, (const uint8_t*)seg_entry,
                       (const long long*)seg_base, pos);
    hipLaunchKernelGGL(k_chain_stitch, dim3(1), dim3(64), 0, st, ctl, (const uint8_t*)seg_exit, (const int32_t*)seg_cnt, seg_entry, seg_base);
    hipLaunchKernelGGL(k_chain_walk<1>, dim3(n_seg), dim3(64), 0, st, (const uint8_t*)cons, ctl, seg_exit, seg_cnt, (const uint8_t*)seg_entry,
                       (const long long*)seg_base, pos);
    return (int)hipGetLastError();
}
extern "C" int vgl_chain_emit(const VglDevParams* p, VglSerialState* S, const VglChainCtl* ctl, const uint32_t* W, const uint32_t* pos,
                              long long n_chunk, double* out, const uint32_t* snap, const long long* snap_words, long long n_snap_cap, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_chunk > 0) hipLaunchKernelGGL(k_beta_emit, dim3((unsigned)((n_chunk + 255) / 256)), dim3(256), 0, st, *p, W, pos, n_chunk, out);
    hipLaunchKernelGGL(k_mt_advance, dim3(1), dim3(256), 0, st, S, ctl, snap, snap_words, snap_words + n_snap_cap);
    return (int)hipGetLastError();
}
