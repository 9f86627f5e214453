// vgl_device.h -- structures shared by the host side of the C ABI (vgl_host.cpp) and the
// gfx950 kernels (vgl_sample.hip, vgl_serial.hip, vgl_gl.hip).
#pragma once
#include <stdint.h>

#define VGL_MASK48 0xFFFFFFFFFFFFULL
#define VGL_LCG_A  0x5DEECE66DULL
#define VGL_LCG_C  0xBULL
#define VGL_WAVE   64
#define VGL_MAX_QS_BINS 32
#define VGL_DEPTH_CHUNK 1024     // evaluations dealt to one wavefront of k_depth: 1024, 2048 or 4096 (vgl_launch_depth: the larger the tile, the
                                 // larger the chunk -- the chunk's last items keep a wavefront going with few busy lanes)
#define VGL_DEPTH_CHUNK_MAX 4096
// htslib's errmod_cal() subsamples a pileup deeper than 255 reads with ks_shuffle() on htslib's OWN rand48 generator
// (hts_drand48: never seeded by its callers, state {0x330e, 0xabcd, 0x1234}); GL model 1 reaches it with --depth > ~200
#define VGL_HTS_RAND48_X0 0x1234ABCD330EULL
#define VGL_HTS_TILE_STRIDE 1024ULL   // VGL_RNG_TILE: evaluation e owns draws [e * 1024, (e + 1) * 1024) of that stream
// -addI16 in VGL_RNG_TILE: the reference draws the tail distances (vcfgl.cpp:647-663) from the never-seeded libc rand(), one serial stream; tile mode
// takes them from a SECOND rand48 sequence (started at VGL_TAIL_RAND48_X0 -- not a state any --seed produces: those end in 0x330E) addressed exactly as
// the first: evaluation e = H(site) N + sample owns draws [e block, (e + 1) block) of it, read r = draw r of the window, as a 31-bit integer (the
// state's top 31 bits, as lrand48) through the reference's range formula (rng.h:12).  block >= read_cap, and the windows fill at most one period (W).
#define VGL_TAIL_RAND48_X0 0x7A11D157A11DULL
// per-site accumulators of those draws (int32 slots of VglTilePtrs::acc): [9] sum of the tail distances, [10..11] sum of their squares (one 64-bit word),
// [12] max over the site's evaluations with reads of (sample + 1) << 2 | base of the evaluation's last read -- the reference's stale r_base
#define VGL_ACC_TAIL 9
#define VGL_ACC_TAILSQ 10
#define VGL_ACC_TAILKEY 12

// x -> a*x + c (mod 2^48): a power of the rand48 step
struct VglAffine { uint64_t a, c; };

// VGL_RNG_TILE, where the windows of a site lie in the rand48 sequence: site index -> H(site), a bijection of [0, 2^W)
// (xorshift / odd multiply / xorshift / odd multiply / xorshift on W bits, every step invertible; H(0) = 0, so site 0 / sample 0
// still starts on the reference's first draws).  Evaluation (site, sample) owns the draws [e block, (e + 1) block) with
// e = H(site) * n_samples + sample.  Without H, sites 2^k apart sit at offsets 2^k * N * block of ONE linear congruential
// sequence mod 2^48, and a^(2^k) = 1 mod 2^(k+2) makes the states at sites s, s + 2^k, s + 2^(k+1) linearly dependent (for
// k >= 19 at 1000 samples, depth 20: u(s) - 2 u(s + 2^k) + u(s + 2^(k+1)) is one constant for every sample and read).  H removes
// the regular spacing: the windows of neighbouring sites and of sites 2^k apart are at pseudo-random offsets of each other.
// W = floor(log2(2^48 / block / n_samples)): the 2^W site groups fill at most the generator's period.
// The oracle (oracle/vgl_oracle.c: site_hash) and the tests restate this function; vgl_rng_tile_site_hash() exports it.
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint64_t vgl_site_hash(uint64_t x, const int W) {
    if (W <= 1) return x;
    const uint64_t mask = (1ULL << W) - 1ULL;
    const int sh = (W + 1) >> 1;
    x ^= x >> sh; x = (x * 0xBF58476D1CE4E5B9ULL) & mask;
    x ^= x >> sh; x = (x * 0x94D049BB133111EBULL) & mask;
    x ^= x >> sh;
    return x;
}

// PoissonSampler (rng.h:249-280), one per run or one per sample
// sqf, lmf: sq and lm rounded to float32; e_hi: an attempt of the rejection method whose sq tan(..) + lm is at least e_hi is rejected
// by every acceptance draw >= 2^-32 (poisson_fast, vgl_common.hip.h; pois_init, vgl_host.cpp)
struct VglPois { double lm, sq, alxm, g; float sqf, lmf, e_hi; int32_t st12; };

// Gamma1Sampler (rng.h:122-173)
struct VglGamma1 { double alpha0, a1, a2; int32_t changed; int32_t pad; };

// Device error flag bits (sticky, OR-ed by kernels)
// Staged reads (VglTilePtrs::reads, 1 byte per read: qScore << 2 | base): read r of evaluation ev is byte (r & 3) of the 32-bit
// word [(r >> 2) * plane + ev], plane = n_sites * n_samples -- four reads of an evaluation per word, so that the wavefronts of
// k_sample store and those of k_gl load whole 256-byte rows per instruction.  read_cap is a multiple of 4.
#if defined(__HIPCC__)
__device__ __forceinline__ size_t vgl_read_byte(const int r, const size_t plane, const size_t ev) {
    return ((((size_t)(r >> 2)) * plane + ev) << 2) | (size_t)(r & 3);
}
#endif
// Staged error probabilities (VglTilePtrs::errp, --precise-gl 1 and the deviate dumps): EVALUATION-major since round 6 -- read r of evaluation ev at
// [ev * read_cap + r].  The lane of k_sample<2> that finishes a read stores its probability; the reads of one evaluation are neighbours in the
// wavefront's pool and are finished within a few iterations of one another, so their 8-byte stores now fall into the same one or two cache lines
// while those are still in L2 (read-major, [r][site][N], the sixteen stores of a line were spread over the whole pool loop: the partial-line
// write-backs cost k_sample<2, PREC> a third of its time).  k_gl reads 8 bytes per lane and read, consecutive reads from the same line.
#if defined(__HIPCC__)
__device__ __forceinline__ size_t vgl_errp_index(const int r, const size_t ev, const int read_cap) { return ev * (size_t)read_cap + (size_t)r; }
#endif
#define VGL_REDO_PARTS 64
#define VGL_REDO_STRIDE 32
#define VGL_DEVERR_CAPACITY 1u
#define VGL_DEVERR_QSBIN    2u
#define VGL_DEVERR_GL1DEPTH 4u
#define VGL_DEVERR_ADJQ     8u
#define VGL_DEVERR_INTERNAL 16u   /* a layout assumption of a kernel does not hold (k_sample<2>: dynamic LDS must start at offset 0) */

// per-site accumulator layout (int32 x 24): [0] INFO/DP, [1..4] ACGT depth, [5..8] forward-strand ACGT depth;
// with -addQS / -addI16 (uint32 bit patterns, modulo 2^32): [16..19] per-base sum over the samples of the quality sums (acgt_fmt_qsum_arr),
// [20..23] of the squared-quality sums -- the integer totals k_siteagg turns into INFO/I16 fields 5-8 where the reference's float32
// running sums stay exact (below 2^24)
#define VGL_ACC_STRIDE 24
#define VGL_ACC_QSUM 16
#define VGL_ACC_QSUMSQ 20

// per-site record produced by k_site for k_gl (one 16-byte row)
struct VglSiteInfo {
    int32_t status;
    int32_t n_alleles;
    uint32_t acgt2alleles;   // 5 nibbles (A,C,G,T,NONREF), 0xF = not present
    uint32_t alleles2acgt;   // 5 nibbles, 0xF = none, 4 = NONREF
};

struct VglDevParams {
    // geometry
    int32_t n_samples;       // N
    int32_t chunks;          // ceil(N / 64): wavefronts per site
    int32_t A, G;            // max alleles / genotypes of the tile layout
    int32_t read_cap;        // staged reads per (site,sample)
    int32_t pool_cap;        // quality-score work items per wavefront and LDS segment
    int32_t pool_lds_bytes;  // LDS bytes per wavefront of k_sample<2>: 512 + 4 * (pool_cap + 2) + pool_cap
    // flags
    int32_t error_qs, gl_model, precise_gl, adjust_qs, n_qs_bins, do_unobserved;
    int32_t rm_invar_sites, rm_empty_sites, sample_strand, per_sample_depth;
    int32_t need_qsum, need_qsumsq, need_adf, i16_mapq;
    int32_t add_i16;
    int32_t gl1_deep;        // GL model 1 and the staging capacity exceeds 255 reads: reads are staged (also with one fixed
                             // quality score) so that k_gl can subsample a deep evaluation the way errmod_cal() does
    int32_t stage_fixed;     // k_sample<0 / 1> stages its reads: GL model 2, gl1_deep, or -addI16 in tile mode (k_tail reads the last read's base)
    int32_t serial;          // VGL_RNG_SERIAL
    int32_t scout_lds_bytes; // dynamic LDS of k_scout_wave (9 bytes per sample when the site fits)
    int32_t beta_std;        // VGL_BETA_STD (serial only)
    int32_t beta_chain;      // serial --error-qs 2 with the std sampler: wave scout + vgl_betachain.hip instead of the one-lane scout
    double  beta_a, beta_b;  // beta shape parameters (std sampler)
    int32_t depth_pre;       // depth mode of k_sample: 1 = drawn by k_depth ahead of it (every mean depth >= 12: rejection method),
                             // 2 = product method in place (every mean depth < 12), 0 = mixed means, general sampler in place
    int32_t gl_sort;         // k_gl: re-deal the lanes of a workgroup in depth order (pays at depth >= 8; below, natural
                             // order keeps every store of a wavefront one contiguous segment and the kernel is HBM bound)
    int32_t gl_flip2;        // k_gl sort: the two-base group in ascending depth order (VGL_GL_FLIP2, default 1)
    int32_t fused;           // tile mode, GL model 2, one fixed quality score, default tag surface, every mean depth < 12, 128 < N <= 512: sampling, site order and
                             // likelihoods of a site in ONE workgroup (k_gl<.., FUSED>; VGL_NO_FUSE=1 turns it off)
    int32_t fused_split;     // fused build: workgroups per site (1: the workgroup is the site)
    int32_t lean_ok;         // the tag surface needs none of the optional per-read state of k_sample's owners (quality sums, strand draws, --adjust-qs):
                             // the LEAN builds serve every tile that asks for no per-read dump
    int32_t gl_wpb;          // k_gl, GL model 2: wavefronts per workgroup, 4 or 8 (VGL_GL_WPB)
    int32_t slow_period;     // k_sample<2>: the bounded-log test of the gamma sampler runs every slow_period-th pool iteration,
    int32_t slow_period_n;   //              that of the normal sampler every slow_period_n-th
    int32_t xcd_map;         // k_gl: workgroup index -> XCD-contiguous logical index (VGL_XCD_MAP=0 turns it off; k_sample, which is
                             // bound by its arithmetic, measured 1-3 % slower with it and keeps the hardware order)
    int32_t dbg_depth_chunk; // test hook (VGL_DEPTH_CHUNK=1024 / 2048 / 4096): k_depth's chunk whatever the tile's size
    int32_t dbg_gl2_ovc;     // test hook (VGL_DEBUG_GL2_OVC=n): a workgroup of k_gl2 with more than n three- / four-base evaluations goes to k_gl_redo (default: VGL_GL2_OVC)
    int32_t dbg_fuse_alone;  // test hook (VGL_DEBUG_FUSE_ALONE=mask): the parts in the mask count as neighbours that never arrive -- the other workgroups of the site do not wait for them and sample their depths themselves
    int32_t dbg_phase;       // diagnostic (VGL_DEBUG_PHASE=n): k_sample returns after phase n; 0 = off
    int32_t dbg_qs_exact;    // test hook (VGL_DEBUG_QS_EXACT=1): k_sample<2> treats every read as undecided in float32
    int32_t dbg_redo_every;  // test hook (VGL_DEBUG_REDO_EVERY=k): the deferred build sends every k-th read and slow-test lane to k_redo
    int32_t dbg_stamps;      // diagnostic (VGL_DEBUG_STAMPS): 1 = the stamped inline-fallback build of k_sample<2> (LEAN 0), 2 = the stamped float32 build of the default tag surface (LEAN 2)
    int32_t qsum_lds;        // k_sample<2, LEAN 3>: the owners' quality sums are gathered in LDS by the dense pass (one atomic per read) -- --adjust-qs 0 or 3
                             // (one score for likelihoods and sums) and at most 130 staged reads (sum of squares << 13 | sum: 130 x 63 = 8190 < 2^13, 130 x 63^2 < 2^19)
    int32_t defer_ok;        // the flag set allows the deferred build of k_sample<2> (vgl_ctx_create; VGL_NO_DEFER=1 turns it off)
    int32_t seg_split;       // k_sample<2, LEAN 2> without --precise-gl 1 runs as k_sample_seg<., 1> + <., 2>: the pool holds a wavefront's summed depth + 8 sigma (VGL_NO_SEG_SPLIT=1 turns it off)
    int32_t seg_limit;       // k_sample_seg<., 1>: a wavefront with more reads goes to the list (= pool_cap; test hook VGL_DEBUG_SEG_LIMIT)
    int32_t pre_q, pre_adjq;           // preCalc qScore / adj_qScore (vcfgl.cpp:1697-1702)
    double  adjust_by;
    double  pre_homT, pre_het, pre_homF;
    uint64_t err_thresh;               // ceil(error_rate * 2^48): u < e  <=>  X < err_thresh
    // rand48 addressing
    uint64_t x0;
    VglAffine off[4];                  // J^(off[k])
    VglAffine site_pow[40];            // J^(block * N * 2^b)  (k_sitebase)
    int32_t out_layout;                // VGL_LAYOUT_*: k_gl stores the multi-valued FORMAT arrays as planes or sample-major
    int32_t site_hash_bits;            // W of vgl_site_hash(): sites [0, 2^W) are addressable
    uint32_t depth_magic;              // k_depth, 2 <= N < chunk: floor(2^32 / N) + 1, so that t / N = mulhi(t, magic) for t < N + chunk <= 2 VGL_DEPTH_CHUNK_MAX (t N < 2^32)
    const VglAffine* samp_tab;         // [N] J^(block * s)
    const VglAffine* qs_read_tab;      // [read_cap] J^(qs_read_stride * r)
    const VglAffine* step_tab;         // [192] J^k (serial-mode scout)
    const VglAffine* depth_tab;        // [N] J^(off[0] + block * s)       (k_depth)
    // samplers
    VglPois pois0;
    const VglPois* pois;               // [N] when per_sample_depth
    VglGamma1 gx, gy;
    double sure_margin;                // absolute slack of k_sample<2>'s sure-accept bound (host: 1e-9 + 1e-14 max a1)
    int32_t qs_bins[VGL_MAX_QS_BINS * 3];
    // tables
    const double* q2gl;                // [3][257]
    const double* gamma_ln_tab;        // [gamma_ln_n] gamma_ln(k), k >= 1 (entry 0 unused)
    int32_t gamma_ln_n;
    const float* pois_zt;              // [gamma_ln_n - 1] one mean depth >= 12 for all samples: (float)((k alxm - gamma_ln(k + 1) - g) log2 e), the exponent of the
                                       // rejection method's acceptance bound (rng.h:308) ready for v_exp_f32 (poisson_fast); null with per-sample depths
    const float* gl2_run;              // [2][read_cap + 1][3] GL model 2 with ONE fixed score: the accumulators (hom present, present / absent, absent / absent) of an
                                       // evaluation whose n reads all show the same base -- the reference's n update / subtract-the-maximum steps
                                       // (gl_methods.cpp:22-59) run once on the host per n instead of once per evaluation; [0] the site has an allele the
                                       // evaluation lacks, [1] it has not (only the first accumulator then enters the maximum).  Null with per-read scores.
    const double* gl1_bsum;            // [256][256]  sum_{i<c} fk[i]*beta[q][n][i]   (fixed qScore)
    const double* gl1_lhet;            // [256][256]
    const double* gl1_fkbeta;          // [60][gl1_nc][gl1_nc] fk[i] * beta[q][n][i] at [q - 4][n][i]  (per-read qScores only)
    int32_t gl2x;                      // GL model 2, three-kernel path: k_gl2 (two evaluations per thread, compact accumulator rows) instead of k_gl
    int32_t gl1_nc;                    // min(255, read_cap) + 1: the table is compact in n and i, so that the part a run touches stays in L2
};

// per-site I16 tail-distance sums (vcfgl.cpp:647-663): all of a site's draws are credited to the base of
// its last simulated read (the reference's stale r_base)
struct VglSiteTail { float sum, sumsq; int32_t base; int32_t pad; };

// per-tile pointers
struct VglTilePtrs {
    int64_t site0;
    int32_t n_sites;
    const uint8_t* gt;
    // staging / scratch (ctx owned)
    uint8_t* reads;          // [read_cap / 4][n_sites][N] words, four reads each (vgl_read_byte)
    unsigned long long* redo_list;   // k_sample<2, deferred>: (evaluation << 10 | read) of the reads whose quality score k_redo draws in double --
                             // VGL_REDO_PARTS partitions of redo_cap entries each; a wavefront appends to partition (wave index mod VGL_REDO_PARTS)
    uint32_t* redo_count;    // entries appended this tile, one counter per partition at redo_count[VGL_REDO_STRIDE * p] (a cache line each: round 5 --
                             // ONE counter took 8.6e5 returning atomics per tile, and same-address atomics are served one at a time: they, not the
                             // sampling, set k_sample<2>'s time).  A count may exceed redo_cap: the rest is marked in redo_bits
    uint32_t redo_cap;       // entries per partition
    uint32_t* redo_bits;     // one bit per (evaluation, read) of the tile, all zero between tiles: overflow of the list
    uint32_t* seg_list;      // k_sample_seg: the wavefronts (index in the tile) whose reads need more than one pool segment; their number is redo_count[1]
    double*  errp;           // [n_sites][N][read_cap] evaluation-major (vgl_errp_index): --precise-gl 1 with error_qs 2, and the deviate dumps
    uint64_t* ad4;           // [n_sites][N]  4 x u16 ACGT depth
    uint64_t* adf4;          // [n_sites][N]  4 x u16 forward-strand depth
    uint32_t* qsum;          // [n_sites][4][N]
    uint32_t* qsumsq;        // [n_sites][4][N]
    int32_t*  acc;           // [n_sites][16]
    VglSiteInfo* sinfo;      // [n_sites]
    uint32_t* gl2_redo_list; // k_gl2_scan: the workgroups whose bit was set ...
    uint32_t* gl2_redo_count;// ... and their number (zeroed before k_gl2)
    uint32_t* gl2_redo;      // one bit per workgroup of k_gl2 (1024 evaluations), zero between tiles: the workgroup could not keep its accumulators
    uint64_t* rowmap8;       // [n_sites][16][2] the same table as byte codes of k_gl2's compact accumulator rows (null: k_gl2 is not used)
    uint64_t* rowmap;        // [n_sites][16] GL model 2: for every set of bases an evaluation may show (4 bits), the accumulator row of each of
                             // the site's genotypes, 4 bits per genotype in bcf_alleles2gt order (k_site writes it, k_gl's epilogue reads one entry per lane)
    uint32_t* errflag;
    int32_t*  dp_pre;        // [n_sites][N] depth draws of k_depth (tile mode)
    uint64_t* site_base;     // [n_sites] tile mode: J^(block N H(site)) (x0), the generator state in front of the site's windows (k_sitebase)
    uint64_t* site_hash;     // [n_sites] tile mode: H(site) (GL model 1 deeper than 255 reads addresses htslib's stream by it)
    uint64_t* tail_base;     // [n_sites] tile mode with an I16 output: the second sequence's state in front of the site's windows, J^(block N H(site)) (VGL_TAIL_RAND48_X0) (k_sitebase); else null
    unsigned long long* fslot;   // [n_sites][fused_split][2] split fused sites: each workgroup's flagged per-base depth sums (zero at the start of a tile)
    // outputs (caller owned device memory; may be null)
    int32_t* site_status; int32_t* n_alleles; int32_t* n_alleles_obs; int8_t* alleles2acgt;
    int32_t* info_dp; int32_t* info_ad; int32_t* info_adf; int32_t* info_adr;
    float* qs; float* i16;
    int32_t* fmt_dp; float* gl; int32_t* pl; float* gp; uint8_t* pl_u8;
    int32_t* fmt_ad; int32_t* fmt_adf; int32_t* fmt_adr;
    uint8_t* reads_out; int32_t reads_out_cap;
    double* site_pick_err;    // [n_sites] per-site beta deviate of --error-qs 1 (dump; may be null)
    unsigned long long* dbg;  // diagnostic cycle stamps (VGL_DEBUG_STAMPS=1), else null
    // VGL_RNG_SERIAL: per-evaluation stream states found by the sequential scout (k_scout)
    uint64_t* sst_hap; uint64_t* sst_base;   // [n_sites][N]
    int32_t*  sdp;           // [n_sites][N] depth draws of the scout
    uint64_t* site_thresh;   // [n_sites] per-site base-pick error threshold (error_qs 1)
    int32_t*  scout_off;     // [N] scratch of the scout: first read index of each sample
    VglSiteTail* site_tail;  // [n_sites] with -addI16: written by the scout in serial mode, by k_tail_fin in tile mode
    const long long* hts_off; // [n_sites][N] serial mode, GL model 1 deeper than 255: first draw of an evaluation's shuffle in htslib's stream
    const uint64_t* hts_base; // [1] state of that stream at the start of the tile (serial mode)
    const long long* roff;   // [n_sites][N] index of an evaluation's first read in draw order (serial --error-qs 2, std beta)
    const double* errp_lin;  // [reads of the tile] beta deviates in draw order (vgl_betachain.hip)
};

// control block of one chunk of the beta chain (vgl_betachain.hip), device resident, read back by the host
struct VglChainCtl {
    long long remaining;     // in: reads still to be served
    long long n_pos;         // in: chain positions of this chunk whose deviate fits inside the word window
    long long n_chunk;       // out: reads served by this chunk
    long long endw;          // out: words consumed (start of the first unserved deviate)
    int32_t n_seg;           // in
    int32_t last_seg;        // out: segment that holds the last served read
    int32_t err;             // out: 1 = a deviate needed more words than the scheme allows
    int32_t pad;
};

// persistent serial-mode generator states (device memory, carried from tile to tile)
struct VglSerialState {
    uint64_t st0, st1, st2;  // rng0 (drand48), rng1, rng2
    uint32_t mt[624];        // std::mt19937 of the default beta sampler
    int32_t  mt_idx;
    int32_t  rand_f, rand_r; // glibc rand() (TYPE_3 additive feedback, degree 31, separation 3)
    uint32_t rand_state[31]; // the reference draws I16 tail distances from the never-seeded rand() (rng.h:12)
    int32_t  pad;
    uint64_t st_hts;         // htslib's private rand48 stream (hts_drand48), consumed by errmod_cal()'s shuffle at depth > 255
};


#ifdef __cplusplus
extern "C" {
#endif
// launch wrappers implemented beside their kernels in vgl_sample.hip / vgl_serial.hip / vgl_gl.hip
// (stream = hipStream_t)
int vgl_launch_sitebase(const VglDevParams* p, const VglTilePtrs* t, void* stream);
int vgl_launch_depth(const VglDevParams* p, const VglTilePtrs* t, void* stream);
int vgl_launch_sample(const VglDevParams* p, const VglTilePtrs* t, void* stream);
int vgl_launch_errp_dump(const VglDevParams* p, const double* errp, double* out, size_t n_eval, int rows, void* stream);   // read_errp = [read][site][N] from the evaluation-major planes
int vgl_launch_redo(const VglDevParams* p, const VglTilePtrs* t, void* stream);      // k_redo, when vgl_launch_sample ran the deferred build (else nothing)
int vgl_launch_site(const VglDevParams* p, const VglTilePtrs* t, void* stream);
int vgl_launch_gl(const VglDevParams* p, const VglTilePtrs* t, void* stream);
int vgl_launch_siteagg(const VglDevParams* p, const VglTilePtrs* t, void* stream);
int vgl_launch_tail(const VglDevParams* p, const VglTilePtrs* t, void* stream);      // k_tail + k_tail_fin: T.site_tail in tile mode (ahead of k_siteagg)
int vgl_launch_fused(const VglDevParams* p, const VglTilePtrs* t, void* stream);
int vgl_launch_hts_offsets(const VglDevParams* p, const VglTilePtrs* t, struct VglSerialState* st, long long* hts_off, uint64_t* hts_base, void* stream);
int vgl_launch_scout(const VglDevParams* p, const VglTilePtrs* t, struct VglSerialState* st, void* stream);
int vgl_launch_sample_serial(const VglDevParams* p, const VglTilePtrs* t, void* stream);
// vgl_betachain.hip
int vgl_chain_read_offsets(const int32_t* sdp, long long n, long long* roff, long long* total, void* stream);
long long vgl_chain_snapshots_needed(long long n_words);
int vgl_chain_seg(void);
int vgl_chain_margin_words(void);
int vgl_chain_chunk(const VglDevParams* p, struct VglSerialState* S, struct VglChainCtl* ctl, uint32_t* W, long long n_words, uint8_t* cons,
                    uint8_t* seg_exit, int32_t* seg_cnt, uint8_t* seg_entry, long long* seg_base, uint32_t* pos,
                    uint32_t* snap, long long* snap_words, void* stream);
int vgl_chain_emit(const VglDevParams* p, struct VglSerialState* S, const struct VglChainCtl* ctl, const uint32_t* W, const uint32_t* pos,
                   long long n_chunk, double* out, const uint32_t* snap, const long long* snap_words, long long n_snap, void* stream);
// vgl_host.cpp (host): PoissonSampler_init with the float32 parameters of poisson_fast; the reference's gamma_ln -- also what vgl_bounds.hip sweeps with
void vgl_pois_init(struct VglPois* o, double lambda);
double vgl_gamma_ln_host(double x);
void vgl_pois_zt_host(const struct VglPois* p, const double* gamma_ln_tab, int n, float* zt);    // VglDevParams::pois_zt from the [n] table of gamma_ln(k)
#ifdef __cplusplus
}
#endif
