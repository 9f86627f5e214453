/*
 * vcfgl_hip.h -- C ABI of the MI355X (gfx950) implementation of vcfgl's per-site
 * genotype-likelihood simulation hot path.
 *
 * The reference (isinaltinkaya/vcfgl) has no plugin / FFI surface.  Its de-facto
 * boundary for this path is
 *
 *     static int simulate_record_values(simRecord* sim)       vcfgl.cpp:327
 *     void (*calculate_gls)(simRecord* sim)                   vcfgl.cpp:222, sole call :788
 *
 * whose inputs are the parsed flags (`argStruct* args`, io.h:40-148), the decoded true
 * genotypes (`true_gts_acgt_int`, vcfgl.cpp:66, filled :132-147) and the RNG states, and
 * whose outputs are the simRecord arrays consumed by simRecord::add_tags()
 * (bcf_utils.cpp:426-507).  The entry points below are what a record loop calls IN PLACE
 * of `simulate_record_values` (vcfgl.cpp:1522,1552,1611): it batches records into a tile,
 * calls vgl_simulate_tile*, and feeds every site of the returned tile to add_tags().
 *
 * Everything is plain C: pointers, sizes, int error codes.  The library never calls
 * exit(); the reference's ERROR()/ASSERT() exits (shared.h:292-327) become VGL_E_* codes
 * plus a message retrievable with vgl_last_error().
 *
 * Tile layout (structure of arrays, sample index fastest so that one wavefront = 64
 * consecutive samples of one site reads and writes contiguous 256-byte segments):
 *
 *     per (site, sample) scalar      x[site * n_samples + sample]
 *     per (site, k, sample) plane    x[(site * K + k) * n_samples + sample]      (VGL_LAYOUT_PLANES, the default)
 *     per site vector                x[site * K + k]
 *
 * With vgl_params.out_layout = VGL_LAYOUT_SAMPLE_MAJOR (ABI 4) the multi-valued FORMAT arrays come back as the reference
 * itself keeps them (simRecord::gl_arr etc., bcf_utils.h:193-196), one slab per site: see VGL_LAYOUT_* below.
 */
#ifndef VCFGL_HIP_H
#define VCFGL_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGL_ABI_VERSION 6
/* the library is built with -fvisibility=hidden: these entry points are its whole dynamic symbol table */
#define VGL_API __attribute__((visibility("default")))

/* ---- error codes (returned by every entry point; 0 = success) ----------------------- */
#define VGL_OK              0
#define VGL_E_ARG          (-1)  /* bad parameter value (reference: io.cpp:757-1000 range checks) */
#define VGL_E_NODEVICE     (-2)  /* no HIP device / HIP runtime error                            */
#define VGL_E_NOMEM        (-3)  /* allocation failure                                           */
#define VGL_E_CAPACITY     (-4)  /* a per-sample read depth exceeded the staging capacity (mean + 8 sigma + 16 reads) where the tile could not be run
                                    again: the host-buffer entry points (vgl_simulate_tile, vgl_tile_wait) run such a tile once more on an internal
                                    context with the layout's largest capacity, 1020 reads (the reference grows its buffers, bcf_utils.cpp:618-648), so
                                    this code is left to vgl_ctx_check() (device buffers), VGL_RNG_SERIAL, per-read dumps and draws beyond 1020 reads */
#define VGL_E_UNSUPPORTED  (-5)  /* flag combination not implemented on the device path (the mt19937 beta sampler in VGL_RNG_TILE) */
#define VGL_E_QSBIN        (-6)  /* "Could not find a range for qs value" (vcfgl.cpp:63)         */
#define VGL_E_ADJQ         (-7)  /* --adjust-qs 1|2 met a read without a valid adjusted quality score: error probability
                                    exactly 0 or 1, or a negative adjusted score (the reference exits on
                                    ASSERT(adjqScore_i != -1), vcfgl.cpp:558, and ASSERT(qs >= 0 ...), gl_methods.cpp:101) */

/* ---- per-site status (reference: return value of simulate_record_values) ------------- */
#define VGL_SITE_OK            0
#define VGL_SITE_SKIP_INVAR  (-3) /* one simulated allele + --rm-invar-sites&4 (vcfgl.cpp:675-681) */
#define VGL_SITE_SKIP_EMPTY  (-4) /* INFO/DP==0 + --rm-empty-sites 1           (vcfgl.cpp:400-402) */
#define VGL_SITE_NO_READS      1  /* INFO/DP==0, site kept: simulate_site_with_no_reads (:228-315) */

/* ---- missing / special values (htslib encodings that add_tags() expects) ------------- */
#define VGL_FLOAT_MISSING_BITS 0x7F800001u  /* bcf_float_missing */
#define VGL_INT32_MISSING      ((int32_t)0x80000000) /* bcf_int32_missing = INT32_MIN */
#define VGL_GT_MISSING         0xF          /* allele nibble: missing true genotype */

/* ---- RNG addressing modes ------------------------------------------------------------ */
/* All uniform streams are the reference's own generator: glibc rand48,
 * X <- (0x5DEECE66D * X + 0xB) mod 2^48, u = X * 2^-48, X0 = (seed << 16) | 0x330E
 * (io.cpp:1054-1061, shared.h:17-22).  They differ in WHICH draw index a consumer uses. */
#define VGL_RNG_TILE    0  /* counter addressed: each (site,sample) owns a private window      */
#define VGL_RNG_SERIAL  1  /* the reference's serial consumption order: reproduces the reference
                              program bit for bit.  On the device scout kernels resolve the serial
                              stream chains (64 stream positions at a time; the std::mt19937 beta
                              stream as a parallel chain over the generator output) and record the
                              stream states per (site,sample); everything else runs in parallel.
                              Tiles must be submitted in site order (site0 = sites done so far). */

/* quality-score error sampler (rng.h:353-500) */
#define VGL_BETA_RAND48 0  /* rng.h:426-446, reference built with -D__USE_STD_BETA__=0, rng2   */
#define VGL_BETA_STD    1  /* rng.h:353-421, std::mt19937 + std::gamma_distribution (default
                              reference build; one global stream => VGL_RNG_SERIAL only)       */

/* Window layout of VGL_RNG_TILE.  Evaluation e = H(site_abs) * n_samples + sample owns draws
 * [e*block, (e+1)*block) of the rand48 sequence; stream k of that evaluation starts at
 * e*block + off[k].  A consumer that needs more draws than its sub-window simply keeps
 * stepping the generator (deterministic, only statistically overlapping).
 * H is a fixed permutation of the site indices [0, 2^W), 2^W = vgl_rng_tile_max_sites() (ABI version 4; H(0) = 0):
 *     sh = (W + 1) / 2;  x ^= x >> sh;  x = x * 0xBF58476D1CE4E5B9 mod 2^W;  x ^= x >> sh;
 *                        x = x * 0x94D049BB133111EB mod 2^W;  x ^= x >> sh            (identity for W <= 1)
 * (vgl_rng_tile_site_hash() evaluates it).  It breaks the regular spacing of the sites' windows: rand48 is a linear
 * congruential generator mod 2^48, and at offsets that are multiples of a high power of two its states are linearly
 * related (sites s, s + 2^k, s + 2^(k+1) at equal spacing would give u(s) - 2 u(s + 2^k) + u(s + 2^(k+1)) = const for
 * large k).  Results still depend only on (seed, absolute site index, sample): tiling and sharding never change a value.
 *   k=0 depth        rng1, Poisson draws                  (vcfgl.cpp:364-368, rng.h:284-351)
 *   k=1 haplotype    rng1, one draw per read              (vcfgl.cpp:473)
 *   k=2 base/strand  rng0, error test, wrong base, strand (vcfgl.cpp:486-488,582)
 *   k=3 qscore       rng2, beta deviates                  (vcfgl.cpp:428,495; rng.h:433-444)
 * Stream 3 is further divided per read: the beta deviate of read r starts at
 * e*block + off[3] + r*qs_read_stride, so the quality scores of one wavefront's reads are
 * independent work items that the device balances across lanes.  With --error-qs 1 the single
 * per-site beta deviate uses stream 3 of sample 0, read 0.
 * I16 tail distances (-addI16; vcfgl.cpp:647-663): the reference draws one per read from libc rand(), never seeded -- one serial
 * stream, which VGL_RNG_SERIAL follows.  VGL_RNG_TILE takes them from a SECOND rand48 sequence, X0 = 0x7A11D157A11D (no --seed
 * produces that state), addressed like the first: evaluation e owns draws [e*block, (e+1)*block) of it and read r uses draw r
 * as a 31-bit integer x (the state's top 31 bits, as lrand48()): tail = min(1 + x / (RAND_MAX / 50 + 1), 25)  (rng.h:12,
 * CAP_TAIL_DIST), credited -- as in the reference -- to the base of the site's last simulated read, float32 sums in (sample,
 * read) order.  (ABI 6, round 6: until then fields 13-16 of INFO/I16 were 0 in VGL_RNG_TILE.) */
typedef struct vgl_rng_layout {
    uint64_t block;
    uint64_t off[4];
    uint64_t qs_read_stride;
} vgl_rng_layout;

/* ---- layout of the per-(site, sample) FORMAT arrays with several values per sample (GL, PL, GP, AD, ADF, ADR, pl_u8) -- ABI 4
 * VGL_LAYOUT_PLANES (default): x[(site * K + k) * n_samples + sample], K = G or A of the context; entries k >= the site's own
 *   count hold the missing value.  Best for consumers on the device (one contiguous plane per genotype).
 * VGL_LAYOUT_SAMPLE_MAJOR: the slab of site i starts at x + i * K * n_samples (same allocation) and holds the record's array
 *   exactly as simRecord keeps it for bcf_update_format_*() (bcf_utils.h:193-196: gl_arr[sample * nGenotypes + g], nGenotypes
 *   variable per record): x[i * K * n_samples + sample * nK(i) + k], nK(i) = nGenotypes(site i) (GL, PL, GP, pl_u8) or
 *   n_alleles[i] (AD, ADF, ADR); what lies behind the record's array in a slab is unspecified (the kernels do not write it).  A record loop hands slab pointers to add_tags()
 *   unchanged: no per-record transposition on the host.  Skipped sites (site_status < 0) write nothing. */
#define VGL_LAYOUT_PLANES        0
#define VGL_LAYOUT_SAMPLE_MAJOR  1

/* ---- parameters = the subset of argStruct (io.h:40-148) the hot path reads ----------- */
typedef struct vgl_params {
    int32_t  abi_version;        /* VGL_ABI_VERSION */
    int32_t  seed;               /* --seed                                   io.cpp:1047-1061 */
    int32_t  n_samples;          /* bcf_hdr_nsamples                                         */
    int32_t  rng_mode;           /* VGL_RNG_*                                                */
    int32_t  beta_sampler;       /* VGL_BETA_*                                               */

    double   depth;              /* --depth (mean)   ; ignored if depths != NULL             */
    const double* depths;        /* --depths-file    ; [n_samples] per-sample means or NULL  */
    double   error_rate;         /* --error-rate                                             */
    int32_t  error_qs;           /* --error-qs 0|1|2                         io.h / README   */
    double   beta_variance;      /* --beta-variance (error_qs != 0)                          */
    int32_t  gl_model;           /* --gl-model 1|2                                           */
    double   gl1_theta;          /* --gl1-theta (default 0.83)               io.cpp:455      */
    int32_t  precise_gl;         /* --precise-gl 0|1 (usePreciseGlError)                     */
    int32_t  adjust_qs;          /* --adjust-qs bitmask                      shared.h:103-117 */
    double   adjust_by;          /* --adjust-by (default 0.499)                              */
    int32_t  n_qs_bins;          /* --qs-bins: number of [start,end,value] triples           */
    const int32_t* qs_bins;      /* [n_qs_bins][3]                           vcfgl.cpp:57-64 */
    int32_t  i16_mapq;           /* --i16-mapq (default 20)                                  */

    int32_t  do_unobserved;      /* -doUnobserved 0..5                       shared.h:70-89  */
    int32_t  rm_invar_sites;     /* --rm-invar-sites bitmask (only bit 4 acts here)          */
    int32_t  rm_empty_sites;     /* --rm-empty-sites                                         */
    int32_t  do_gvcf;            /* -doGVCF (only affects the no-reads site, vcfgl.cpp:242)  */

    int32_t  add_gl, add_gp, add_pl, add_i16, add_qs;              /* -addGL ... -addQS      */
    int32_t  add_fmt_dp, add_info_dp;                              /* -addFormatDP/-addInfoDP */
    int32_t  add_fmt_ad, add_info_ad;
    int32_t  add_fmt_adf, add_info_adf;
    int32_t  add_fmt_adr, add_info_adr;

    vgl_rng_layout layout;       /* VGL_RNG_TILE window layout; block==0 => library default  */
    int32_t  out_layout;         /* VGL_LAYOUT_* of the multi-valued FORMAT arrays (ABI 4)   */
} vgl_params;

/* ---- one tile of outputs = the simRecord arrays add_tags() reads (bcf_utils.h:157-211) -
 * Any pointer may be NULL (that output is not produced / not copied back), except
 * site_status, n_alleles and alleles2acgt which are always written.
 * G = vgl_max_genotypes(params) (10 or 15), A = vgl_max_alleles(params) (4 or 5).        */
typedef struct vgl_tile_out {
    /* per site */
    int32_t* site_status;    /* [n_sites]      VGL_SITE_*                                      */
    int32_t* n_alleles;      /* [n_sites]      sim->nAlleles (incl. <*> / <NON_REF>)           */
    int32_t* n_alleles_obs;  /* [n_sites]      sim->nAllelesObserved                           */
    int8_t*  alleles2acgt;   /* [n_sites][5]   sim->alleles2acgt; 4 = unobserved allele, -1 = none */
    int32_t* info_dp;        /* [n_sites]      INFO/DP                                         */
    int32_t* info_ad;        /* [n_sites][A]   INFO/AD  (allele order)                         */
    int32_t* info_adf;       /* [n_sites][A]                                                   */
    int32_t* info_adr;       /* [n_sites][A]                                                   */
    float*   qs;             /* [n_sites][A]   INFO/QS                                         */
    float*   i16;            /* [n_sites][16]  INFO/I16; fields 12-15 (tail distance): the reference's own
                                               libc rand() draws in VGL_RNG_SERIAL, a counter-addressed rand48
                                               sequence in VGL_RNG_TILE (see vgl_rng_layout)                */
    /* per (site, sample) */
    int32_t* fmt_dp;         /* [n_sites][n_samples]          FORMAT/DP                        */
    float*   gl;             /* [n_sites][G][n_samples]       FORMAT/GL, VCF genotype order;
                                entries g >= nGenotypes(site) hold VGL_FLOAT_MISSING_BITS       */
    int32_t* pl;             /* [n_sites][G][n_samples]       FORMAT/PL                        */
    float*   gp;             /* [n_sites][G][n_samples]       FORMAT/GP                        */
    int32_t* fmt_ad;         /* [n_sites][A][n_samples]       FORMAT/AD (allele order)         */
    int32_t* fmt_adf;        /* [n_sites][A][n_samples]                                        */
    int32_t* fmt_adr;        /* [n_sites][A][n_samples]                                        */
    /* optional per-read staging dump (reference: -printPileup, vcfgl.cpp:616-634)            */
    uint8_t* reads;          /* [read_capacity][n_sites][n_samples]  (qs << 2) | base           */
    int32_t  read_capacity;  /* rows available in `reads` / `read_errp` (0 = not requested)    */
    /* optional dumps behind -printQsError / -printGlError / -printQScores / -printBasePickError and
     * --adjust-qs 4|8|16 (vcfgl.cpp:430-435, 533-554): the deviates themselves, from which the caller derives
     * qScore and adjusted qScore exactly as vcfgl.cpp:500-523 does.  (ABI version 2.)                    */
    double*  read_errp;      /* [read_capacity][n_sites][n_samples]  error_prob_forQs_i of every read
                                (error_qs 2 only; rows >= the library's staging capacity hold NaN)         */
    double*  site_pick_err;  /* [n_sites]  base_pick_error_prob of the site (error_qs 1 only; written for
                                sites that reach the read loop, i.e. INFO/DP > 0)                          */
    /* ABI 4: FORMAT/PL in one byte per value (PL is capped at 255, shared.h:208; htslib narrows the int32 array the
     * reference hands it when it writes the record, vcfgl.cpp:907-939): a quarter of `pl`'s bytes over HBM and PCIe.
     * Same layout as `pl` (G planes, or sample-major); a missing PL (sample without reads, g >= nGenotypes(site)) is 255
     * here -- tell it from a capped value by fmt_dp == 0 / n_alleles.  Independent of `pl`: either, both or neither. */
    uint8_t* pl_u8;          /* [n_sites][G][n_samples]                                        */
} vgl_tile_out;

typedef struct vgl_ctx vgl_ctx;

/* Layout helpers (pure host arithmetic, usable without a GPU). */
VGL_API int32_t vgl_max_alleles(const vgl_params* p);      /* 4 or 5:  shared.h:148-152              */
VGL_API int32_t vgl_max_genotypes(const vgl_params* p);    /* 10 or 15: lut_nAlleles_to_nGenotypes    */
VGL_API int     vgl_default_rng_layout(const vgl_params* p, vgl_rng_layout* out);
/* VGL_RNG_TILE addresses one rand48 sequence of period 2^48: a job may use sites [0, *max_sites) before its windows
 * would repeat; *max_sites = 2^W, the largest power of two with 2^W * n_samples * block <= 2^48 (BASELINE config C4, 1e7
 * sites x 2000 samples at depth 30: 2^24 = 1.68e7).  vgl_simulate_tile* return VGL_E_ARG beyond that.  The reference's
 * serial streams have no such limit (rng.h:8-10) -- VGL_RNG_SERIAL neither. */
VGL_API int     vgl_rng_tile_max_sites(const vgl_params* p, int64_t* max_sites);
/* H(site) of the window layout above (pure host arithmetic; VGL_E_ARG outside [0, max_sites)). */
VGL_API int     vgl_rng_tile_site_hash(const vgl_params* p, int64_t site, int64_t* hashed);
VGL_API int     vgl_abi_version(void);
VGL_API const char* vgl_last_error(void);

/* Replaces args_get()'s sampler/LUT construction (io.cpp:1036-1074,1276) and main()'s
 * preCalc block (vcfgl.cpp:1661-1767): validates flags, builds the Poisson constants,
 * beta shape parameters, fixed-qscore terms, GL1 error-model tables and rand48 jump
 * tables, uploads them, and sizes the staging workspace for `max_sites_per_tile`.
 * `device` is the HIP device ordinal.  Fails with VGL_E_NODEVICE when no GPU is present. */
VGL_API int vgl_ctx_create(const vgl_params* p, int32_t device, int32_t max_sites_per_tile, vgl_ctx** out);
VGL_API int vgl_ctx_destroy(vgl_ctx* ctx);

/* Replaces the call of simulate_record_values() (vcfgl.cpp:1522,1552,1611) for `n_sites`
 * consecutive records whose absolute indices (in simulation order) start at `site0`.
 *   gt : [n_sites][n_samples] one byte per sample, (allele1 << 4) | allele0, alleles in ACGT
 *        space 0..3 exactly as check_rec_alleles() leaves them in true_gts_acgt_int
 *        (vcfgl.cpp:132-147); VGL_GT_MISSING (0xF) in either nibble = missing genotype.
 * Host variant: `gt` and every pointer in `out` are host memory; the call is synchronous. */
VGL_API int vgl_simulate_tile(vgl_ctx* ctx, int64_t site0, int32_t n_sites,
                      const uint8_t* gt, vgl_tile_out* out);

/* Asynchronous host variant (the tags of a tile are 65 B per evaluation and must stream over PCIe while the next
 * tile is computed).  vgl_simulate_tile_async() enqueues the tile -- its kernels on the context's compute stream, the copies of
 * its outputs into the caller's host buffers on a second stream behind them -- and returns a ticket; vgl_tile_wait(ticket)
 * returns once the outputs are complete (and reports the tile's device-side errors, like the synchronous call).  At most two
 * tiles may be in flight per context, submitted in site order; the buffers of an in-flight tile must not be touched.
 * Buffers obtained from vgl_host_alloc() are page-locked and are written by DMA at the link's rate; ordinary (pageable)
 * buffers work too, more slowly.  vgl_simulate_tile() is the two calls back to back. */
VGL_API int   vgl_simulate_tile_async(vgl_ctx* ctx, int64_t site0, int32_t n_sites, const uint8_t* gt, vgl_tile_out* out, int32_t* ticket);
VGL_API int   vgl_tile_wait(vgl_ctx* ctx, int32_t ticket);
VGL_API void* vgl_host_alloc(size_t bytes);            /* page-locked host memory (NULL on failure); needs a HIP device */
/* the same, placed for DMA from `device` (on a two-socket host page-locked memory lands on the NUMA node next to the device
 * that is current when it is allocated: 53 against 35 GB/s of copy-back measured): buffers of a multi-device record loop (ABI 4) */
VGL_API void* vgl_host_alloc_on(int32_t device, size_t bytes);
VGL_API void  vgl_host_free(void* p);

/* Device variant: `gt` and every pointer in `out` are device memory owned by the caller
 * (hipMalloc / a torch tensor's data_ptr); work is enqueued on `hip_stream` (a hipStream_t
 * cast to void*, NULL = default stream) and the call returns without synchronising.  A context owns
 * one staging workspace: tiles of one context must be enqueued on one stream (or otherwise ordered);
 * use one context per stream / per GPU for concurrent tiles.
 * (VGL_RNG_SERIAL with error_qs 2 and VGL_BETA_STD synchronises the stream inside the call: the number of
 * reads of the tile and the progress of the beta chain come back to the host.) */
VGL_API int vgl_simulate_tile_device(vgl_ctx* ctx, int64_t site0, int32_t n_sites,
                             const uint8_t* gt, vgl_tile_out* out, void* hip_stream);

/* Sticky device-side error flags of the last tiles (capacity overflow, qs-bin miss):
 * synchronises the stream, returns VGL_OK or the first VGL_E_* raised, and clears them. */
VGL_API int vgl_ctx_check(vgl_ctx* ctx, void* hip_stream);

/* Kernel timing hook for bench.py: brackets the device work of every following
 * vgl_simulate_tile_device call with hipEvents on its stream.  vgl_ctx_kernel_ms returns
 * accumulated milliseconds and launch counts since the last reset, one entry per bucket (ABI 5: six buckets,
 * the caller passes the length of its arrays; entries beyond VGL_N_TIMING_BUCKETS are zeroed):
 *   [VGL_T_DEPTH]   what runs ahead of the sampling kernel: k_sitebase + k_depth (the stream scouts in VGL_RNG_SERIAL)
 *   [VGL_T_SAMPLE]  k_sample
 *   [VGL_T_REDO]    k_redo (the reads the deferred build of k_sample<2> leaves to a double-precision second look)
 *   [VGL_T_SITE]    k_site
 *   [VGL_T_GL]      k_gl (the fused kernel, when it runs, is all of the tile and is counted here)
 *   [VGL_T_SITEAGG] what runs behind k_gl: k_siteagg (INFO/QS, INFO/I16) and the device copies of the per-read dumps
 * (ABI 3 had four buckets and left k_siteagg untimed; k_redo was part of k_sample's.) */
#define VGL_N_TIMING_BUCKETS 6
#define VGL_T_DEPTH   0
#define VGL_T_SAMPLE  1
#define VGL_T_REDO    2
#define VGL_T_SITE    3
#define VGL_T_GL      4
#define VGL_T_SITEAGG 5
VGL_API int vgl_ctx_timing(vgl_ctx* ctx, int32_t enable);
VGL_API int vgl_ctx_kernel_ms(vgl_ctx* ctx, double* ms, int64_t* launches, int32_t n_buckets, int32_t reset);

/* ---- what a context will launch (ABI 5) ------------------------------------------------------------------------------
 * The library picks one of several builds of its kernels from the flags (vgl_ctx_create); a record loop sizes its tiles
 * and a test asserts "the fused kernel runs here" from this record instead of inferring either from timings.
 * `size` is set by the caller to sizeof(vgl_ctx_info_t) (fields beyond it are not written; new fields are appended). */
#define VGL_DEPTH_INPLACE_MIXED   0  /* mixed per-sample means: rng.h:284-351's general sampler inside k_sample          */
#define VGL_DEPTH_KDEPTH          1  /* every mean >= 12: k_depth (rejection method, rng.h:300-312) ahead of k_sample    */
#define VGL_DEPTH_INPLACE_PRODUCT 2  /* every mean < 12: the product method's loop (rng.h:289-299) inside k_sample        */
#define VGL_DEPTH_SERIAL_SCOUT    3  /* VGL_RNG_SERIAL: the scout kernels walk the reference's stream                     */
typedef struct vgl_ctx_info_t {
    int32_t size;                /* in: sizeof(vgl_ctx_info_t) of the caller                                              */
    int32_t abi_version;
    int32_t device;
    int32_t n_samples, max_sites_per_tile;
    int32_t max_alleles, max_genotypes;
    int32_t rng_mode;            /* VGL_RNG_*                                                                             */
    int32_t depth_mode;          /* VGL_DEPTH_*                                                                           */
    int32_t fused;               /* 1: a tile that asks for no QS / I16 / per-read dump runs as ONE kernel (k_gl<.., FUSED>) */
    int32_t fused_split;         /* workgroups per site of that kernel (1, or several with the site sums exchanged in HBM)   */
    int32_t sample_lean;         /* build of k_sample a tile without per-read dump gets: 0 every option's state carried, double-precision
                                    fallbacks inline; 1 default tag surface; 2 = 1 with the fallbacks deferred to k_redo; 3 = 0 with the
                                    fallbacks deferred (optional tags: -addQS / -addI16 / strand tags / --adjust-qs; a tile with the strand
                                    tags and the quality sums and no --adjust-qs gets that build with its options fixed, "LEAN 4").
                                    ABI 6: the float32 builds (2, 3) run as two kernels, k_sample_seg<., 1> + <., 2>, when a wavefront's reads
                                    fit one pool (see pool_cap)                                                                     */
    int32_t gl_sort;             /* k_gl re-deals a workgroup's evaluations in (distinct bases, depth) order                */
    int32_t gl_wpb;              /* natural wavefronts per k_gl workgroup (4 or 8); 16 = the context is ELIGIBLE for k_gl2 (two evaluations per thread):
                                    a tile that asks for GP or FORMAT/AD* still runs k_gl with 8 -- the choice is per tile (vgl_launch_gl)              */
    int32_t read_cap;            /* staged reads per (site, sample); a deeper draw: see VGL_E_CAPACITY                       */
    int32_t pool_cap;            /* quality-score work items per wavefront and LDS segment (--error-qs 2)                   */
    int32_t pool_lds_bytes;      /* LDS bytes per wavefront of k_sample<2>                                                  */
    int32_t test_hooks;          /* 1: this library was built with -DVGL_TEST_HOOKS (environment overrides, vgl_dbg_*)      */
    int64_t workspace_bytes;     /* device memory the context owns for max_sites_per_tile (tables + staging)                */
    int64_t rng_tile_max_sites;  /* VGL_RNG_TILE: sites [0, this) are addressable (vgl_rng_tile_max_sites); 0 in serial mode */
} vgl_ctx_info_t;
VGL_API int vgl_ctx_info(const vgl_ctx* ctx, vgl_ctx_info_t* out);

/* ---- record packing on the device (ABI 6) ---------------------------------------------------------------------------------
 * The reference writes one record at a time with bcf_write (vcfgl.cpp:167-206) from the arrays simRecord::add_tags() filled
 * (bcf_utils.cpp:426-507: of every FORMAT tag only the record's nGenotypes / nAlleles values per sample).  With the sites of a job
 * sharded over the GPUs of a node, each rank hands the writer its kept sites in that form -- skipped sites (site_status < 0) dropped,
 * nG(site) = nA (nA + 1) / 2 planes of GL / PL / GP and nA(site) planes of AD / ADF / ADR -- and these two calls are the producer side
 * of that gather: an exclusive prefix sum over the tile's sites, then coalesced row copies (every byte read once, written once).
 * Every pointer but `totals` is device memory of `device`; work is enqueued on `hip_stream`.
 *
 *   vgl_pack_plan_device     offsets: int32 [3][n_sites + 1] -- row c = exclusive prefix sums of the rows a site contributes to a field of
 *                            kind c (VGL_PACK_ROW: 1 per kept site; _ROWS_G: nG(site); _ROWS_A: nA(site)), entry [n_sites] = the total.
 *                            The three totals also come back to the host (the caller sizes the packed arrays from them): synchronises the stream.
 *   vgl_pack_records_device  index_out: int32 [n_kept][3] = (site index in the tile, site_status, n_alleles) of the kept sites (may be NULL);
 *                            field f: the rows of src -- [n_sites][planes] rows of row_bytes bytes -- that belong to records, in site order,
 *                            into dst (sized from the plan: total rows of its kind x row_bytes).  Asynchronous. */
#define VGL_PACK_ROW    0   /* one row per site: per-site vectors and FORMAT tags with one value per sample (planes = 1) */
#define VGL_PACK_ROWS_G 1   /* a [site][planes][N] array of which a record keeps its nGenotypes(site) first planes        */
#define VGL_PACK_ROWS_A 2   /* ... its nAlleles(site) first planes                                                       */
typedef struct vgl_pack_field {
    const void* src;
    void*       dst;
    int32_t     kind;        /* VGL_PACK_*                                   */
    int32_t     planes;      /* rows per site in src                         */
    int64_t     row_bytes;   /* bytes of one row (N x element size, or the per-site vector) */
} vgl_pack_field;
typedef struct vgl_pack_plan { int64_t n_kept, rows_g, rows_a; } vgl_pack_plan;
VGL_API int vgl_pack_plan_device(int32_t device, int32_t n_sites, const int32_t* site_status, const int32_t* n_alleles, int32_t* offsets,
                                 vgl_pack_plan* totals, void* hip_stream);
VGL_API int vgl_pack_records_device(int32_t device, int32_t n_sites, const int32_t* site_status, const int32_t* n_alleles, const int32_t* offsets,
                                    int32_t* index_out, const vgl_pack_field* fields, int32_t n_fields, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* VCFGL_HIP_H */
