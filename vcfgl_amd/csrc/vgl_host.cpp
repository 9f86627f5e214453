// vgl_host.cpp -- host side of the C ABI declared in include/vcfgl_hip.h.
// Builds the constant tables a run needs (Poisson constants, beta shape parameters,
// fixed-qscore terms, qScore LUT, GL-model-1 error-model tables, rand48 jump tables),
// owns the device workspace, and enqueues the gfx950 kernels of vgl_sample.hip, vgl_serial.hip and vgl_gl.hip.
// There is no CPU compute path in this library.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/vcfgl_hip.h"
#include "vgl_device.h"

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? VGL_E_NOMEM : VGL_E_NODEVICE, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

// Environment overrides (tuning switches and test hooks) exist only in the -DVGL_TEST_HOOKS build of the library
// (lib/libvcfgl_hip_hooks.so, what the test-suite's hook cases and tools/ load): the shipped library reads no environment variable.
#ifdef VGL_TEST_HOOKS
static const char* hook_env(const char* name) { return getenv(name); }
#else
static const char* hook_env(const char*) { return nullptr; }
#endif
static int hook_int(const char* name, int dflt) { const char* v = hook_env(name); return v ? atoi(v) : dflt; }

extern "C" const char* vgl_last_error(void) { return g_err; }
extern "C" int vgl_abi_version(void) { return VGL_ABI_VERSION; }
extern "C" int vgl_pack_set_error(int code, const char* msg);      // (vgl_pack.hip reports through vgl_last_error() too; not exported)

// PROGRAM_WILL_ADD_UNOBSERVED (shared.h:151-152): <*> / <NON_REF> appended => 5 alleles
extern "C" int32_t vgl_max_alleles(const vgl_params* p) {
    const int d = p->do_unobserved;
    return (d == 1 || d == 2 || d == 4 || d == 5) ? 5 : 4;
}
extern "C" int32_t vgl_max_genotypes(const vgl_params* p) { return vgl_max_alleles(p) == 5 ? 15 : 10; }

static double max_depth(const vgl_params* p) {
    double dmax = p->depth;
    if (p->depths) { dmax = 0; for (int i = 0; i < p->n_samples; i++) if (p->depths[i] > dmax) dmax = p->depths[i]; }
    if (!(dmax >= 0)) dmax = 0;
    return dmax;
}

extern "C" int vgl_default_rng_layout(const vgl_params* p, vgl_rng_layout* out) {
    if (!p || !out) return fail(VGL_E_ARG, "null argument");
    const uint64_t d = (uint64_t)ceil(max_depth(p));
    const uint64_t s0 = 64;                                   // depth draws (Poisson)
    const uint64_t s1 = 4 * d + 64;                           // one haplotype draw per read
    const uint64_t s2 = 3 * s1;                               // error test + wrong base + strand
    const uint64_t qstride = 32;                              // draws reserved per beta deviate
    const uint64_t s3 = (p->error_qs == 2) ? qstride * s1 : 64;
    out->qs_read_stride = qstride;
    out->off[0] = 0; out->off[1] = s0; out->off[2] = s0 + s1; out->off[3] = s0 + s1 + s2;
    out->block = (s0 + s1 + s2 + s3) | 1;
    return VGL_OK;
}

// W of vgl_site_hash(): the largest W with 2^W * n_samples * block <= 2^48, the period of rand48
// (-1: not even one site's windows fit the period -- block * n_samples > 2^48)
static int site_hash_bits(const vgl_params* p) {
    vgl_rng_layout lay;
    if (p->layout.block) lay = p->layout; else vgl_default_rng_layout(p, &lay);
    const uint64_t raw = (uint64_t)((((unsigned __int128)1 << 48) / lay.block) / (uint64_t)p->n_samples);
    if (raw < 1) return -1;
    int W = 0;
    while (W < 40 && (2ULL << W) <= raw) ++W;
    return W;
}

// Sites [0, max) a VGL_RNG_TILE job of this shape may address: evaluation (site, sample) owns draws [e block, (e + 1) block),
// e = H(site) * n_samples + sample, and H (vgl_site_hash, vgl_device.h) permutes [0, 2^W) with 2^W * n_samples * block <= 2^48.
extern "C" int vgl_rng_tile_max_sites(const vgl_params* p, int64_t* max_sites) {
    if (!p || !max_sites || p->n_samples <= 0) return fail(VGL_E_ARG, "null argument");
    const int W = site_hash_bits(p);
    if (W < 0) return fail(VGL_E_ARG, "VGL_RNG_TILE: layout.block x n_samples exceeds the 2^48 period of rand48: not even one site is addressable");
    *max_sites = (int64_t)1 << W;
    return VGL_OK;
}

extern "C" int vgl_rng_tile_site_hash(const vgl_params* p, int64_t site, int64_t* hashed) {
    if (!p || !hashed || p->n_samples <= 0) return fail(VGL_E_ARG, "null argument");
    const int W = site_hash_bits(p);
    if (W < 0) return fail(VGL_E_ARG, "VGL_RNG_TILE: layout.block x n_samples exceeds the 2^48 period of rand48: not even one site is addressable");
    if (site < 0 || site >= ((int64_t)1 << W)) return fail(VGL_E_ARG, "site %lld outside [0, 2^%d)", (long long)site, W);
    *hashed = (int64_t)vgl_site_hash((uint64_t)site, W);
    return VGL_OK;
}

// ---- rand48 affine powers ---------------------------------------------------------------
static VglAffine aff_compose(VglAffine f, VglAffine g) {      // f after g
    VglAffine r; r.a = (f.a * g.a) & VGL_MASK48; r.c = (f.a * g.c + f.c) & VGL_MASK48; return r;
}
static VglAffine aff_pow(uint64_t n) {                        // J^n, J = one rand48 step
    VglAffine base = {VGL_LCG_A, VGL_LCG_C}, r = {1, 0};
    while (n) { if (n & 1) r = aff_compose(base, r); base = aff_compose(base, base); n >>= 1; }
    return r;
}
static VglAffine aff_pow_of(VglAffine base, uint64_t n) {
    VglAffine r = {1, 0};
    while (n) { if (n & 1) r = aff_compose(base, r); base = aff_compose(base, base); n >>= 1; }
    return r;
}

// ---- qScore -> log10 GL terms: shared.cpp:110-114 lists them with 7 significant digits
// (generator: shared.h:512-527); the same doubles are obtained by rounding the formula.
static double round7(double v) {
    if (isinf(v) || v == 0.0) return v;
    char buf[64]; snprintf(buf, sizeof buf, "%.7g", v);
    return strtod(buf, NULL);
}
static void build_q2gl(double* t /*[3][257]*/) {
    for (int q = 0; q <= 256; q++) {
        const double p = pow(10.0, -q / 10.0);
        t[q] = round7(log10(1.0 - p));
        t[257 + q] = round7(log10((1.0 - p) / 2.0 + p / 6.0));
        t[514 + q] = round7(log10(p) - log10(3.0));
    }
}

// ---- GL model 1 tables (htslib errmod.c cal_coef(), restated from the published model) --
// For one fixed qScore q the per-base sums of errmod_cal() depend only on (n, count):
//   bsum[n][c] = sum_{i<c} fk[i] * beta[q][n][i],   lhet[n][k] = lC[n][k] - n ln 2
// With per-read qScores (q < 0) the full fk[256] and beta[64][256][256] tables are returned instead.
static void build_gl1_tables(double depcorr, int q, std::vector<double>& bsum, std::vector<double>& lhet,
                             std::vector<double>* fk_out = nullptr, std::vector<double>* beta_out = nullptr) {
    const double eta = 0.03;
    double fk[256];
    fk[0] = 1.0;
    for (int n = 1; n != 256; ++n) fk[n] = pow(1. - depcorr, n) * (1.0 - eta) + eta;
    std::vector<double> lC(256 * 256, 0.0), beta(256, 0.0);
    for (int n = 1; n <= 255; ++n)
        for (int k = 1; k <= n; ++k)
            lC[n << 8 | k] = lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1);
    bsum.assign(256 * 256, 0.0);
    lhet.assign(256 * 256, 0.0);
    for (int n = 0; n < 256; ++n)
        for (int k = 0; k < 256; ++k) lhet[n << 8 | k] = lC[n << 8 | k] - M_LN2 * n;
    if (fk_out && beta_out) {
        fk_out->assign(fk, fk + 256);
        beta_out->assign((size_t)64 * 256 * 256, 0.0);
        for (int qv = 1; qv < 64; ++qv) {
            const double e = pow(10.0, -qv / 10.0), le = log(e), le1 = log(1.0 - e);
            for (int n = 1; n <= 255; ++n) {
                double* b = beta_out->data() + ((size_t)qv << 16 | (size_t)n << 8);
                double sum, sum1 = lC[n << 8 | n] + n * le;
                b[n] = HUGE_VAL;
                for (int k = n - 1; k >= 0; --k, sum1 = sum) {
                    sum = sum1 + log1p(exp(lC[n << 8 | k] + k * le + (n - k) * le1 - sum1));
                    b[k] = -10. / M_LN10 * (sum1 - sum);
                }
            }
        }
        return;
    }
    int qq = q < 4 ? 4 : q; if (qq > 63) qq = 63;              // errmod_cal clamps qual to [4,63]
    const double e = pow(10.0, -qq / 10.0), le = log(e), le1 = log(1.0 - e);
    for (int n = 1; n <= 255; ++n) {
        double sum, sum1 = lC[n << 8 | n] + n * le;
        beta[n] = HUGE_VAL;
        for (int k = n - 1; k >= 0; --k, sum1 = sum) {
            sum = sum1 + log1p(exp(lC[n << 8 | k] + k * le + (n - k) * le1 - sum1));
            beta[k] = -10. / M_LN10 * (sum1 - sum);
        }
        double acc = 0.0;
        bsum[n * 256 + 0] = 0.0;
        for (int c = 1; c <= n; ++c) { acc += fk[c - 1] * beta[c - 1]; bsum[n * 256 + c] = acc; }
    }
}

// ---- context ---------------------------------------------------------------------------
struct vgl_ctx {
    vgl_params p;                                                   // (depths / qs_bins: the copies below)
    std::vector<double> depths_copy; std::vector<int32_t> bins_copy;
    int device;
    int max_sites;
    VglDevParams dp;
    // device tables
    VglAffine* d_depth_tab = nullptr; int32_t* d_dp_pre = nullptr; uint64_t* d_site_base = nullptr; uint64_t* d_site_hash = nullptr;
    VglAffine* d_samp_tab = nullptr; VglAffine* d_qs_read_tab = nullptr; VglPois* d_pois = nullptr;
    float* d_gl2_run = nullptr; float* d_pois_zt = nullptr; unsigned long long* d_fslot = nullptr;
    double* d_q2gl = nullptr; double* d_gamma_ln = nullptr; double* d_gl1_fk = nullptr; double* d_gl1_beta = nullptr; double* d_gl1_bsum = nullptr; double* d_gl1_lhet = nullptr;
    // workspace
    uint8_t* d_reads = nullptr; double* d_errp = nullptr; uint64_t* d_ad4 = nullptr; uint64_t* d_adf4 = nullptr;
    uint32_t* d_qsum = nullptr; uint32_t* d_qsumsq = nullptr; int32_t* d_acc = nullptr; VglSiteInfo* d_sinfo = nullptr; uint64_t* d_rowmap = nullptr; uint64_t* d_rowmap8 = nullptr; uint32_t* d_gl2_redo = nullptr; uint32_t* d_gl2_list = nullptr; uint32_t* d_gl2_count = nullptr; size_t gl2_redo_words = 0;
    uint32_t* d_errflag = nullptr;
    unsigned long long* d_redo_list = nullptr; uint32_t* d_redo_count = nullptr; uint32_t redo_cap = 0; uint32_t* d_redo_bits = nullptr;   // k_sample<2, deferred> -> k_redo
    uint32_t* d_seg_list = nullptr;                                      // k_sample_seg<., 1> -> k_sample_seg<., 2>
    // beta chain of VGL_RNG_SERIAL with --error-qs 2 and the std beta sampler (vgl_betachain.hip); grow-only buffers
    long long* d_roff = nullptr; long long* d_rtotal = nullptr; double* d_errp_lin = nullptr; size_t errp_lin_cap = 0;
    uint32_t* d_cw = nullptr; uint8_t* d_ccons = nullptr; uint8_t* d_cexit = nullptr; int32_t* d_ccnt = nullptr; uint8_t* d_centry = nullptr;
    long long* d_cbase = nullptr; uint32_t* d_cpos = nullptr; uint32_t* d_csnap = nullptr; long long* d_csnapw = nullptr; VglChainCtl* d_cctl = nullptr;
    long long chain_words_cap = 0;
    unsigned long long* d_dbg = nullptr;
    // VGL_RNG_SERIAL
    long long* d_hts_off = nullptr; uint64_t* d_hts_base = nullptr;
    VglSerialState* d_serial = nullptr; uint64_t* d_sst = nullptr; uint64_t* d_site_thresh = nullptr; int32_t* d_scout_dp = nullptr; int32_t* d_sdp = nullptr; VglAffine* d_step_tab = nullptr; VglSiteTail* d_site_tail = nullptr;
    uint64_t* d_tail_base = nullptr;   // VGL_RNG_TILE with -addI16 (k_tail)
    int64_t serial_next_site = 0;   // VGL_DEBUG_STAMPS=1 diagnostic counters
    // host variant (vgl_simulate_tile / _async): two slots of device mirrors, so that the copies of one tile's tags back to the
    // host (copy stream) run beside the kernels of the next tile (compute stream) -- SURVEY H8
    struct HostSlot {
        uint8_t* d_gt = nullptr; uint8_t* h_gt = nullptr;          // h_gt: pinned staging of the packed genotypes
        void* d_out[18] = {nullptr}; size_t d_out_bytes[18] = {0};
        uint8_t* d_reads_out = nullptr; size_t d_reads_out_bytes = 0;
        double* d_errp_out = nullptr; size_t d_errp_out_bytes = 0;
        double* d_pick_out = nullptr;
        uint32_t* h_flag = nullptr;                                 // pinned: the tile's device error flags
        hipEvent_t ev_kernels = nullptr, ev_copied = nullptr;
        bool busy = false; int rc = VGL_OK;
        int64_t site0 = 0; int32_t n_sites = 0; vgl_tile_out o;     // the tile in flight (vgl_tile_wait may run it again through `deep`)
    } slot[2];
    // a draw deeper than the staging capacity (vcfgl grows its read buffers, bcf_utils.cpp:618-648): the host entry points run such a tile again on
    // this sibling context, created on first need with the staging layout's largest capacity (VGL_READ_CAP_MAX reads) and tiles of at most
    // VGL_DEEP_TILE_SITES sites.  VGL_RNG_TILE only (a value depends on (seed, site, sample) alone, so the second run is the same tile)
    vgl_ctx* deep = nullptr;
    int32_t deep_runs = 0;
    hipStream_t s_compute = nullptr, s_copy = nullptr;
    int next_slot = 0;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;     // groups of VGL_N_TIMING_BUCKETS + 1
    double ms[VGL_N_TIMING_BUCKETS] = {0}; int64_t launches[VGL_N_TIMING_BUCKETS] = {0};    // VGL_T_*
    size_t ws_bytes = 0;            // device memory owned (vgl_ctx_info)
};
#define VGL_NEV (VGL_N_TIMING_BUCKETS + 1)

static int errprob_to_qs_fixed(const vgl_params* p, double ep, int* qs, int* adjqs) {
    // vcfgl.cpp:1668-1694
    const int adj = p->adjust_qs != 0;
    int q = -1, aq = -1;
    if (0.0 == ep) { q = 63; aq = 63; }
    else if (1.0 == ep) { q = 0; aq = 0; }
    else if (0.0 < ep && ep < 1.0) {
        const double tmp = -10.0 * log10(ep);
        q = (int)tmp;
        if (adj) aq = (int)(tmp + p->adjust_by);
    } else return fail(VGL_E_ARG, "Bad error probability value: %f", ep);
    auto bins = [&](int in, int* out) -> int {
        for (int i = 0; i < p->n_qs_bins; ++i)
            if (in >= p->qs_bins[3 * i] && in <= p->qs_bins[3 * i + 1]) { *out = p->qs_bins[3 * i + 2]; return 0; }
        return fail(VGL_E_QSBIN, "Could not find a range for qs value %d", in);
    };
    if (p->n_qs_bins != 0) {
        int r = bins(q, &q); if (r) return r;
        if (adj) { r = bins(aq, &aq); if (r) return r; }
    } else {
        q = q > 63 ? 63 : q;
        if (adj) aq = aq > 63 ? 63 : aq;
    }
    if (!adj) aq = -1;
    *qs = q; *adjqs = aq;
    return VGL_OK;
}

static double gamma_ln_host(double xx) {                      // gamma_ln, rng.h:38-43,60-64
    static const double cof[6] = {76.18009172947146, -86.50532032941677, 24.01409824083091,
                                  -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5};
    double x = xx, y = xx, tmp = x + 5.5;
    tmp -= (x + 0.5) * log(tmp);
    double ser = 1.000000000190015;
    for (int j = 0; j <= 5; j++) ser += cof[j] / ++y;
    return -tmp + log(2.5066282746310005 * ser / x);
}

static void pois_init(VglPois* o, double lambda) {            // PoissonSampler_init, rng.h:259-280
    o->lm = lambda; o->sq = -1.0; o->alxm = -1.0; o->g = -1.0; o->st12 = 1; o->sqf = -1.0f; o->lmf = (float)lambda; o->e_hi = INFINITY;
    if (lambda < 12.0) o->g = exp(-lambda);
    else {
        o->st12 = 0; o->sq = sqrt(2.0 * lambda); o->alxm = log(lambda);
        // gamma_ln (rng.h:60-64)
        static const double cof[6] = {76.18009172947146, -86.50532032941677, 24.01409824083091,
                                      -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5};
        double x = lambda + 1.0, y = x, tmp = x + 5.5;
        tmp -= (x + 0.5) * log(tmp);
        double ser = 1.000000000190015;
        for (int j = 0; j <= 5; j++) ser += cof[j] / ++y;
        o->g = lambda * o->alxm - (-tmp + log(2.5066282746310005 * ser / x));
        // poisson_fast (vgl_common.hip.h): the float32 parameters, and e_hi = the smallest integer E with
        //     B(em) = 0.9 (1 + ((em + 1 - lm) / sq + 1e-6)^2) exp(em alxm - lgamma(em + 1) - g) < 2^-60   for every em >= E.
        // B(em) bounds the acceptance threshold t of every attempt whose floor is em (y < (em + 1 - lm) / sq), and B decreases from
        // em + 1 - lm = k0 >= 2 sqrt(lm) + 8 on: B(em + 1) / B(em) <= (1 + 2.2 / k) / (1 + k / lm) < 1 for k^2 > 2.2 lm -- so E is found by
        // bisection above k0.  (lgamma against the reference's six-term gamma_ln: 2e-10 relative, against a margin of 2^28.)
        o->sqf = (float)o->sq;
        const double lim = -60.0 * 0.6931471805599453;
        auto logB = [&](double em) {
            const double yb = (em + 1.0 - lambda) / o->sq + 1e-6;
            return log(0.9) + log1p(yb * yb) + em * o->alxm - lgamma(em + 1.0) - o->g;
        };
        double lo = ceil(lambda + 2.0 * sqrt(lambda) + 8.0);                // B decreases from here on
        if (logB(lo) >= lim) {
            double hi = 2.0 * lo + 64.0;
            while (logB(hi) >= lim && hi < 1e12) hi *= 2.0;
            while (hi - lo > 1.0) { const double mid = floor(0.5 * (lo + hi)); if (logB(mid) >= lim) lo = mid; else hi = mid; }
            lo = hi;
        }
        o->e_hi = (lo < 8.0e6) ? (float)lo : INFINITY;                      // (integers below 2^23 are float32 values)
    }
}

extern "C" void vgl_pois_init(VglPois* o, double lambda) { pois_init(o, lambda); }
extern "C" double vgl_gamma_ln_host(double x) { return gamma_ln_host(x); }
// VglDevParams::pois_zt: zt[k] = (float)((k alxm - gamma_ln(k + 1) - g) log2 e), k < n - 1 (the float64 operations of poisson_fast's other branch)
extern "C" void vgl_pois_zt_host(const VglPois* p, const double* gl, int n, float* zt) {
    for (int k = 0; k + 1 < n; k++) zt[k] = (float)((((double)k * p->alxm - gl[k + 1]) - p->g) * 1.4426950408889634);
    zt[n - 1] = 0.0f;
}

static void gamma1_init(VglGamma1* g, double shape) {         // Gamma1Sampler_init, rng.h:155-173
    double alpha = shape;
    g->alpha0 = shape; g->changed = 0; g->pad = 0;
    if (alpha < 1.0) { alpha += 1.0; g->changed = 1; }
    g->a1 = alpha - 1.0 / 3.0;
    g->a2 = 1.0 / sqrt(9. * g->a1);
}

static thread_local size_t* g_acct = nullptr;      // where dmalloc tallies the bytes it hands out (the context being built / grown)
struct AcctScope { explicit AcctScope(vgl_ctx* c) { g_acct = &c->ws_bytes; } ~AcctScope() { g_acct = nullptr; } };
template <typename T> static int dmalloc(T** p, size_t n) {
    if (n == 0) n = 1;
    HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
    if (g_acct) *g_acct += n * sizeof(T);
    return VGL_OK;
}

extern "C" int vgl_ctx_destroy(vgl_ctx* c) {
    if (!c) return VGL_OK;
    if (c->deep) { (void)vgl_ctx_destroy(c->deep); c->deep = nullptr; }
    (void)hipSetDevice(c->device);
    void* ptrs[] = {c->d_fslot, c->d_gl2_run, c->d_pois_zt, c->d_gl1_fk, c->d_gl1_beta, c->d_gamma_ln, c->d_samp_tab, c->d_qs_read_tab, c->d_pois, c->d_q2gl, c->d_gl1_bsum, c->d_gl1_lhet, c->d_reads, c->d_errp, c->d_ad4,
                    c->d_adf4, c->d_qsum, c->d_qsumsq, c->d_acc, c->d_sinfo, c->d_rowmap, c->d_rowmap8, c->d_gl2_redo, c->d_gl2_list, c->d_gl2_count, c->d_errflag, c->d_dbg, c->d_redo_list, c->d_redo_count, c->d_redo_bits, c->d_seg_list,
                    c->d_serial, c->d_sst, c->d_site_thresh, c->d_scout_dp, c->d_site_tail, c->d_sdp, c->d_step_tab, c->d_tail_base,
                    c->d_depth_tab, c->d_site_base, c->d_site_hash, c->d_dp_pre, c->d_hts_off, c->d_hts_base, c->d_roff, c->d_rtotal, c->d_errp_lin, c->d_cw, c->d_ccons, c->d_cexit,
                    c->d_ccnt, c->d_centry, c->d_cbase, c->d_cpos, c->d_csnap, c->d_csnapw, c->d_cctl};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    for (auto& S : c->slot) {
        if (S.busy && S.ev_copied) (void)hipEventSynchronize(S.ev_copied);
        for (void* q : S.d_out) if (q) (void)hipFree(q);
        void* dq[] = {S.d_gt, S.d_reads_out, S.d_errp_out, S.d_pick_out};
        for (void* q : dq) if (q) (void)hipFree(q);
        if (S.h_gt) (void)hipHostFree(S.h_gt);
        if (S.h_flag) (void)hipHostFree(S.h_flag);
        if (S.ev_kernels) (void)hipEventDestroy(S.ev_kernels);
        if (S.ev_copied) (void)hipEventDestroy(S.ev_copied);
    }
    if (c->s_compute) (void)hipStreamDestroy(c->s_compute);
    if (c->s_copy) (void)hipStreamDestroy(c->s_copy);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    delete c;
    return VGL_OK;
}

#define VGL_READ_CAP_MAX 1020          // the staging layout's largest capacity (four reads per word, below 1024)
#define VGL_DEEP_TILE_SITES 2048       // tiles of the sibling context that takes over a tile with a deeper draw
static int ctx_create_cap(const vgl_params* p, int32_t device, int32_t max_sites, vgl_ctx** out, int cap_override);
extern "C" int vgl_ctx_create(const vgl_params* p, int32_t device, int32_t max_sites, vgl_ctx** out) { return ctx_create_cap(p, device, max_sites, out, 0); }
static int ctx_create_cap(const vgl_params* p, int32_t device, int32_t max_sites, vgl_ctx** out, const int cap_override) {
    if (!p || !out) return fail(VGL_E_ARG, "null argument");
    *out = nullptr;
    if (p->abi_version != VGL_ABI_VERSION) return fail(VGL_E_ARG, "abi version mismatch");
    if (p->n_samples <= 0) return fail(VGL_E_ARG, "n_samples must be positive");
    if (max_sites <= 0) return fail(VGL_E_ARG, "max_sites_per_tile must be positive");
    if (p->gl_model != 1 && p->gl_model != 2) return fail(VGL_E_ARG, "[Bad argument value: '--gl-model %d'] Allowed range is [1,2]", p->gl_model);
    if (p->error_qs < 0 || p->error_qs > 2) return fail(VGL_E_ARG, "[Bad argument value: '--error-qs %d'] Allowed range is [0,2]", p->error_qs);
    if (p->do_unobserved < 0 || p->do_unobserved > 5) return fail(VGL_E_ARG, "[Bad argument value: '-doUnobserved %d'] Allowed range is [0,5]", p->do_unobserved);
    if (!(p->error_rate >= 0.0 && p->error_rate < 1.0)) return fail(VGL_E_ARG, "[Bad argument value: '--error-rate %f'] Allowed range is [0,1)", p->error_rate);
    if (p->n_qs_bins < 0 || p->n_qs_bins > VGL_MAX_QS_BINS) return fail(VGL_E_ARG, "at most %d qs bins are supported", VGL_MAX_QS_BINS);
    // a staged read is one byte, score << 2 | base, and the two-byte items / LDS sum words of k_sample<2> give a score six bits too: a binned score above
    // 63 (the reference takes --qs-bins values up to 255, io.cpp:161-163; its own default scores stop at CAP_BASEQ = 63) would be cut, so such a run is refused
    if (p->n_qs_bins > 0 && !p->qs_bins) return fail(VGL_E_ARG, "n_qs_bins > 0 without qs_bins");
    for (int i = 0; i < p->n_qs_bins; ++i)
        if (p->qs_bins[3 * i + 2] < 0 || p->qs_bins[3 * i + 2] > 63)
            return fail(VGL_E_UNSUPPORTED, "--qs-bins: bin %d maps to quality score %d; the device path stages quality scores in six bits (0 ... 63)", i, p->qs_bins[3 * i + 2]);
    if (p->gl_model == 1 && p->precise_gl) return fail(VGL_E_ARG, "Precise genotype likelihood error (--precise-gl 1) is not supported with genotype likelihood model 1 (--gl-model 1).");
    if (p->rng_mode != VGL_RNG_TILE && p->rng_mode != VGL_RNG_SERIAL) return fail(VGL_E_ARG, "rng_mode must be VGL_RNG_TILE or VGL_RNG_SERIAL");
    if (p->out_layout != VGL_LAYOUT_PLANES && p->out_layout != VGL_LAYOUT_SAMPLE_MAJOR) return fail(VGL_E_ARG, "out_layout must be VGL_LAYOUT_PLANES or VGL_LAYOUT_SAMPLE_MAJOR");
    if (p->rng_mode == VGL_RNG_TILE && p->error_qs != 0 && p->beta_sampler != VGL_BETA_RAND48)
        return fail(VGL_E_UNSUPPORTED, "the mt19937 beta sampler is one global serial stream: use VGL_RNG_SERIAL, or VGL_BETA_RAND48 with VGL_RNG_TILE");
    const double dmax = max_depth(p);
    if (p->depths) { for (int i = 0; i < p->n_samples; i++) if (!(p->depths[i] >= 0.0)) return fail(VGL_E_ARG, "depths must be >= 0"); }
    else if (!(p->depth >= 0.0)) return fail(VGL_E_ARG, "[Bad argument value: '--depth %f'] Allowed range is [0,500]", p->depth);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VGL_E_NODEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(VGL_E_NODEVICE, "device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));

    vgl_ctx* c = new vgl_ctx();
    AcctScope acct(c);
    c->p = *p; c->p.depths = nullptr; c->p.qs_bins = nullptr;
    if (p->depths) c->depths_copy.assign(p->depths, p->depths + p->n_samples);
    if (p->n_qs_bins > 0 && p->qs_bins) c->bins_copy.assign(p->qs_bins, p->qs_bins + 3 * (size_t)p->n_qs_bins);
    c->device = device; c->max_sites = max_sites;
    VglDevParams& D = c->dp;
    memset(&D, 0, sizeof D);
    const int N = p->n_samples;
    D.n_samples = N; D.chunks = (N + 63) / 64;
    D.A = vgl_max_alleles(p); D.G = vgl_max_genotypes(p);
    int cap = (int)ceil(dmax + 8.0 * sqrt(dmax) + 16.0);
    D.read_cap = (cap + 3) & ~3;
    if (hook_env("VGL_DEBUG_READ_CAP")) D.read_cap = (atoi(hook_env("VGL_DEBUG_READ_CAP")) + 3) & ~3;   // test hook: force the overflow path (a multiple of 4: staged reads are packed four per word)
    if (cap_override) D.read_cap = cap_override;                  // the sibling context of a tile with a deeper draw (vgl_tile_wait)
    if (D.read_cap > 1023) { delete c; return fail(VGL_E_ARG, "mean depth too large for the staging layout"); }
    int pool_want = 0; double pool_lmax = 0.0;
    {   // quality-score pool of one wavefront: the summed depth of its (up to) 64 samples
        double lmax = 0.0;
        for (int c0 = 0; c0 < N; c0 += 64) {
            double l = 0.0;
            for (int s = c0; s < N && s < c0 + 64; s++) l += p->depths ? p->depths[s] : p->depth;
            if (l > lmax) lmax = l;
        }
        int pc = (int)ceil(lmax + 8.0 * sqrt(lmax) + 64.0);
        pc = (pc + 63) & ~63;
        pool_want = pc; pool_lmax = lmax;              // (the two-byte-item builds below take their own limit from these)
        if (pc > 1920) pc = 1920;                      // 520 + 5 x 1920 B per wavefront: 16 wavefronts (the 4 per SIMD the kernel is
                                                       // built for) fit a CU's 160 KB LDS; larger pools run in several segments
        D.pool_cap = pc;
        D.pool_lds_bytes = (576 + 4 * (pc + 2) + pc + 7) & ~7;         // stream bases | gamma constants by stage | item slots (+ zero slot, counter) | bases
    }
    D.error_qs = p->error_qs; D.gl_model = p->gl_model; D.precise_gl = p->precise_gl; D.adjust_qs = p->adjust_qs;
    D.n_qs_bins = p->n_qs_bins; D.do_unobserved = p->do_unobserved; D.rm_invar_sites = p->rm_invar_sites;
    D.rm_empty_sites = p->rm_empty_sites;
    D.sample_strand = (p->add_i16 || p->add_fmt_adf || p->add_fmt_adr || p->add_info_adf || p->add_info_adr) ? 1 : 0;  // shared.h:160-161
    D.per_sample_depth = p->depths ? 1 : 0;
    D.need_qsum = (p->add_qs || p->add_i16) ? 1 : 0; D.need_qsumsq = p->add_i16 ? 1 : 0; D.need_adf = D.sample_strand;
    D.i16_mapq = p->i16_mapq; D.add_i16 = p->add_i16;
    D.out_layout = p->out_layout;
    D.adjust_by = p->adjust_by;
    D.serial = (p->rng_mode == VGL_RNG_SERIAL) ? 1 : 0;
    D.gl1_deep = (p->gl_model == 1 && D.read_cap > 255) ? 1 : 0;
    D.stage_fixed = (p->gl_model != 1 || D.gl1_deep || (p->add_i16 && !D.serial)) ? 1 : 0;
    D.scout_lds_bytes = ((size_t)p->n_samples * 9 <= 144 * 1024) ? (int32_t)(((size_t)p->n_samples * 9 + 15) & ~(size_t)15) : 0;
    D.beta_std = (p->beta_sampler == VGL_BETA_STD) ? 1 : 0;
    D.beta_chain = (D.serial && D.beta_std && p->error_qs == 2 && !hook_env("VGL_NO_BETA_CHAIN")) ? 1 : 0;
    {   // depth mode: k_depth pays for the rejection sampler (lambda >= 12, rng.h:300); the product method's short loop
        // stays inside k_sample, which is specialised for "all product" (2) and "mixed" (0)
        double dmin = p->depth, dmx = p->depth;
        if (p->depths) { dmin = dmx = p->depths[0]; for (int i = 1; i < N; i++) { dmin = std::min(dmin, p->depths[i]); dmx = std::max(dmx, p->depths[i]); } }
        D.depth_pre = D.serial ? 0 : (dmin >= 12.0 ? 1 : (dmx < 12.0 ? 2 : 0));
    }
    // k_gl lane order: 0 natural, 1 depth-sorted lanes storing their own evaluations (4-byte pieces), 2 depth-sorted lanes and
    // natural-order stores through LDS -- the last is at least as fast as the others from depth 5 (config C5) to depth 30
    D.gl_sort = hook_int("VGL_GL_SORT", dmax >= 1.0 ? 2 : 0);
    D.gl_flip2 = hook_int("VGL_GL_FLIP2", 1);
    D.gl_wpb = hook_int("VGL_GL_WPB", D.gl_sort ? 8 : 4);
    D.slow_period = hook_int("VGL_SLOW_PERIOD", 4);
    if (D.slow_period < 1) D.slow_period = 1;
    D.slow_period_n = hook_int("VGL_SLOW_PERIOD_N", 4);
    if (D.slow_period_n < 1) D.slow_period_n = 1;
    D.xcd_map = hook_int("VGL_XCD_MAP", 1);
    D.dbg_phase = hook_int("VGL_DEBUG_PHASE", 0);
    D.dbg_stamps = hook_int("VGL_DEBUG_STAMPS", 0);
    D.dbg_fuse_alone = hook_int("VGL_DEBUG_FUSE_ALONE", 0);
    D.dbg_depth_chunk = hook_int("VGL_DEPTH_CHUNK", 0);
    D.dbg_qs_exact = hook_int("VGL_DEBUG_QS_EXACT", 0);
    for (int i = 0; i < p->n_qs_bins * 3; i++) D.qs_bins[i] = p->qs_bins[i];
    D.err_thresh = (uint64_t)ceil(ldexp(p->error_rate, 48));

    int rc = VGL_OK;
    std::vector<double> q2gl(3 * 257);
    build_q2gl(q2gl.data());
    D.pre_q = D.pre_adjq = -1;
    if (p->error_qs == 0 || p->error_qs == 1) {                  // preCalc, vcfgl.cpp:1661-1743
        if ((rc = errprob_to_qs_fixed(p, p->error_rate, &D.pre_q, &D.pre_adjq))) { vgl_ctx_destroy(c); return rc; }
        if ((p->adjust_qs & 3) && D.pre_adjq < 0) { vgl_ctx_destroy(c); return fail(VGL_E_ADJQ, "--adjust-qs %d --adjust-by %g: the adjusted quality score is negative", p->adjust_qs, p->adjust_by); }
        if (p->gl_model == 2) {
            if (!p->precise_gl) {
                const int q = (p->adjust_qs & 1) ? D.pre_adjq : D.pre_q;
                D.pre_homT = q2gl[q]; D.pre_het = q2gl[257 + q]; D.pre_homF = q2gl[514 + q];
            } else {
                const double e = p->error_rate;
                if (0.0 == e) { D.pre_homT = 0; D.pre_het = -0.3010299956639812; D.pre_homF = -INFINITY; }
                else { D.pre_homT = log10(1.0 - e); D.pre_het = log10((1.0 - e) / 2.0 + e / 6.0); D.pre_homF = log10(e) - 0.47712125471966244; }
            }
        }
    }
    if (p->error_qs != 0) {                                       // rng.h:455-477
        const double mean = p->error_rate, var = p->beta_variance;
        if (!(mean > 0.0 && mean < 1.0 && var > 0.0)) { vgl_ctx_destroy(c); return fail(VGL_E_ARG, "--error-qs 1 or 2 requires 0 < --error-rate < 1 and --beta-variance > 0"); }
        const double oom = 1.0 / mean;
        const double a = (((1.0 - mean) / var) - oom) * pow(mean, 2), b = a * (oom - 1);
        if (a <= 0.0 || b <= 0.0) { vgl_ctx_destroy(c); return fail(VGL_E_ARG, "Beta shape parameters must be positive (alpha=%f beta=%f); use different --error-rate / --beta-variance", a, b); }
        gamma1_init(&D.gx, a); gamma1_init(&D.gy, b);
        // k_sample<2>'s sure-accept bound: far above the rounding of the reference's own right-hand side
        // 0.5 x^2 + a1 (1 - v + log v), which is about 4e-16 a1 + 1e-16 x^2
        D.sure_margin = 1e-9 + 1e-14 * std::max(D.gx.a1, D.gy.a1);
        D.beta_a = a; D.beta_b = b;
    }
    // The deferred builds of k_sample<2> (5 wavefronts per SIMD, no double-precision fallback code in the kernel: the reads a float32
    // bound cannot settle go to k_redo) serve every tag surface -- LEAN 2 the default one, LEAN 3 (round 4) -addQS / -addI16, strand tags
    // and --adjust-qs, with or without --precise-gl 1 (k_redo then also rewrites the read's staged error probability).  The build with
    // the fallbacks inline (LEAN 0 / 1) remains for a per-read dump and for a beta shape parameter below 8 (the gamma sampler's bounded
    // test then leaves its series' range |a2 x| <= 1/3 too often).
    D.dbg_redo_every = hook_int("VGL_DEBUG_REDO_EVERY", 0);
    // the tag surface needs none of the owners' optional per-read state (quality sums, strand draws, --adjust-qs): the LEAN builds of k_sample
    bool bins_below_255 = true;
    for (int i = 0; i < p->n_qs_bins; ++i) if (p->qs_bins[3 * i] > 254 || p->qs_bins[3 * i + 1] > 254) bins_below_255 = false;
    D.lean_ok = (!D.need_qsum && !D.sample_strand && !D.need_adf && p->adjust_qs == 0 && !hook_env("VGL_NO_LEAN")) ? 1 : 0;
    D.defer_ok = (!D.serial && p->error_qs == 2 &&
                  !D.gx.changed && !D.gy.changed && D.gx.alpha0 >= 8.0 && D.gy.alpha0 >= 8.0 && !hook_env("VGL_NO_DEFER") && !hook_env("VGL_DEBUG_QS_EXACT") && !hook_env("VGL_NO_LEAN") &&
#ifdef VGL_PREC_F64
                  !(!p->precise_gl && (D.read_cap > 256 || !bins_below_255))) ? 1 : 0;   // (the two-byte items of the float32 builds hold a read index of 8 bits
#else
                  !(D.read_cap > 256 || !bins_below_255)) ? 1 : 0;                       // (the two-byte items of the float32 builds hold a read index of 8 bits
#endif
                                                                             // and look binned scores up in a 256-entry table: other runs take the inline build)
    // one workgroup per site does everything (k_gl<.., FUSED>, vgl_gl.hip): sampling with one fixed score, the site's allele order and the
    // likelihoods, with nothing staged in HBM between them
    // (round 4: sites of more than 512 samples split over up to four consecutive workgroups, up to 128 staged reads.  The kernel also takes its
    // depths from k_depth where the rejection method draws them, but at depth 20 the three kernels measure faster, so that stays behind the
    // hooks build's VGL_FUSE_DEEP; VGL_FUSE_MAX_SPLIT: tuning hook)
    D.fused_split = N <= 512 ? 1 : (N + 511) / 512;
    D.fused = (!D.serial && p->error_qs == 0 && p->gl_model == 2 && !p->precise_gl && (D.depth_pre == 2 || D.depth_pre == 1) && !D.need_qsum && !D.sample_strand &&
               (D.depth_pre == 2 || hook_int("VGL_FUSE_DEEP", 0)) &&      // measured (tools/fuse_ab.sh): at depth 20 the three kernels are faster (1.50e10 against 1.40e10 at N = 500, 1.55e10 against 1.36e10 at N = 1000)
               !D.need_adf && p->adjust_qs == 0 && N > 128 && D.fused_split <= hook_int("VGL_FUSE_MAX_SPLIT", 4) && D.read_cap <= 128 &&
               !hook_env("VGL_NO_FUSE") && !hook_env("VGL_NO_LEAN")) ? 1 : 0;
    if (!D.fused) D.fused_split = 0;
    D.qsum_lds = (D.defer_ok && !D.lean_ok && ((p->adjust_qs & 3) == 0 || (p->adjust_qs & 3) == 3) && D.read_cap <= 130) ? 1 : 0;    // 130 x 63 = 8190 < 2^13, 130 x 63^2 = 515970 < 2^19
    if (D.defer_ok) {
        // pools of the deferred builds.  Without --precise-gl 1 (float32 loop, round 5) an item is TWO bytes: 576 B + 2 x (items + 2) (+ 1 KB of
        // quality-sum words with qsum_lds, + 256 B of binned scores with --qs-bins) -- 2240 items (depth 30 in one segment) leave LDS for the eight
        // wavefronts per SIMD k_sample<2, LEAN 2> is built for (32 x 5.1 KB in a CU's 160 KB).  With --precise-gl 1 (float64 loop): five bytes, 1472
        // items = 5 wavefronts per SIMD (1416 with the 512 B of sum words)
#if !defined(VGL_POOL_F64) && !defined(VGL_PREC_F64)
        const bool p16 = true;                                   // (round 6: --precise-gl 1 runs the float32 loop too, + 32 bytes of double constants)
#elif !defined(VGL_POOL_F64)
        const bool p16 = !p->precise_gl;
#else
        const bool p16 = false;
#endif
        const int extra16 = (D.qsum_lds ? 1024 : 0) + (p->n_qs_bins ? 256 : 0) + (p->precise_gl ? 32 : 0);
        const int cap_defer = p16 ? ((5120 - 576 - 8 - (D.lean_ok ? (p->n_qs_bins ? 256 : 0) : 0)) / 2 / 64 * 64) : (D.qsum_lds ? 1416 : 1472);
        // the float32 build of the default tag surface as two kernels (k_sample_seg, vgl_sample.hip) when a wavefront's reads fit one pool up to 8 sigma
        // (a pool that holds the summed depth + 4 sigma: 3e-5 of the wavefronts go through the list -- depth 30 with --qs-bins: 2112 items for 1920 + 4 x 43.8)
        D.seg_split = (p16 && (double)cap_defer >= pool_lmax + 4.0 * sqrt(pool_lmax) && !hook_env("VGL_NO_SEG_SPLIT")) ? 1 : 0;
        // round 6: the two-byte-item builds take min(summed depth + 8 sigma, what eight wavefronts per SIMD leave) -- the five-byte limit of 1920 above
        // was still applied first, so that depth 30 (mean 1920 reads per wavefront) ran half of its wavefronts in two segments
        if (p16) D.pool_cap = pool_want;
        if (D.pool_cap > cap_defer) D.pool_cap = cap_defer;
        if (hook_env("VGL_DEBUG_POOL_CAP")) { D.pool_cap = std::max(64, std::min(D.pool_cap, atoi(hook_env("VGL_DEBUG_POOL_CAP")) / 64 * 64)); if (hook_int("VGL_SEG_SPLIT", 0)) D.seg_split = p16 ? 1 : 0; }   // test hooks: small pools, the split forced on
        D.seg_limit = hook_int("VGL_DEBUG_SEG_LIMIT", D.pool_cap);
        if (D.seg_limit > D.pool_cap) D.seg_limit = D.pool_cap;
        D.pool_lds_bytes = p16 ? (((576 + 2 * (D.pool_cap + 2) + 7) & ~7) + extra16)
                               : (((576 + 4 * (D.pool_cap + 2) + D.pool_cap + 7) & ~7) + (D.qsum_lds ? 512 : 0));   // (vgl_launch_sample sizes the LDS of the build it launches)
    }
    pois_init(&D.pois0, p->depths ? 0.0 : p->depth);

    // rand48 addressing
    vgl_rng_layout lay;
    if (p->layout.block) lay = p->layout; else vgl_default_rng_layout(p, &lay);
    c->p.layout = lay;
    D.x0 = ((((uint64_t)(uint32_t)p->seed) << 16) | 0x330EULL) & VGL_MASK48;   // io.cpp:1054-1061
    for (int k = 0; k < 4; k++) D.off[k] = aff_pow(lay.off[k]);
    const VglAffine jb = aff_pow(lay.block);                       // one evaluation block
    VglAffine js = aff_pow_of(jb, (uint64_t)N);                    // one site = N blocks
    for (int b = 0; b < 40; b++) { D.site_pow[b] = js; js = aff_compose(js, js); }
    std::vector<VglAffine> samp(N);
    { VglAffine cur = {1, 0}; for (int s = 0; s < N; s++) { samp[s] = cur; cur = aff_compose(jb, cur); } }

#define TRY(x) do { if ((rc = (x))) { vgl_ctx_destroy(c); return rc; } } while (0)
#define TRYHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { vgl_ctx_destroy(c); return fail(e_ == hipErrorOutOfMemory ? VGL_E_NOMEM : VGL_E_NODEVICE, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
    if (!D.serial) {                                               // k_sitebase's outputs; k_depth: J^(off0 + block*s)
        D.site_hash_bits = site_hash_bits(&c->p);
        if (D.site_hash_bits < 0) { vgl_ctx_destroy(c); return fail(VGL_E_ARG, "VGL_RNG_TILE: layout.block x n_samples exceeds the 2^48 period of rand48: not even one site is addressable"); }
        D.depth_magic = (uint32_t)((1ULL << 32) / (uint64_t)N + 1ULL);
        std::vector<VglAffine> dt(N);
        for (int s = 0; s < N; s++) dt[s] = aff_compose(D.off[0], samp[s]);
        TRY(dmalloc(&c->d_depth_tab, (size_t)N));
        TRYHIP(hipMemcpy(c->d_depth_tab, dt.data(), sizeof(VglAffine) * N, hipMemcpyHostToDevice));
        D.depth_tab = c->d_depth_tab;
        TRY(dmalloc(&c->d_site_base, (size_t)max_sites)); TRY(dmalloc(&c->d_site_hash, (size_t)max_sites));
        if (p->add_i16) {    // INFO/I16 fields 13-16 (k_tail, vgl_gl.hip): the same windows of a second rand48 sequence; the staging capacity fits a window
            if ((uint64_t)D.read_cap > lay.block) { vgl_ctx_destroy(c); return fail(VGL_E_ARG, "-addI16: layout.block (%llu) is smaller than the staging capacity of %d reads", (unsigned long long)lay.block, D.read_cap); }
            TRY(dmalloc(&c->d_tail_base, (size_t)max_sites)); TRY(dmalloc(&c->d_site_tail, (size_t)max_sites));
        }
    }
    TRY(dmalloc(&c->d_samp_tab, (size_t)N));
    TRYHIP(hipMemcpy(c->d_samp_tab, samp.data(), sizeof(VglAffine) * N, hipMemcpyHostToDevice));
    D.samp_tab = c->d_samp_tab;
    if (p->error_qs == 2) {
        if (lay.qs_read_stride == 0) { vgl_ctx_destroy(c); return fail(VGL_E_ARG, "layout.qs_read_stride must be > 0 with --error-qs 2"); }
        std::vector<VglAffine> rt(D.read_cap);
        const VglAffine jr = aff_pow(lay.qs_read_stride);
        VglAffine cur = {1, 0};
        for (int r = 0; r < D.read_cap; r++) { rt[r] = cur; rt[r].c <<= 4; cur = aff_compose(jr, cur); }   // constants scaled by 16: the pool loop's states are (aff52)
        // at least 256 entries (zeros behind the reads'): the two-byte-item pool loop prefetches the table entry of a lane's NEXT item from the item's 8-bit
        // read index without asking whether there is a next item -- a lane past the end of its items reads an arbitrary slot and never uses what comes back
        const size_t rt_n = std::max<size_t>(256, (size_t)D.read_cap);
        TRY(dmalloc(&c->d_qs_read_tab, rt_n));
        TRYHIP(hipMemset(c->d_qs_read_tab, 0, sizeof(VglAffine) * rt_n));
        TRYHIP(hipMemcpy(c->d_qs_read_tab, rt.data(), sizeof(VglAffine) * D.read_cap, hipMemcpyHostToDevice));
        D.qs_read_tab = c->d_qs_read_tab;
    }
    if (p->depths) {
        std::vector<VglPois> pv(N);
        for (int s = 0; s < N; s++) pois_init(&pv[s], p->depths[s]);
        TRY(dmalloc(&c->d_pois, (size_t)N));
        TRYHIP(hipMemcpy(c->d_pois, pv.data(), sizeof(VglPois) * N, hipMemcpyHostToDevice));
        D.pois = c->d_pois;
    }
    {
        const int n = 2048;
        std::vector<double> gl(n, 0.0);
        for (int k = 1; k < n; k++) gl[k] = gamma_ln_host((double)k);
        TRY(dmalloc(&c->d_gamma_ln, (size_t)n));
        TRYHIP(hipMemcpy(c->d_gamma_ln, gl.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        D.gamma_ln_tab = c->d_gamma_ln; D.gamma_ln_n = n;
        if (!p->depths && !D.pois0.st12 && !hook_env("VGL_NO_POIS_ZT")) {
            std::vector<float> zt(n);
            vgl_pois_zt_host(&D.pois0, gl.data(), n, zt.data());
            TRY(dmalloc(&c->d_pois_zt, (size_t)n));
            TRYHIP(hipMemcpy(c->d_pois_zt, zt.data(), sizeof(float) * n, hipMemcpyHostToDevice));
            D.pois_zt = c->d_pois_zt;
        }
    }
    TRY(dmalloc(&c->d_q2gl, (size_t)3 * 257));
    TRYHIP(hipMemcpy(c->d_q2gl, q2gl.data(), sizeof(double) * 3 * 257, hipMemcpyHostToDevice));
    D.q2gl = c->d_q2gl;
    if (p->gl_model == 2 && p->error_qs != 2 && !hook_env("VGL_NO_GL2_RUN")) {
        // GL model 2 with one fixed score: an evaluation whose n reads all show one base ends in accumulators that depend on n alone.
        // The reference's loop (gl_methods.cpp:22-59: per read one double add rounded to float per genotype, float maximum over the
        // genotypes that exist, float subtraction) is run here once per n and variant; k_gl looks the three values up instead of
        // running the loop for every such evaluation (most of them: all reads of a homozygous sample without a base-call error).
        // Same operations in the same order and precision as k_gl's read loop (-ffp-contract=off; float / double are IEEE on this host).
        const int rows = D.read_cap + 1;
        std::vector<float> run((size_t)2 * rows * 3);
        const double term[3] = {D.pre_homT, D.pre_het, D.pre_homF};
        for (int variant = 0; variant < 2; ++variant) {
            volatile float tr[3] = {-0.0f, -0.0f, -0.0f};                   // bcf_utils.h:310 (volatile: every step rounds to float32 in memory)
            for (int i = 0; i < 3; ++i) run[((size_t)variant * rows) * 3 + i] = tr[i];
            for (int n = 1; n < rows; ++n) {
                float mx = -INFINITY;
                for (int i = 0; i < 3; ++i) {
                    const float v = (float)((double)tr[i] + term[i]);
                    tr[i] = v;
                    if (variant == 0 || i == 0) mx = (v > mx) ? v : mx;
                }
                for (int i = 0; i < 3; ++i) { const float d = tr[i] - mx; tr[i] = d; }
                for (int i = 0; i < 3; ++i) run[((size_t)variant * rows + n) * 3 + i] = tr[i];
            }
        }
        TRY(dmalloc(&c->d_gl2_run, run.size()));
        TRYHIP(hipMemcpy(c->d_gl2_run, run.data(), sizeof(float) * run.size(), hipMemcpyHostToDevice));
        D.gl2_run = c->d_gl2_run;
    }
    if (p->gl_model == 1) {
        std::vector<double> bsum, lhet, fkv, betav;
        if (p->error_qs == 2) {                                    // gl_methods.cpp:233-302: per-read qScores
            build_gl1_tables(1.0 - p->gl1_theta, -1, bsum, lhet, &fkv, &betav);
            // errmod_cal()'s per-read term fk[w] * beta[q << 16 | n << 8 | c] (one double product, the same bits wherever it is
            // formed), q in [4, 63], n <= min(255, staging capacity), c < n: compact in n and c
            const int nc = std::min(255, D.read_cap) + 1;
            std::vector<double> fb((size_t)60 * nc * nc, 0.0);
            for (int q = 4; q < 64; ++q)
                for (int n = 1; n < nc; ++n)
                    for (int i = 0; i < n; ++i)
                        fb[((size_t)(q - 4) * nc + n) * nc + i] = fkv[i] * betav[(size_t)q << 16 | (size_t)n << 8 | (size_t)i];
            TRY(dmalloc(&c->d_gl1_beta, fb.size()));
            TRYHIP(hipMemcpy(c->d_gl1_beta, fb.data(), sizeof(double) * fb.size(), hipMemcpyHostToDevice));
            D.gl1_fkbeta = c->d_gl1_beta; D.gl1_nc = nc;
        } else
        build_gl1_tables(1.0 - p->gl1_theta, (p->adjust_qs & 1) ? D.pre_adjq : D.pre_q, bsum, lhet);   // io.cpp:1276, gl_methods.cpp:318
        TRY(dmalloc(&c->d_gl1_bsum, bsum.size())); TRY(dmalloc(&c->d_gl1_lhet, lhet.size()));
        TRYHIP(hipMemcpy(c->d_gl1_bsum, bsum.data(), sizeof(double) * bsum.size(), hipMemcpyHostToDevice));
        TRYHIP(hipMemcpy(c->d_gl1_lhet, lhet.data(), sizeof(double) * lhet.size(), hipMemcpyHostToDevice));
        D.gl1_bsum = c->d_gl1_bsum; D.gl1_lhet = c->d_gl1_lhet;
    }
    const size_t E = (size_t)max_sites * N;
    TRY(dmalloc(&c->d_reads, E * D.read_cap));
    if (!D.serial) TRY(dmalloc(&c->d_dp_pre, E));
    if ((p->precise_gl || (D.serial && !D.beta_chain)) && p->error_qs == 2) TRY(dmalloc(&c->d_errp, E * D.read_cap));
    if (D.beta_chain) { TRY(dmalloc(&c->d_roff, E)); TRY(dmalloc(&c->d_rtotal, (size_t)1)); TRY(dmalloc(&c->d_cctl, (size_t)1)); }
    if (D.serial) {
        TRY(dmalloc(&c->d_sst, E * 2)); TRY(dmalloc(&c->d_site_thresh, (size_t)max_sites)); TRY(dmalloc(&c->d_scout_dp, (size_t)N)); TRY(dmalloc(&c->d_sdp, E));
        {
            std::vector<VglAffine> stp(192);
            VglAffine cur = {1, 0}; const VglAffine j1 = aff_pow(1);
            for (int k = 0; k < 192; k++) { stp[k] = cur; cur = aff_compose(j1, cur); }
            TRY(dmalloc(&c->d_step_tab, (size_t)192));
            TRYHIP(hipMemcpy(c->d_step_tab, stp.data(), sizeof(VglAffine) * 192, hipMemcpyHostToDevice));
            D.step_tab = c->d_step_tab;
        }
        TRY(dmalloc(&c->d_serial, (size_t)1));
        VglSerialState hs; memset(&hs, 0, sizeof hs);
        hs.st0 = hs.st1 = hs.st2 = D.x0;                         // io.cpp:1054-1061: all three streams start equal
        hs.mt[0] = (uint32_t)p->seed;                            // io.cpp:1039, rng.h:400
        for (int i = 1; i < 624; i++) hs.mt[i] = 1812433253u * (hs.mt[i - 1] ^ (hs.mt[i - 1] >> 30)) + (uint32_t)i;
        hs.mt_idx = 624;
        hs.st_hts = VGL_HTS_RAND48_X0;                           // htslib never seeds hts_drand48
        {   // glibc srandom_r(1) + the 310 discarded outputs: the state a process that never calls srand() starts from
            int32_t word = 1; hs.rand_state[0] = 1;
            for (int i = 1; i < 31; i++) { const long hi = word / 127773, lo = word % 127773; long w = 16807 * lo - 2836 * hi; if (w < 0) w += 2147483647; word = (int32_t)w; hs.rand_state[i] = (uint32_t)word; }
            hs.rand_f = 3; hs.rand_r = 0;
            for (int k = 0; k < 310; k++) {
                hs.rand_state[hs.rand_f] += hs.rand_state[hs.rand_r];
                if (++hs.rand_f >= 31) { hs.rand_f = 0; ++hs.rand_r; } else if (++hs.rand_r >= 31) hs.rand_r = 0;
            }
        }
        if (p->add_i16) TRY(dmalloc(&c->d_site_tail, (size_t)max_sites));
        if (D.gl1_deep) { TRY(dmalloc(&c->d_hts_off, E)); TRY(dmalloc(&c->d_hts_base, (size_t)1)); }
        TRYHIP(hipMemcpy(c->d_serial, &hs, sizeof hs, hipMemcpyHostToDevice));
    }
    TRY(dmalloc(&c->d_ad4, E));
    if (D.need_adf) TRY(dmalloc(&c->d_adf4, E));
    if (D.need_qsum) TRY(dmalloc(&c->d_qsum, E * 4));
    if (D.need_qsumsq) TRY(dmalloc(&c->d_qsumsq, E * 4));
    TRY(dmalloc(&c->d_acc, (size_t)max_sites * VGL_ACC_STRIDE));
    TRY(dmalloc(&c->d_sinfo, (size_t)max_sites));
    if (p->gl_model == 2) TRY(dmalloc(&c->d_rowmap, (size_t)max_sites * 16));
    {
        // GL model 2, three-kernel path: k_gl2 (two evaluations per thread: vgl_gl.hip) where it measured faster than k_gl (tools/gl2x_sweep.py,
        // k_gl's time per tile with k_gl2 / with k_gl): one fixed score 0.77 - 0.85 at depths 12 ... 60, per-read scores 0.99 at depth 16, 0.95 at
        // 20, 0.91 at 30, 0.87 at 40.  Its pool holds the upper accumulator rows of 256 three- / four-base evaluations of a workgroup's 1024:
        // beyond ~0.8 expected base-call errors per evaluation workgroups start to overflow into k_gl_redo, and k_gl is the better choice.
        // Planes layout, sort on, no --precise-gl 1
        double dsum = 0.0;
        for (int i = 0; i < N; i++) dsum += p->depths ? p->depths[i] : p->depth;
        const bool can = p->gl_model == 2 && !p->precise_gl && D.gl_sort != 0 && D.gl_wpb == 8 && p->out_layout == VGL_LAYOUT_PLANES && !D.fused;
        const double dmean = dsum / (double)N, errs = dmean * p->error_rate;      // expected base-call errors per evaluation: what makes three- and four-base evaluations
        // (at the bench's full tile size per-read scores at depth 20 measured equal with the first version, 2.40-2.43 ms either way, depth 30 -6.5 %; and with
        //  GP or the AD-type FORMAT tags k_gl2's two epilogues per thread cost more than they hide -- all tags: 4.9 -> 5.7 ms: vgl_launch_gl looks at the tile)
        // (... and without the GP / AD epilogue in the shipped k_gl2, depth 20 measures 2.355-2.388 against 2.397-2.413 ms: from depth 18)
        const bool want = dmean >= (p->error_qs != 2 ? 12.0 : 18.0) && errs <= 0.8;
        D.gl2x = can ? hook_int("VGL_GL2X", want ? 1 : 0) : 0;                    // (VGL_GL2X=2: also for tiles with GP / FORMAT/AD*)
        D.dbg_gl2_ovc = hook_int("VGL_DEBUG_GL2_OVC", 0);
    }
    if (D.gl2x) {
        const size_t wg2 = (size_t)max_sites * D.chunks / 16 + 1;
        TRY(dmalloc(&c->d_rowmap8, (size_t)max_sites * 32));
        c->gl2_redo_words = (wg2 + 31) / 32;
        TRY(dmalloc(&c->d_gl2_redo, c->gl2_redo_words));
        TRYHIP(hipMemset(c->d_gl2_redo, 0, c->gl2_redo_words * sizeof(uint32_t)));
        TRY(dmalloc(&c->d_gl2_list, c->gl2_redo_words * 32));
        TRY(dmalloc(&c->d_gl2_count, (size_t)1));
    }
    TRY(dmalloc(&c->d_errflag, (size_t)1));
    if (D.fused && D.fused_split > 1) TRY(dmalloc(&c->d_fslot, (size_t)max_sites * D.fused_split * 2));
    if (D.defer_ok) {
        // about 6 reads in 10^4 take this path at C3 / C4 (tools/redo_rate.py); the list has room for 1 in 64 of the staging capacity
        // (VGL_DEBUG_REDO_CAP: test hook), what does not fit is marked in a bitmap over the staged reads (all zero between tiles)
        const size_t reads = E * (size_t)D.read_cap;
        c->redo_cap = (uint32_t)std::min<size_t>(0xFFFFFFF0u, hook_env("VGL_DEBUG_REDO_CAP") ? (size_t)atol(hook_env("VGL_DEBUG_REDO_CAP")) : std::max<size_t>(65536, reads / 64));
        TRY(dmalloc(&c->d_redo_bits, (reads + 31) / 32));
        TRYHIP(hipMemset(c->d_redo_bits, 0, sizeof(uint32_t) * ((reads + 31) / 32)));
        c->redo_cap /= VGL_REDO_PARTS;                                     // entries per partition (0 with a tiny VGL_DEBUG_REDO_CAP: every entry goes to the bitmap)
        TRY(dmalloc(&c->d_redo_list, std::max<size_t>(1, (size_t)c->redo_cap * VGL_REDO_PARTS)));
        TRY(dmalloc(&c->d_redo_count, (size_t)VGL_REDO_PARTS * VGL_REDO_STRIDE));
        if (D.seg_split) TRY(dmalloc(&c->d_seg_list, (size_t)max_sites * D.chunks));
        TRYHIP(hipMemset(c->d_redo_count, 0, sizeof(uint32_t) * VGL_REDO_PARTS * VGL_REDO_STRIDE));
    }
    TRYHIP(hipMemset(c->d_errflag, 0, sizeof(uint32_t)));
    if (hook_env("VGL_DEBUG_STAMPS") || hook_env("VGL_DEBUG_PHASE")) { TRY(dmalloc(&c->d_dbg, (size_t)16)); TRYHIP(hipMemset(c->d_dbg, 0, 128)); }
    TRYHIP(hipDeviceSynchronize());          // tables and cleared words are in place before any (non-blocking) stream uses them
    *out = c;
    return VGL_OK;
}

static int resolve_timing(vgl_ctx* c) {
    for (size_t i = 0; i + VGL_NEV - 1 < c->ev.size(); i += VGL_NEV) {
        HIPCHK(hipEventSynchronize(c->ev[i + VGL_NEV - 1]));
        for (int k = 0; k < VGL_N_TIMING_BUCKETS; k++) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, c->ev[i + k], c->ev[i + k + 1]));
            c->ms[k] += ms; c->launches[k] += 1;
        }
    }
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    c->ev.clear();
    return VGL_OK;
}

extern "C" int vgl_ctx_timing(vgl_ctx* c, int32_t enable) {
    if (!c) return fail(VGL_E_ARG, "null ctx");
    c->timing = enable != 0;
    return VGL_OK;
}

extern "C" int vgl_ctx_kernel_ms(vgl_ctx* c, double* ms, int64_t* launches, int32_t n_buckets, int32_t reset) {
    if (!c || !ms || !launches || n_buckets < 0) return fail(VGL_E_ARG, "null ctx / arrays");
    HIPCHK(hipSetDevice(c->device));
    int rc = resolve_timing(c);
    if (rc) return rc;
    for (int k = 0; k < n_buckets; k++) { ms[k] = k < VGL_N_TIMING_BUCKETS ? c->ms[k] : 0.0; launches[k] = k < VGL_N_TIMING_BUCKETS ? c->launches[k] : 0; }
    if (reset) for (int k = 0; k < VGL_N_TIMING_BUCKETS; k++) { c->ms[k] = 0; c->launches[k] = 0; }
    return VGL_OK;
}

extern "C" int vgl_pack_set_error(int code, const char* msg) { return fail(code, "%s", msg); }

// what this context launches (include/vcfgl_hip.h: vgl_ctx_info_t)
extern "C" int vgl_ctx_info(const vgl_ctx* c, vgl_ctx_info_t* out) {
    if (!c || !out) return fail(VGL_E_ARG, "null argument");
    if (out->size < (int32_t)sizeof(int32_t) * 2) return fail(VGL_E_ARG, "vgl_ctx_info_t.size must be set by the caller");
    vgl_ctx_info_t r;
    memset(&r, 0, sizeof r);
    const VglDevParams& D = c->dp;
    r.size = out->size < (int32_t)sizeof r ? out->size : (int32_t)sizeof r;
    r.abi_version = VGL_ABI_VERSION; r.device = c->device;
    r.n_samples = D.n_samples; r.max_sites_per_tile = c->max_sites; r.max_alleles = D.A; r.max_genotypes = D.G;
    r.rng_mode = c->p.rng_mode;
    r.depth_mode = D.serial ? VGL_DEPTH_SERIAL_SCOUT : D.depth_pre;
    r.fused = D.fused; r.fused_split = D.fused ? (D.fused_split > 0 ? D.fused_split : 1) : 0;
    r.sample_lean = D.serial ? 0 : (D.lean_ok ? ((D.error_qs == 2 && D.defer_ok) ? 2 : 1) : ((D.error_qs == 2 && D.defer_ok) ? 3 : 0));
    r.gl_sort = D.gl_sort; r.gl_wpb = D.gl2x ? 16 : ((D.gl_model == 2 && D.gl_wpb == 8) ? 8 : 4);   // 16: k_gl2 (sixteen natural wavefronts, two evaluations per thread)
    r.read_cap = D.read_cap; r.pool_cap = D.error_qs == 2 ? D.pool_cap : 0; r.pool_lds_bytes = D.error_qs == 2 ? D.pool_lds_bytes : 0;
#ifdef VGL_TEST_HOOKS
    r.test_hooks = 1;
#endif
    r.workspace_bytes = (int64_t)c->ws_bytes;
    r.rng_tile_max_sites = D.serial ? 0 : ((int64_t)1 << D.site_hash_bits);
    memcpy(out, &r, (size_t)r.size);
    return VGL_OK;
}

// VGL_RNG_SERIAL, --error-qs 2, std beta sampler: the beta deviates of the tile's reads in draw order
// (vgl_betachain.hip).  Synchronises the stream: the number of reads and each chunk's progress come back to the host.
static int run_beta_chain(vgl_ctx* c, const VglDevParams& D, int n_sites, hipStream_t st) {
    const long long E = (long long)n_sites * D.n_samples;
    if (vgl_chain_read_offsets(c->d_sdp, E, c->d_roff, c->d_rtotal, st)) return fail(VGL_E_NODEVICE, "k_read_offsets launch failed");
    long long R = 0;
    HIPCHK(hipMemcpyAsync(&R, c->d_rtotal, sizeof R, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (R == 0) return VGL_OK;
    if ((size_t)R > c->errp_lin_cap) {
        if (c->d_errp_lin) { HIPCHK(hipFree(c->d_errp_lin)); c->d_errp_lin = nullptr; }
        c->errp_lin_cap = (size_t)R + (size_t)R / 8 + 1024;
        if (dmalloc(&c->d_errp_lin, c->errp_lin_cap)) return VGL_E_NOMEM;
    }
    const long long margin = vgl_chain_margin_words(), seg = vgl_chain_seg();
    long long done = 0;
    while (done < R) {
        const long long remaining = R - done;
        // a deviate takes ~15-19 words on average; the chunk is sized for the rest of the tile, at most 2^28 words
        long long n_words = remaining * 24 + 4096 + margin;
        long long cap_words = 1LL << 28;
        if (hook_env("VGL_CHAIN_MAX_WORDS")) cap_words = atoll(hook_env("VGL_CHAIN_MAX_WORDS"));      // test hook: many small chunks
        if (n_words > cap_words) n_words = cap_words;
        n_words &= ~1LL;
        if (n_words > c->chain_words_cap) {
            void* old[] = {c->d_cw, c->d_ccons, c->d_cexit, c->d_ccnt, c->d_centry, c->d_cbase, c->d_cpos, c->d_csnap, c->d_csnapw};
            for (void* q : old) if (q) HIPCHK(hipFree(q));
            c->d_cw = nullptr; c->d_ccons = nullptr; c->d_cexit = nullptr; c->d_ccnt = nullptr; c->d_centry = nullptr; c->d_cbase = nullptr;
            c->d_cpos = nullptr; c->d_csnap = nullptr; c->d_csnapw = nullptr;
            const size_t npos = (size_t)n_words / 2, nseg = npos / (size_t)seg + 2, nsnap = (size_t)vgl_chain_snapshots_needed(n_words);
            if (dmalloc(&c->d_cw, (size_t)n_words) || dmalloc(&c->d_ccons, npos) || dmalloc(&c->d_cexit, nseg * 64) || dmalloc(&c->d_ccnt, nseg * 64) ||
                dmalloc(&c->d_centry, nseg) || dmalloc(&c->d_cbase, nseg) || dmalloc(&c->d_cpos, npos / 4 + 1024) ||
                dmalloc(&c->d_csnap, nsnap * 624) || dmalloc(&c->d_csnapw, nsnap + 1)) return VGL_E_NOMEM;
            c->chain_words_cap = n_words;
        }
        VglChainCtl h; memset(&h, 0, sizeof h);
        h.remaining = remaining; h.n_pos = (n_words - margin) / 2; h.n_seg = (int)((h.n_pos + seg - 1) / seg);
        HIPCHK(hipMemcpyAsync(c->d_cctl, &h, sizeof h, hipMemcpyHostToDevice, st));
        if (vgl_chain_chunk(&D, c->d_serial, c->d_cctl, c->d_cw, n_words, c->d_ccons, c->d_cexit, c->d_ccnt, c->d_centry, c->d_cbase, c->d_cpos,
                            c->d_csnap, c->d_csnapw, st)) return fail(VGL_E_NODEVICE, "beta chain launch failed: %s", hipGetErrorString(hipGetLastError()));
        HIPCHK(hipMemcpyAsync(&h, c->d_cctl, sizeof h, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (h.err) return fail(VGL_E_UNSUPPORTED, "a beta deviate consumed more generator words than the chain scheme allows");
        if (h.n_chunk <= 0 || h.endw <= 0) return fail(VGL_E_NODEVICE, "beta chain made no progress");
        if (vgl_chain_emit(&D, c->d_serial, c->d_cctl, c->d_cw, c->d_cpos, h.n_chunk, c->d_errp_lin + done, c->d_csnap, c->d_csnapw,
                           vgl_chain_snapshots_needed(n_words), st)) return fail(VGL_E_NODEVICE, "beta chain launch failed");
        done += h.n_chunk;
    }
    return VGL_OK;
}

extern "C" int vgl_simulate_tile_device(vgl_ctx* c, int64_t site0, int32_t n_sites, const uint8_t* gt,
                                        vgl_tile_out* o, void* stream) {
    if (!c || !o) return fail(VGL_E_ARG, "null argument");
    if (n_sites < 0 || n_sites > c->max_sites) return fail(VGL_E_ARG, "n_sites %d exceeds max_sites_per_tile %d", n_sites, c->max_sites);
    if (n_sites == 0) return VGL_OK;
    if (!gt || !o->site_status || !o->n_alleles || !o->alleles2acgt) return fail(VGL_E_ARG, "gt, site_status, n_alleles and alleles2acgt are required");
    if (site0 < 0) return fail(VGL_E_ARG, "site0 must be >= 0");
    if (!c->dp.serial) {
        // VGL_RNG_TILE windows are slices of ONE rand48 sequence of period 2^48: evaluation (site, sample) owns draws
        // [e block, (e + 1) block), e = H(site) n_samples + sample, H a permutation of [0, 2^W).  Past 2^W sites the windows would
        // silently repeat earlier ones.
        if ((uint64_t)site0 + (uint64_t)n_sites > (1ULL << c->dp.site_hash_bits))
            return fail(VGL_E_ARG, "VGL_RNG_TILE: sites [%lld, %lld) x %d samples x %llu draws per evaluation run past the 2^48 period of rand48 "
                        "(at most %llu sites with this layout); split the job over seeds or use a smaller layout.block",
                        (long long)site0, (long long)site0 + n_sites, c->dp.n_samples, (unsigned long long)c->p.layout.block,
                        (unsigned long long)(1ULL << c->dp.site_hash_bits));
    }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const VglDevParams& D = c->dp;
    VglTilePtrs T;
    memset(&T, 0, sizeof T);
    T.site0 = site0; T.n_sites = n_sites; T.gt = gt;
    T.reads = c->d_reads; T.errp = c->d_errp; T.ad4 = c->d_ad4; T.adf4 = c->d_adf4; T.qsum = c->d_qsum; T.qsumsq = c->d_qsumsq;
    T.acc = c->d_acc; T.sinfo = c->d_sinfo; T.rowmap = c->d_rowmap; T.rowmap8 = c->d_rowmap8; T.gl2_redo = c->d_gl2_redo; T.gl2_redo_list = c->d_gl2_list; T.gl2_redo_count = c->d_gl2_count; T.errflag = c->d_errflag; T.dbg = c->d_dbg; T.dp_pre = c->d_dp_pre;
    T.site_base = c->d_site_base; T.site_hash = c->d_site_hash; T.fslot = c->d_fslot;
    T.redo_list = c->d_redo_list; T.redo_count = c->d_redo_count; T.redo_cap = c->redo_cap; T.redo_bits = c->d_redo_bits;
    T.seg_list = c->d_seg_list;
    if (!D.serial && o->i16 && c->d_tail_base) { T.tail_base = c->d_tail_base; T.site_tail = c->d_site_tail; }   // (k_sitebase, k_tail; a tile without an I16 output skips both)
    if (D.serial) {
        const size_t E = (size_t)c->max_sites * D.n_samples;
        T.sst_hap = c->d_sst; T.sst_base = c->d_sst + E; T.sdp = c->d_sdp;
        T.site_thresh = c->d_site_thresh; T.scout_off = c->d_scout_dp; T.site_tail = c->d_site_tail;
        if (site0 != c->serial_next_site)
            return fail(VGL_E_ARG, "VGL_RNG_SERIAL consumes the streams in call order: expected site0 %lld, got %lld", (long long)c->serial_next_site, (long long)site0);
    }
    T.site_status = o->site_status; T.n_alleles = o->n_alleles; T.n_alleles_obs = o->n_alleles_obs; T.alleles2acgt = o->alleles2acgt;
    T.info_dp = o->info_dp; T.info_ad = o->info_ad; T.info_adf = o->info_adf; T.info_adr = o->info_adr;
    T.qs = o->qs; T.i16 = o->i16; T.fmt_dp = o->fmt_dp; T.gl = o->gl; T.pl = o->pl; T.gp = o->gp;
    T.fmt_ad = o->fmt_ad; T.fmt_adf = o->fmt_adf; T.fmt_adr = o->fmt_adr; T.pl_u8 = o->pl_u8;
    T.reads_out = o->read_capacity > 0 ? o->reads : nullptr;
    T.reads_out_cap = o->read_capacity > 0 ? (o->read_capacity < D.read_cap ? o->read_capacity : D.read_cap) : 0;
    if (o->read_capacity > D.read_cap && o->reads)
        HIPCHK(hipMemsetAsync(o->reads + (size_t)D.read_cap * n_sites * D.n_samples, 0xFF,
                              (size_t)(o->read_capacity - D.read_cap) * n_sites * D.n_samples, st));
    if ((o->qs && !D.need_qsum) || (o->i16 && !D.need_qsumsq))
        return fail(VGL_E_ARG, "qs / i16 outputs need -addQS / -addI16 in the context parameters");
    // dumps of the deviates (ABI 2): the per-read error probabilities go through the --precise-gl staging planes
    const bool dump_errp = o->read_errp && o->read_capacity > 0 && D.error_qs == 2;
    AcctScope acct(c);
    if (dump_errp && !c->d_errp && dmalloc(&c->d_errp, (size_t)c->max_sites * D.n_samples * D.read_cap))
        return fail(VGL_E_NOMEM, "out of device memory (read_errp staging)");
    const bool errp_always = (c->p.precise_gl || (D.serial && !D.beta_chain)) && D.error_qs == 2;    // as sized by vgl_ctx_create
    T.errp = (errp_always || dump_errp) ? c->d_errp : nullptr;
    T.site_pick_err = (D.error_qs == 1) ? o->site_pick_err : nullptr;

    // the timing events belong to this call until the last one is recorded: any early return below destroys them (a failed call
    // leaks nothing), the successful end hands them to the context
    struct EvGuard { hipEvent_t e[VGL_NEV]; bool armed = true;
                     EvGuard() { for (int k = 0; k < VGL_NEV; k++) e[k] = nullptr; }
                     ~EvGuard() { if (armed) for (int k = 0; k < VGL_NEV; k++) if (e[k]) (void)hipEventDestroy(e[k]); } } evg;
    hipEvent_t* const e = evg.e;
    if (c->timing) for (int k = 0; k < VGL_NEV; k++) HIPCHK(hipEventCreate(&e[k]));
    HIPCHK(hipMemsetAsync(c->d_acc, 0, sizeof(int32_t) * VGL_ACC_STRIDE * (size_t)n_sites, st));
    if (c->d_redo_count) HIPCHK(hipMemsetAsync(c->d_redo_count, 0, sizeof(uint32_t) * VGL_REDO_PARTS * VGL_REDO_STRIDE, st));
    if (c->d_fslot) HIPCHK(hipMemsetAsync(c->d_fslot, 0, sizeof(unsigned long long) * 2 * (size_t)D.fused_split * (size_t)n_sites, st));
    if (c->timing) HIPCHK(hipEventRecord(e[VGL_T_DEPTH], st));        // depth draws ahead of k_sample (k_sitebase + k_depth; the scouts in serial mode)
    if (D.serial) {
        if (vgl_launch_scout(&D, &T, c->d_serial, st)) return fail(VGL_E_NODEVICE, "k_scout launch failed");
        c->serial_next_site = site0 + n_sites;
        if (D.beta_chain) {
            const int rc = run_beta_chain(c, D, n_sites, st);
            if (rc != VGL_OK) return rc;
            T.roff = c->d_roff; T.errp_lin = c->d_errp_lin;
        }
    } else {
        if (vgl_launch_sitebase(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_sitebase launch failed");
        if (D.depth_pre == 1 && vgl_launch_depth(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_depth launch failed");
    }
    const bool fused = D.fused && !T.reads_out && !o->qs && !o->i16 && !dump_errp;
    if (c->timing) HIPCHK(hipEventRecord(e[VGL_T_SAMPLE], st));
    if (!fused && vgl_launch_sample(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_sample launch failed: %s", hipGetErrorString(hipGetLastError()));
    if (c->timing) HIPCHK(hipEventRecord(e[VGL_T_REDO], st));
    if (!fused && vgl_launch_redo(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_redo launch failed");
    // INFO/I16 tail distances in tile mode (k_tail wants the LAST read's base): with the other site aggregates, behind k_gl -- unless k_gl's GL model 1
    // path may shuffle a deep evaluation's staged reads in place (gl1_deep), then ahead of it
    if (T.tail_base && D.gl1_deep && vgl_launch_tail(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_tail launch failed");
    if (c->timing) HIPCHK(hipEventRecord(e[VGL_T_SITE], st));
    if (!fused && vgl_launch_site(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_site launch failed");
    if (c->timing) HIPCHK(hipEventRecord(e[VGL_T_GL], st));
    if (D.serial && D.gl1_deep) {                            // where each deep evaluation's shuffle starts in htslib's stream
        if (vgl_launch_hts_offsets(&D, &T, c->d_serial, c->d_hts_off, c->d_hts_base, st)) return fail(VGL_E_NODEVICE, "k_hts_offsets launch failed");
        T.hts_off = c->d_hts_off; T.hts_base = c->d_hts_base;
    }
    if (fused) { if (vgl_launch_fused(&D, &T, st)) return fail(VGL_E_NODEVICE, "fused k_gl launch failed"); }
    else if (vgl_launch_gl(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_gl launch failed");
    if (c->timing) HIPCHK(hipEventRecord(e[VGL_T_SITEAGG], st));
    if (T.tail_base && !D.gl1_deep && vgl_launch_tail(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_tail launch failed");
    if (o->qs || o->i16) if (vgl_launch_siteagg(&D, &T, st)) return fail(VGL_E_NODEVICE, "k_siteagg launch failed");
    if (dump_errp) {
        const size_t row = (size_t)n_sites * D.n_samples;
        const size_t rows = (size_t)(o->read_capacity < D.read_cap ? o->read_capacity : D.read_cap);
        if (vgl_launch_errp_dump(&D, c->d_errp, o->read_errp, row, (int)rows, st)) return fail(VGL_E_NODEVICE, "k_errp_dump launch failed");
        if ((size_t)o->read_capacity > rows)
            HIPCHK(hipMemsetAsync(o->read_errp + rows * row, 0xFF, ((size_t)o->read_capacity - rows) * row * sizeof(double), st));
    }
    if (c->timing) HIPCHK(hipEventRecord(e[VGL_NEV - 1], st));
    if (c->timing) for (int k = 0; k < VGL_NEV; k++) c->ev.push_back(e[k]);
    evg.armed = false;
    return VGL_OK;
}

#ifdef VGL_TEST_HOOKS
// diagnostic (not in the public header): the beta deviates of the last serial tile in draw order
extern "C" __attribute__((visibility("default"))) long long vgl_dbg_chain(vgl_ctx* c, double* out, long long n) {
    if (!c || !c->d_errp_lin) return -1;
    long long R = 0;
    if (hipMemcpy(&R, c->d_rtotal, sizeof R, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (n > R) n = R;
    if (hipMemcpy(out, c->d_errp_lin, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return R;
}

// diagnostic (not in the public header): the generator states in front of the windows of the last tile's sites (k_sitebase)
extern "C" __attribute__((visibility("default"))) int vgl_dbg_site_base(vgl_ctx* c, uint64_t* out, int n) {
    if (!c || !c->d_site_base || n > c->max_sites) return VGL_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, c->d_site_base, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
    return VGL_OK;
}

// diagnostic (not in the public header): read and clear the VGL_DEBUG_STAMPS counters
extern "C" __attribute__((visibility("default"))) int vgl_dbg_stamps(vgl_ctx* c, unsigned long long out[16]) {
    if (!c || !c->d_dbg) return fail(VGL_E_ARG, "context was not created with VGL_DEBUG_STAMPS=1");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, c->d_dbg, 128, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(c->d_dbg, 0, 128));
    return VGL_OK;
}

// diagnostic (not part of the C ABI): entries of the last tile's redo list (k_sample<2, deferred> -> k_redo)
extern "C" __attribute__((visibility("default"))) int vgl_dbg_redo_count(vgl_ctx* c, unsigned* n) {
    if (!c || !n) return VGL_E_ARG;
    *n = 0;
    if (!c->d_redo_count) return VGL_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    uint32_t h[VGL_REDO_PARTS * VGL_REDO_STRIDE];                         // one counter per partition of the list
    HIPCHK(hipMemcpy(h, c->d_redo_count, sizeof h, hipMemcpyDeviceToHost));
    for (int p = 0; p < VGL_REDO_PARTS; ++p) *n += h[p * VGL_REDO_STRIDE];
    return VGL_OK;
}
#endif

extern "C" int vgl_ctx_check(vgl_ctx* c, void* stream) {
    if (!c) return fail(VGL_E_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    uint32_t flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, c->d_errflag, sizeof flag, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (flag) HIPCHK(hipMemsetAsync(c->d_errflag, 0, sizeof flag, (hipStream_t)stream));
    if (flag & VGL_DEVERR_CAPACITY) return fail(VGL_E_CAPACITY, "a simulated read depth exceeded the staging capacity of %d reads per sample", c->dp.read_cap);
    if (flag & VGL_DEVERR_QSBIN) return fail(VGL_E_QSBIN, "Could not find a range for a simulated qs value in --qs-bins");
    if (flag & VGL_DEVERR_ADJQ) return fail(VGL_E_ADJQ, "--adjust-qs %d: a read has no valid adjusted quality score (error probability exactly 0 or 1, or a negative adjusted score)", c->dp.adjust_qs);
    if (flag & VGL_DEVERR_INTERNAL) return fail(VGL_E_NODEVICE, "internal: a kernel's LDS layout assumption does not hold on this build (k_sample<2>)");
    return VGL_OK;
}

// field table of vgl_tile_out in declaration order: element size and per-tile element count
struct FieldDesc { size_t off; size_t esz; int kind; };
enum { K_SITE, K_SITE5, K_SITEA, K_SITE16, K_EVAL, K_PLANEG, K_PLANEA };
enum { N_FIELDS = 18 };
static const FieldDesc FIELDS[N_FIELDS] = {
    {offsetof(vgl_tile_out, site_status), 4, K_SITE}, {offsetof(vgl_tile_out, n_alleles), 4, K_SITE},
    {offsetof(vgl_tile_out, n_alleles_obs), 4, K_SITE}, {offsetof(vgl_tile_out, alleles2acgt), 1, K_SITE5},
    {offsetof(vgl_tile_out, info_dp), 4, K_SITE}, {offsetof(vgl_tile_out, info_ad), 4, K_SITEA},
    {offsetof(vgl_tile_out, info_adf), 4, K_SITEA}, {offsetof(vgl_tile_out, info_adr), 4, K_SITEA},
    {offsetof(vgl_tile_out, qs), 4, K_SITEA}, {offsetof(vgl_tile_out, i16), 4, K_SITE16},
    {offsetof(vgl_tile_out, fmt_dp), 4, K_EVAL}, {offsetof(vgl_tile_out, gl), 4, K_PLANEG},
    {offsetof(vgl_tile_out, pl), 4, K_PLANEG}, {offsetof(vgl_tile_out, gp), 4, K_PLANEG},
    {offsetof(vgl_tile_out, fmt_ad), 4, K_PLANEA}, {offsetof(vgl_tile_out, fmt_adf), 4, K_PLANEA},
    {offsetof(vgl_tile_out, fmt_adr), 4, K_PLANEA}, {offsetof(vgl_tile_out, pl_u8), 1, K_PLANEG},
};
static size_t field_count(const vgl_ctx* c, int kind, size_t n_sites) {
    const size_t N = c->dp.n_samples, A = c->dp.A, G = c->dp.G;
    switch (kind) {
        case K_SITE: return n_sites; case K_SITE5: return n_sites * 5; case K_SITEA: return n_sites * A;
        case K_SITE16: return n_sites * 16; case K_EVAL: return n_sites * N; case K_PLANEG: return n_sites * G * N;
        default: return n_sites * A * N;
    }
}

static int flags_to_rc(vgl_ctx* c, uint32_t flag) {
    if (flag & VGL_DEVERR_CAPACITY) return fail(VGL_E_CAPACITY, "a simulated read depth exceeded the staging capacity of %d reads per sample", c->dp.read_cap);
    if (flag & VGL_DEVERR_QSBIN) return fail(VGL_E_QSBIN, "Could not find a range for a simulated qs value in --qs-bins");
    if (flag & VGL_DEVERR_ADJQ) return fail(VGL_E_ADJQ, "--adjust-qs %d: a read has no valid adjusted quality score (error probability exactly 0 or 1, or a negative adjusted score)", c->dp.adjust_qs);
    if (flag & VGL_DEVERR_INTERNAL) return fail(VGL_E_NODEVICE, "internal: a kernel's LDS layout assumption does not hold on this build (k_sample<2>)");
    return VGL_OK;
}

extern "C" void* vgl_host_alloc(size_t bytes) {
    void* p = nullptr;
    // default flags: page-locked, placed on the host NUMA node nearest to the calling thread's current device (measured: 53 GB/s
    // of DMA into it against 35 GB/s into hipHostMallocPortable memory on the two-socket box); every device of the process can
    // still write it.  VGL_HOST_ALLOC_FLAGS overrides (diagnostic).
    const unsigned flags = hook_env("VGL_HOST_ALLOC_FLAGS") ? (unsigned)strtoul(hook_env("VGL_HOST_ALLOC_FLAGS"), nullptr, 0) : hipHostMallocDefault;
    if (hipHostMalloc(&p, bytes ? bytes : 1, flags) != hipSuccess) { fail(VGL_E_NOMEM, "hipHostMalloc of %zu bytes failed", bytes); return nullptr; }
    return p;
}
extern "C" void* vgl_host_alloc_on(int32_t device, size_t bytes) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess || hipSetDevice(device) != hipSuccess) { fail(VGL_E_NODEVICE, "device %d is not available", device); return nullptr; }
    void* p = vgl_host_alloc(bytes);
    (void)hipSetDevice(cur);
    return p;
}
extern "C" void vgl_host_free(void* p) { if (p) (void)hipHostFree(p); }

// Host buffers in, host buffers out, asynchronously: the tile's kernels are enqueued on the context's compute stream, the copies of
// its tags back to the host on its copy stream behind them; with two tiles in flight the copies of tile t overlap the kernels of
// tile t + 1.  Destination buffers from vgl_host_alloc() (pinned) are written by DMA directly; pageable ones work, more slowly.
// the fallible part of vgl_simulate_tile_async, from the first enqueue on (its caller cleans up after a failure)
static int enqueue_host_tile(vgl_ctx* c, vgl_ctx::HostSlot& S, int64_t site0, int32_t n_sites, const uint8_t* gt, vgl_tile_out* o) {
    const size_t N = c->dp.n_samples;
    memcpy(S.h_gt, gt, (size_t)n_sites * N);
    HIPCHK(hipMemcpyAsync(S.d_gt, S.h_gt, (size_t)n_sites * N, hipMemcpyHostToDevice, c->s_compute));
    vgl_tile_out d;
    memset(&d, 0, sizeof d);
    for (int f = 0; f < N_FIELDS; f++) {
        void* host = *(void**)((char*)o + FIELDS[f].off);
        if (!host) continue;
        const size_t need = field_count(c, FIELDS[f].kind, (size_t)c->max_sites) * FIELDS[f].esz;
        if (S.d_out_bytes[f] < need) {
            if (S.d_out[f]) (void)hipFree(S.d_out[f]);
            c->ws_bytes -= S.d_out_bytes[f];                 // vgl_ctx_info.workspace_bytes counts these buffers too (the largest of an all-tags run)
            S.d_out[f] = nullptr; S.d_out_bytes[f] = 0;
            HIPCHK(hipMalloc(&S.d_out[f], need));
            c->ws_bytes += need;
            // VGL_LAYOUT_SAMPLE_MAJOR: the kernels write n_samples x nK(site) values of a slab, the copy below takes the slab whole --
            // what lies behind a record's array is then zeros from here, not another job's memory (once per buffer, not per tile)
            HIPCHK(hipMemsetAsync(S.d_out[f], 0, need, c->s_compute));
            S.d_out_bytes[f] = need;
        }
        *(void**)((char*)&d + FIELDS[f].off) = S.d_out[f];
    }
    if (o->reads && o->read_capacity > 0) {
        const size_t need = (size_t)o->read_capacity * c->max_sites * N;
        if (S.d_reads_out_bytes < need) {
            if (S.d_reads_out) (void)hipFree(S.d_reads_out);
            S.d_reads_out = nullptr; S.d_reads_out_bytes = 0;
            HIPCHK(hipMalloc((void**)&S.d_reads_out, need));
            S.d_reads_out_bytes = need;
        }
        d.reads = S.d_reads_out; d.read_capacity = o->read_capacity;
    }
    if (o->read_errp && o->read_capacity > 0) {
        const size_t need = (size_t)o->read_capacity * c->max_sites * N * sizeof(double);
        if (S.d_errp_out_bytes < need) {
            if (S.d_errp_out) (void)hipFree(S.d_errp_out);
            S.d_errp_out = nullptr; S.d_errp_out_bytes = 0;
            HIPCHK(hipMalloc((void**)&S.d_errp_out, need));
            S.d_errp_out_bytes = need;
        }
        d.read_errp = S.d_errp_out; d.read_capacity = o->read_capacity;
    }
    if (o->site_pick_err) {
        if (!S.d_pick_out) HIPCHK(hipMalloc((void**)&S.d_pick_out, (size_t)c->max_sites * sizeof(double)));
        HIPCHK(hipMemsetAsync(S.d_pick_out, 0xFF, (size_t)n_sites * sizeof(double), c->s_compute));
        d.site_pick_err = S.d_pick_out;
    }
    int rc = vgl_simulate_tile_device(c, site0, n_sites, S.d_gt, &d, c->s_compute);
    if (rc) return rc;
    // this tile's device error flags, then a clean word for the next tile
    HIPCHK(hipMemcpyAsync(S.h_flag, c->d_errflag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->s_compute));
    HIPCHK(hipMemsetAsync(c->d_errflag, 0, sizeof(uint32_t), c->s_compute));
    HIPCHK(hipEventRecord(S.ev_kernels, c->s_compute));
    HIPCHK(hipStreamWaitEvent(c->s_copy, S.ev_kernels, 0));
    for (int f = 0; f < N_FIELDS; f++) {
        void* host = *(void**)((char*)o + FIELDS[f].off);
        if (!host) continue;
        HIPCHK(hipMemcpyAsync(host, S.d_out[f], field_count(c, FIELDS[f].kind, (size_t)n_sites) * FIELDS[f].esz, hipMemcpyDeviceToHost, c->s_copy));
    }
    if (d.reads) HIPCHK(hipMemcpyAsync(o->reads, d.reads, (size_t)o->read_capacity * n_sites * N, hipMemcpyDeviceToHost, c->s_copy));
    if (d.read_errp && c->dp.error_qs == 2) HIPCHK(hipMemcpyAsync(o->read_errp, d.read_errp, (size_t)o->read_capacity * n_sites * N * sizeof(double), hipMemcpyDeviceToHost, c->s_copy));
    if (d.site_pick_err) HIPCHK(hipMemcpyAsync(o->site_pick_err, d.site_pick_err, (size_t)n_sites * sizeof(double), hipMemcpyDeviceToHost, c->s_copy));
    HIPCHK(hipEventRecord(S.ev_copied, c->s_copy));
    return VGL_OK;
}

// Host buffers in, host buffers out, asynchronously: the tile's kernels are enqueued on the context's compute stream, the copies of
// its tags back to the host on its copy stream behind them; with two tiles in flight the copies of tile t overlap the kernels of
// tile t + 1.  Destination buffers from vgl_host_alloc() (pinned) are written by DMA directly; pageable ones work, more slowly.
// The ticket and the slot are committed only when everything is enqueued: after a failure part-way the streams are drained, the
// sticky device error word is cleared and the slot is free again -- no later tile inherits this one's flags or shares its buffers
// with work still in flight.
extern "C" int vgl_simulate_tile_async(vgl_ctx* c, int64_t site0, int32_t n_sites, const uint8_t* gt, vgl_tile_out* o, int32_t* ticket) {
    if (!c || !o || !ticket) return fail(VGL_E_ARG, "null argument");
    if (n_sites < 0 || n_sites > c->max_sites) return fail(VGL_E_ARG, "n_sites %d exceeds max_sites_per_tile %d", n_sites, c->max_sites);
    if (n_sites > 0 && !gt) return fail(VGL_E_ARG, "null gt");
    HIPCHK(hipSetDevice(c->device));
    const int k = c->next_slot;
    vgl_ctx::HostSlot& S = c->slot[k];
    if (S.busy) return fail(VGL_E_ARG, "two tiles are already in flight: vgl_tile_wait() the older one first");
    if (!c->s_compute) { HIPCHK(hipStreamCreateWithFlags(&c->s_compute, hipStreamNonBlocking)); HIPCHK(hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking)); }
    if (!S.ev_kernels) { HIPCHK(hipEventCreateWithFlags(&S.ev_kernels, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&S.ev_copied, hipEventDisableTiming)); }
    const size_t N = c->dp.n_samples;
    if (!S.d_gt) HIPCHK(hipMalloc((void**)&S.d_gt, (size_t)c->max_sites * N));
    if (!S.h_gt) HIPCHK(hipHostMalloc((void**)&S.h_gt, (size_t)c->max_sites * N, hipHostMallocDefault));
    if (!S.h_flag) HIPCHK(hipHostMalloc((void**)&S.h_flag, sizeof(uint32_t), hipHostMallocDefault));
    S.rc = VGL_OK; *S.h_flag = 0;
    S.site0 = site0; S.n_sites = n_sites; S.o = *o;
    if (n_sites == 0) HIPCHK(hipEventRecord(S.ev_copied, c->s_copy));
    else {
        const int rc = enqueue_host_tile(c, S, site0, n_sites, gt, o);
        if (rc != VGL_OK) {
            char keep[sizeof g_err];
            memcpy(keep, g_err, sizeof keep);                        // the first error is the one to report
            (void)hipStreamSynchronize(c->s_compute);
            (void)hipStreamSynchronize(c->s_copy);
            (void)hipMemset(c->d_errflag, 0, sizeof(uint32_t));
            memcpy(g_err, keep, sizeof keep);
            return rc;
        }
    }
    *ticket = k; c->next_slot = k ^ 1;
    S.busy = true;
    return VGL_OK;
}

// A tile whose device flags report a draw deeper than the staging capacity, run again through the sibling context (host buffers: the slot's own copy of
// the genotypes, the caller's output arrays), VGL_DEEP_TILE_SITES sites at a time -- every tag array of a tile is site-major, so a sub-tile is a slice of
// it.  Returns VGL_E_CAPACITY (quietly) where that cannot be done: serial mode (the streams have moved on), a per-read dump (read-major planes of the
// caller's own capacity), a capacity already at the layout's maximum, or no memory for the sibling.
static int deep_rerun(vgl_ctx* c, vgl_ctx::HostSlot& S) {
    const VglDevParams& D = c->dp;
    if (D.serial || D.read_cap >= VGL_READ_CAP_MAX || (S.o.read_capacity > 0 && (S.o.reads || S.o.read_errp))) return VGL_E_CAPACITY;
    if (!c->deep) {
        vgl_params p = c->p;
        std::vector<double> depths; std::vector<int32_t> bins;
        if (c->depths_copy.size()) { depths = c->depths_copy; p.depths = depths.data(); }
        if (c->bins_copy.size()) { bins = c->bins_copy; p.qs_bins = bins.data(); }
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        const int rc = ctx_create_cap(&p, c->device, c->max_sites < VGL_DEEP_TILE_SITES ? c->max_sites : VGL_DEEP_TILE_SITES, &c->deep, VGL_READ_CAP_MAX);
        if (rc != VGL_OK) { c->deep = nullptr; memcpy(g_err, keep, sizeof keep); return VGL_E_CAPACITY; }
    }
    const size_t N = (size_t)D.n_samples;
    for (int32_t k = 0; k < S.n_sites; k += c->deep->max_sites) {
        const int32_t n = (S.n_sites - k < c->deep->max_sites) ? (S.n_sites - k) : c->deep->max_sites;
        vgl_tile_out o = S.o;
        for (int f = 0; f < N_FIELDS; f++) {
            char* host = *(char**)((char*)&S.o + FIELDS[f].off);
            if (host) *(char**)((char*)&o + FIELDS[f].off) = host + field_count(c, FIELDS[f].kind, (size_t)k) * FIELDS[f].esz;
        }
        if (o.site_pick_err) o.site_pick_err += k;
        const int rc = vgl_simulate_tile(c->deep, S.site0 + k, n, S.h_gt + (size_t)k * N, &o);
        if (rc != VGL_OK) return rc;                                 // (a draw beyond VGL_READ_CAP_MAX reads: VGL_E_CAPACITY after all)
    }
    c->deep_runs++;
    return VGL_OK;
}

extern "C" int vgl_tile_wait(vgl_ctx* c, int32_t ticket) {
    if (!c || ticket < 0 || ticket > 1) return fail(VGL_E_ARG, "bad ticket");
    vgl_ctx::HostSlot& S = c->slot[ticket];
    if (!S.busy) return fail(VGL_E_ARG, "no tile in flight under this ticket");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(S.ev_copied));
    S.busy = false;
    if ((*S.h_flag & VGL_DEVERR_CAPACITY) && S.n_sites > 0) {
        const int rc = deep_rerun(c, S);
        if (rc != VGL_E_CAPACITY) return rc;                         // done (or failed for another reason, reported as such)
    }
    return flags_to_rc(c, *S.h_flag);
}

extern "C" int vgl_simulate_tile(vgl_ctx* c, int64_t site0, int32_t n_sites, const uint8_t* gt, vgl_tile_out* o) {
    if (c) for (int k = 0; k < 2; k++) if (c->slot[k].busy) return fail(VGL_E_ARG, "vgl_simulate_tile with a tile in flight: vgl_tile_wait() it first");
    int32_t t = 0;
    int rc = vgl_simulate_tile_async(c, site0, n_sites, gt, o, &t);
    if (rc) return rc;
    return vgl_tile_wait(c, t);
}
