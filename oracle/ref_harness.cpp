/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE.  Thin extern "C" shim around the reference's OWN
 * sampler code, compiled from where it lies:  #include "<REF>/rng.h"  +  <REF>/shared.cpp.
 * Nothing of the reference is copied into this repository; the resulting shared objects go
 * to oracle/_ref/ (git-ignored).  rng.h needs only the four rand48 seeder arrays that the
 * reference defines in vcfgl.cpp:214-219; they are defined here with the same initialiser.
 *
 * gl_methods.cpp / vcfgl.cpp are NOT buildable here: they include htslib headers
 * (bcf_utils.h -> htslib/vcf.h) and the image has no htslib; no stand-in headers are written.
 *
 * Built twice by oracle/Makefile:  libvgl_ref_stdbeta.so (reference default,
 * __USE_STD_BETA__==1) and libvgl_ref_rand48beta.so (-D__USE_STD_BETA__=0).
 */
#include <stdlib.h>
#include <stdio.h>
#include "rng.h"          /* -I<reference root> */

unsigned short int rng1_seeder[3] = SEEDER_INIT;
unsigned short int rng1_seeder_save[3] = SEEDER_INIT;
unsigned short int rng2_seeder[3] = SEEDER_INIT;
unsigned short int rng2_seeder_save[3] = SEEDER_INIT;

extern "C" {

/* io.cpp:1054-1061 */
void ref_seed(int seed) {
    srand48(seed);
    rng1_seeder[0] = 0x330e; rng2_seeder[0] = 0x330e;
    rng1_seeder[1] = (unsigned short)((long)seed);
    rng1_seeder[2] = (unsigned short)(((long)seed) >> 16);
    rng2_seeder[1] = (unsigned short)((long)seed);
    rng2_seeder[2] = (unsigned short)(((long)seed) >> 16);
}

double ref_rng0(void) { return sample_uniform_rng0(); }
double ref_rng1(void) { return sample_uniform_rng1(); }
double ref_rng2(void) { return sample_uniform_rng2(); }
double ref_gamma_ln(double x) { return gamma_ln(x); }

/* rng.h:284-316 on rng1 */
void ref_poisson(double lambda, int n, int* out) {
    PoissonSampler* p = PoissonSampler_init(lambda);
    poissonSampler_sample_depths_same_mean(p, out, n);
    free(p);
}

/* rng.h:318-351 */
void ref_poisson_multi(const double* lambdas, int n, int* out) {
    PoissonSampler** mp = (PoissonSampler**)malloc(sizeof(PoissonSampler*) * n);
    for (int i = 0; i < n; i++) mp[i] = PoissonSampler_init(lambdas[i]);
    poissonSampler_sample_depths_perSample_means(mp, out, n);
    for (int i = 0; i < n; i++) free(mp[i]);
    free(mp);
}

int ref_uses_std_beta(void) { return __USE_STD_BETA__; }

void ref_beta(double mean, double var, int seed, int n, double* out) {
    FILE* devnull = fopen("/dev/null", "w");
    FILE* saved = stderr;
    stderr = devnull;                       /* the constructor prints the shape parameters */
#if __USE_STD_BETA__ == 1
    BetaSampler* b = new BetaSampler(mean, var, seed, devnull);
    for (int i = 0; i < n; i++) out[i] = b->sample();
    delete b;
#else
    BetaSampler* b = BetaSampler_init(mean, var, seed, devnull);
    /* rng.h:155-173 leaves old_alpha unset; give it the intended value so alpha<1 is defined */
    b->gamma_x->old_alpha = b->alpha;
    b->gamma_y->old_alpha = b->beta;
    for (int i = 0; i < n; i++) out[i] = b->sample();
    BetaSampler_destroy(b);
#endif
    stderr = saved;
    fclose(devnull);
}

/* shared.cpp tables */
double ref_q2log10gl(int row, int q) { return qScore_to_log10_gl[row][q]; }
int ref_qs2(int q) { return QS_TO_QSSQ(q); }
double ref_qs_to_errprob(int q) { return QS_TO_ERRPROB(q); }      /* shared.h:493 over shared.cpp:31 */
int ref_ngt(int n) { return NALLELES_TO_NGTS(n); }

}
