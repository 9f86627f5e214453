// vgl_sample_seg.hip -- the k_sample_seg kernels (the float32 builds of k_sample<2> as one pool segment per wavefront + a follow-up kernel, DESIGN.md
// section 4.2) as a translation unit of their own: vgl_sample.hip with VGL_SAMPLE_SEG_TU builds the body template, these kernels and their launcher
// (vgl_sample_seg_launch) and nothing else.  Why: the Makefile compiles this file with SEGFLAGS -- an LLVM scheduling strategy applies to a whole module,
// and the one that suits the pool loop cost k_depth and k_sample<0> time when it was set for all of vgl_sample.hip (docs/tried.md, round 6).
#define VGL_SAMPLE_SEG_TU 1
#include "vgl_sample.hip"
