// tools/microbench/gl_chain.hip -- what one wavefront's GL model 2 read loop costs on the vector pipe, without memory: per read and accumulator
// (float)((double)acc + term), the maximum, the subtraction (gl_methods.cpp:22-59 as k_gl runs it), NA accumulators, 1 / 2 / 4 / 8 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/gl_chain tools/microbench/gl_chain.hip && /tmp/gl_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#define READS 4096
template <int NA>
__global__ void k_chain(unsigned long long* out, const double* terms, float* sink, int sel) {
    float tr[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) tr[i] = -0.0f;
    const double t0 = terms[0], t1 = terms[1], t2 = terms[2];
    unsigned s = sel + threadIdx.x;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < READS; ++r) {
        s = s * 1664525u + 1013904223u;
        const bool b = (s >> 31) != 0;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const double t = (i % 3 == 0) ? (b ? t0 : t2) : ((i % 3 == 1) ? t1 : (b ? t2 : t0));
            const float v = (float)((double)tr[i] + t);
            tr[i] = v;
            mx = __builtin_fmaxf(mx, v);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) tr[i] -= mx;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) acc += tr[i];
    if (acc == 12345.0f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = c1 - c0;
}
template <int NA> static void run(unsigned long long* d, const double* dt, float* ds) {
    for (int wps : {1, 2, 4, 8}) {
        const int waves = 4 * wps;
        unsigned long long h[32];
        for (int rep = 0; rep < 2; ++rep) { k_chain<NA><<<1, 64 * waves>>>(d, dt, ds, 7); hipDeviceSynchronize(); }
        hipMemcpy(h, d, 8 * waves, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
        // s_memtime counts at 100 MHz on this part; report both raw ticks and ns per read
        printf("NA=%2d waves/SIMD %d: %.3f ticks per read per wave, %.3f per read of SIMD time\n", NA, wps, (double)mx / READS, (double)mx / READS / wps);
    }
}
int main() {
    unsigned long long* d; double* dt; float* ds;
    hipMalloc(&d, 8 * 64); hipMalloc(&dt, 24); hipMalloc(&ds, 4);
    const double t[3] = {-0.0004345118, -0.3024732, -2.477121};
    hipMemcpy(dt, t, 24, hipMemcpyHostToDevice);
    run<3>(d, dt, ds); run<6>(d, dt, ds); run<10>(d, dt, ds); run<15>(d, dt, ds);
    // clock of s_memtime: time a known interval
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long h[32];
    hipEventRecord(e0); k_chain<6><<<1, 64>>>(d, dt, ds, 7); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("one wave NA=6: %llu ticks in %.3f ms of launch (tick = %.2f ns at most)\n", h[0], ms, ms * 1e6 / (double)h[0]);
    return 0;
}
