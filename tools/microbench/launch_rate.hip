// tools/microbench/launch_rate.hip -- how fast does the dispatcher start one-wavefront workgroups, with and without dynamic LDS?
// (k_sample<2> launches 1 048 576 workgroups of 64 threads with 7944 B of LDS each per 65536 x 1000 tile)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_rate tools/microbench/launch_rate.hip && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ unsigned char lds[];
__global__ void k_empty(unsigned* out, int spin) {
    unsigned x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
    if (x == 0x12345u) { lds[threadIdx.x] = 1; out[0] = lds[63 - threadIdx.x]; }
}
int main() {
    unsigned* d; hipMalloc(&d, 4);
    const int n = 1 << 20;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ldss[] = {0, 1024, 4096, 7944, 16384, 32768};
    const int spins[] = {0, 1000, 10000};
    for (int sp : spins) for (int l : ldss) {
        hipLaunchKernelGGL(k_empty, dim3(n), dim3(64), l, 0, d, sp);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_empty, dim3(n), dim3(64), l, 0, d, sp);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("spin %6d  lds %6d B: %8.3f ms per launch of %d workgroups = %7.1f workgroups/us\n", sp, l, ms, n, n / (ms * 1e3));
    }
    return 0;
}
