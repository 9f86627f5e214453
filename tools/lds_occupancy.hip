// Diagnostic: how many one-wavefront workgroups with a given dynamic LDS size fit one CU (the pool size of k_sample<2> is chosen
// from this).  usage (GPU box): hipcc --offload-arch=gfx950 -o build/lds_occ tools/lds_occupancy.hip && build/lds_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o) { extern __shared__ int s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (o) o[threadIdx.x] = s[63 - threadIdx.x]; }
int main() {
    for (int b = 7168; b <= 12288; b += 128) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, b) != hipSuccess) { printf("error at %d\n", b); return 1; }
        static int last = -1;
        if (n != last) printf("dynamic LDS %5d B: %d workgroups per CU\n", b, n);
        last = n;
    }
    return 0;
}
