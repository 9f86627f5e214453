// Diagnostic: HBM write rate of the store shapes k_gl could use for its planes (N samples per row, K rows per site), no compute.
//   rows4   what k_gl does: a wavefront = 64 consecutive samples of one site, one 4-byte store per lane and row (256 B runs, row stride 4 N)
//   rows16  a wavefront = 256 consecutive samples, one 16-byte store per lane and row (1 KB runs)
//   flat16  a plain fill: 16 bytes per lane, consecutive wavefronts consecutive (the box's fill rate)
// usage (GPU box): hipcc -O3 --offload-arch=gfx950 -o build/store_pattern tools/store_pattern.hip && build/store_pattern [N] [K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(512) void rows4(uint32_t* out, int N, int K, int chunks, long waves) {
    const long w = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
    if (w >= waves) return;
    const long site = w / chunks; const int c = (int)(w - site * chunks);
    const int s = c * 64 + (threadIdx.x & 63);
    if (s >= N) return;
    uint32_t* p = out + site * K * N + s;
    for (int k = 0; k < K; ++k) p[(long)k * N] = (uint32_t)(k + s);
}
__global__ __launch_bounds__(512) void rows16(uint32_t* out, int N, int K, int chunks4, long waves) {
    const long w = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
    if (w >= waves) return;
    const long site = w / chunks4; const int c = (int)(w - site * chunks4);
    const int s = c * 256 + (threadIdx.x & 63) * 4;
    if (s + 3 >= N) return;
    uint32_t* p = out + site * K * N + s;
    for (int k = 0; k < K; ++k) { uint4 v = {(uint32_t)k, (uint32_t)s, 0u, 1u}; __builtin_memcpy(p + (long)k * N, &v, 16); }     // (16-byte aligned when N % 4 == 0)
}
__global__ __launch_bounds__(512) void flat16(uint4* out, long n) {
    const long i = (long)blockIdx.x * 512 + threadIdx.x;
    if (i < n) out[i] = uint4{1u, 2u, 3u, (uint32_t)i};
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 500, K = argc > 2 ? atoi(argv[2]) : 31;
    const long sites = (long)(4.0e9 / 4 / K / N);
    const size_t bytes = (size_t)sites * K * N * 4;
    uint32_t* d; CHECK(hipMalloc(&d, bytes + 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int chunks = (N + 63) / 64, chunks4 = (N + 255) / 256;
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) { const long waves = sites * chunks; hipLaunchKernelGGL(rows4, dim3((unsigned)((waves + 7) / 8)), dim3(512), 0, 0, d, N, K, chunks, waves); }
            else if (mode == 1) { const long waves = sites * chunks4; hipLaunchKernelGGL(rows16, dim3((unsigned)((waves + 7) / 8)), dim3(512), 0, 0, d, N, K, chunks4, waves); }
            else { const long n = (long)(bytes / 16); hipLaunchKernelGGL(flat16, dim3((unsigned)((n + 511) / 512)), dim3(512), 0, 0, (uint4*)d, n); }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("N %d, K %d rows, %.2f GB: %-7s %.3f ms = %.2f TB/s\n", N, K, bytes / 1e9, mode == 0 ? "rows4" : mode == 1 ? "rows16" : "flat16", best, bytes / best / 1e9);
    }
    return 0;
}
