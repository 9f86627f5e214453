// store_bw.hip -- what rate of pure streaming STORES one MI355X sustains, by access pattern (tools/microbench: not part of the library).
// The fused kernel of the c5 workload writes 124 B per evaluation and nothing else of size: its roofline is the chip's write rate for
// ITS pattern (one site = 31 rows of N = 500 values, a wavefront writes 256 contiguous bytes of a row per instruction).
// build: hipcc -O3 --offload-arch=gfx950 -o store_bw store_bw.hip ; run: ./store_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_dword(uint32_t* out, size_t n_words) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) out[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void fill_x4(uint4* out, size_t n_vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void fill_x4_nt(uint4* out, size_t n_vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
        uint32_t* p = (uint32_t*)(out + i);
        __builtin_nontemporal_store((uint32_t)i, p); __builtin_nontemporal_store(1u, p + 1); __builtin_nontemporal_store(2u, p + 2); __builtin_nontemporal_store(3u, p + 3);
    }
}
// one workgroup of 512 threads per site, N values per row, rows = 1 (dp) + G (gl) + G (pl): thread t writes value t of every row
template <int NT>
__global__ __launch_bounds__(512) void rows_dword(uint32_t* dp, uint32_t* gl, uint32_t* pl, int N, int G) {
    const size_t site = blockIdx.x;
    const int t = threadIdx.x;
    if (t >= N) return;
    uint32_t* a = dp + site * N + t;
    if (NT) __builtin_nontemporal_store((uint32_t)t, a); else *a = (uint32_t)t;
    uint32_t* g = gl + site * (size_t)G * N + t;
    uint32_t* p = pl + site * (size_t)G * N + t;
    for (int k = 0; k < G; ++k) { if (NT) __builtin_nontemporal_store((uint32_t)(t + k), g + (size_t)k * N); else g[(size_t)k * N] = (uint32_t)(t + k); }
    for (int k = 0; k < G; ++k) { if (NT) __builtin_nontemporal_store((uint32_t)(t - k), p + (size_t)k * N); else p[(size_t)k * N] = (uint32_t)(t - k); }
}
// the same bytes with 16-byte stores: a site's (1 + 2G) rows are contiguous per array; thread t writes vector t, t + 512, ... of each array's block
__global__ __launch_bounds__(512) void rows_x4(uint4* dp, uint4* gl, uint4* pl, int N, int G) {
    const size_t site = blockIdx.x;
    const int t = threadIdx.x;
    const int nv_dp = N / 4, nv_g = G * N / 4;                      // (N a multiple of 4)
    if (t < nv_dp) dp[site * nv_dp + t] = make_uint4(t, 1, 2, 3);
    for (int i = t; i < nv_g; i += 512) gl[site * (size_t)nv_g + i] = make_uint4(i, 1, 2, 3);
    for (int i = t; i < nv_g; i += 512) pl[site * (size_t)nv_g + i] = make_uint4(i, 3, 2, 1);
}
// sample-major slab of a site: [N][G] values per array -- thread t writes ITS G consecutive values (60 contiguous bytes per lane)
__global__ __launch_bounds__(512) void slab_dword(uint32_t* dp, uint32_t* gl, uint32_t* pl, int N, int G) {
    const size_t site = blockIdx.x;
    const int t = threadIdx.x;
    if (t >= N) return;
    dp[site * N + t] = (uint32_t)t;
    uint32_t* g = gl + (site * N + t) * (size_t)G;
    uint32_t* p = pl + (site * N + t) * (size_t)G;
    for (int k = 0; k < G; ++k) g[k] = (uint32_t)(t + k);
    for (int k = 0; k < G; ++k) p[k] = (uint32_t)(t - k);
}

template <class F> static double time_ms(F&& launch, int reps) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipGetLastError());
    return ms / reps;
}

int main() {
    const int N = 500, G = 15, S = 65536, reps = 10;
    const size_t n_dp = (size_t)S * N, n_g = (size_t)S * G * N;
    const size_t bytes = 4 * (n_dp + 2 * n_g);
    uint32_t *dp, *gl, *pl;
    CHECK(hipMalloc((void**)&dp, 4 * n_dp)); CHECK(hipMalloc((void**)&gl, 4 * n_g)); CHECK(hipMalloc((void**)&pl, 4 * n_g));
    printf("{\"bytes_per_launch\": %zu", bytes);
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(fill_dword, dim3(256 * 32), dim3(256), 0, 0, gl, n_g); }, reps);
    printf(", \"fill_dword_TBps\": %.3f", 4.0 * n_g / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(fill_x4, dim3(256 * 32), dim3(256), 0, 0, (uint4*)gl, n_g / 4); }, reps);
    printf(", \"fill_x4_TBps\": %.3f", 4.0 * n_g / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(fill_x4_nt, dim3(256 * 32), dim3(256), 0, 0, (uint4*)gl, n_g / 4); }, reps);
    printf(", \"fill_x4_nt_TBps\": %.3f", 4.0 * n_g / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(rows_dword<0>, dim3(S), dim3(512), 0, 0, dp, gl, pl, N, G); }, reps);
    printf(", \"c5_rows_dword_TBps\": %.3f", bytes / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(rows_dword<1>, dim3(S), dim3(512), 0, 0, dp, gl, pl, N, G); }, reps);
    printf(", \"c5_rows_dword_nt_TBps\": %.3f", bytes / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(rows_x4, dim3(S), dim3(512), 0, 0, (uint4*)dp, (uint4*)gl, (uint4*)pl, N, G); }, reps);
    printf(", \"c5_rows_x4_TBps\": %.3f", bytes / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL(slab_dword, dim3(S), dim3(512), 0, 0, dp, gl, pl, N, G); }, reps);
    printf(", \"c5_slab_dword_TBps\": %.3f", bytes / ms * 1e-9);
    // a float4 copy for reference (read + write)
    printf("}\n");
    return 0;
}
