// vgl_pack.hip -- a tile's kept sites as variable-length records, packed on the device (ABI 6: vgl_pack_plan_device,
// vgl_pack_records_device).  What it replaces: the reference hands one record at a time to bcf_write (vcfgl.cpp:167-206) from the
// arrays simRecord::add_tags() filled (bcf_utils.cpp:426-507: per record only nGenotypes / nAlleles values per sample).  With the
// sites sharded over GPUs each rank hands the writer its kept sites in that form -- skipped sites dropped, of every FORMAT tag only the
// nG(site) / nA(site) valid planes -- and this is the gather's producer side: an exclusive prefix sum over the tile's sites, then
// coalesced row copies (HBM-bound: every byte is read once and written once).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcfgl_hip.h"

namespace {

__device__ __forceinline__ int rows_of(const int kind, const int status, const int na) {
    if (status < 0) return 0;                                   // a skipped site leaves no record
    if (kind == VGL_PACK_ROW) return 1;
    return kind == VGL_PACK_ROWS_G ? na * (na + 1) / 2 : na;    // lut_nAlleles_to_nGenotypes (shared.h) / nAlleles
}

// exclusive prefix sums of (kept, genotype rows, allele rows) over the sites: one workgroup walks the tile in chunks of 1024 sites
// (a 65536-site tile: 64 chunks, ~20 us -- the copies behind it move gigabytes)
__global__ __launch_bounds__(1024) void k_pack_scan(const int n, const int32_t* __restrict__ status, const int32_t* __restrict__ n_alleles, int32_t* __restrict__ off) {
    __shared__ int32_t s_w[3][16];
    __shared__ int32_t s_carry[3];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 3) s_carry[t] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + t;
        int v[3] = {0, 0, 0};
        if (i < n) {
            const int st = status[i], na = n_alleles[i];
            v[0] = rows_of(VGL_PACK_ROW, st, na); v[1] = rows_of(VGL_PACK_ROWS_G, st, na); v[2] = rows_of(VGL_PACK_ROWS_A, st, na);
        }
        int inc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int x = v[c];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
            inc[c] = x;
            if (lane == 63) s_w[c][w] = x;
        }
        __syncthreads();
        if (t < 3) {                                            // the sixteen wavefront totals -> their exclusive prefixes (+ the carry of earlier chunks)
            int run = s_carry[t];
            for (int k = 0; k < 16; ++k) { const int x = s_w[t][k]; s_w[t][k] = run; run += x; }
            s_carry[t] = run;
        }
        __syncthreads();
        if (i < n) {
#pragma unroll
            for (int c = 0; c < 3; ++c) off[(size_t)c * (n + 1) + i] = s_w[c][w] + inc[c] - v[c];
        }
        __syncthreads();
    }
    if (t < 3) off[(size_t)t * (n + 1) + n] = s_carry[t];
}

// index rows of the kept sites: (site index in the tile, site_status, n_alleles)
__global__ __launch_bounds__(256) void k_pack_index(const int n, const int32_t* __restrict__ status, const int32_t* __restrict__ n_alleles, const int32_t* __restrict__ off,
                                                    int32_t* __restrict__ index_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || status[i] < 0) return;
    int32_t* o = index_out + (size_t)off[i] * 3;
    o[0] = i; o[1] = status[i]; o[2] = n_alleles[i];
}

// rows of a plane array: source row r = site * K + k goes to packed row off[site] + k when k < rows(site).  One wavefront per source row, four rows
// per workgroup; VEC = bytes per lane and step (16 when rows and bases are 16-byte aligned, else 4, else 1): a wavefront reads and writes whole
// contiguous kilobytes
template <int VEC>
__global__ __launch_bounds__(256) void k_pack_rows(const int n, const int K, const int kind, const int64_t row_bytes, const int32_t* __restrict__ status,
                                                   const int32_t* __restrict__ n_alleles, const int32_t* __restrict__ off, const uint8_t* __restrict__ src,
                                                   uint8_t* __restrict__ dst) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (int64_t)n * K) return;
    const int site = (int)(r / K), k = (int)(r - (int64_t)site * K);
    if (k >= rows_of(kind, status[site], n_alleles[site])) return;                       // (wave-uniform)
    const uint8_t* s = src + r * row_bytes;
    uint8_t* d = dst + ((int64_t)off[site] + k) * row_bytes;
    const int lane = threadIdx.x & 63;
    if (VEC == 16) { for (int64_t b = (int64_t)lane * 16; b < row_bytes; b += 1024) *(uint4*)(d + b) = *(const uint4*)(s + b); }
    else if (VEC == 4) { for (int64_t b = (int64_t)lane * 4; b < row_bytes; b += 256) *(uint32_t*)(d + b) = *(const uint32_t*)(s + b); }
    else { for (int64_t b = lane; b < row_bytes; b += 64) d[b] = s[b]; }
}

// short rows (per-site vectors of a few bytes): one lane per source row
__global__ __launch_bounds__(256) void k_pack_small(const int n, const int K, const int kind, const int row_bytes, const int32_t* __restrict__ status,
                                                    const int32_t* __restrict__ n_alleles, const int32_t* __restrict__ off, const uint8_t* __restrict__ src,
                                                    uint8_t* __restrict__ dst) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= (int64_t)n * K) return;
    const int site = (int)(r / K), k = (int)(r - (int64_t)site * K);
    if (k >= rows_of(kind, status[site], n_alleles[site])) return;
    const uint8_t* s = src + r * row_bytes;
    uint8_t* d = dst + ((int64_t)off[site] + k) * row_bytes;
    if ((row_bytes & 3) == 0 && (((uintptr_t)s | (uintptr_t)d) & 3) == 0) { for (int b = 0; b < row_bytes; b += 4) *(uint32_t*)(d + b) = *(const uint32_t*)(s + b); }
    else for (int b = 0; b < row_bytes; ++b) d[b] = s[b];
}

}  // namespace

extern "C" int vgl_pack_set_error(int code, const char* msg);       // vgl_host.cpp: records the message for vgl_last_error()

extern "C" int vgl_pack_plan_device(int32_t device, int32_t n_sites, const int32_t* site_status, const int32_t* n_alleles, int32_t* offsets,
                                    vgl_pack_plan* totals, void* hip_stream) {
    if (n_sites < 0 || !totals || (n_sites > 0 && (!site_status || !n_alleles || !offsets))) return vgl_pack_set_error(VGL_E_ARG, "vgl_pack_plan_device: null argument");
    totals->n_kept = totals->rows_g = totals->rows_a = 0;
    if (n_sites == 0) return VGL_OK;
    if (hipSetDevice(device) != hipSuccess) return vgl_pack_set_error(VGL_E_NODEVICE, "vgl_pack_plan_device: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, (int)n_sites, site_status, n_alleles, offsets);
    int32_t h[3];
    for (int c = 0; c < 3; ++c)
        if (hipMemcpyAsync(&h[c], offsets + (size_t)c * ((size_t)n_sites + 1) + (size_t)n_sites, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
            return vgl_pack_set_error(VGL_E_NODEVICE, "vgl_pack_plan_device: copy of the totals failed");
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return vgl_pack_set_error(VGL_E_NODEVICE, "vgl_pack_plan_device: k_pack_scan failed");
    totals->n_kept = h[0]; totals->rows_g = h[1]; totals->rows_a = h[2];
    return VGL_OK;
}

extern "C" int vgl_pack_records_device(int32_t device, int32_t n_sites, const int32_t* site_status, const int32_t* n_alleles, const int32_t* offsets,
                                       int32_t* index_out, const vgl_pack_field* fields, int32_t n_fields, void* hip_stream) {
    if (n_sites < 0 || n_fields < 0 || (n_fields > 0 && !fields)) return vgl_pack_set_error(VGL_E_ARG, "vgl_pack_records_device: bad argument");
    if (n_sites == 0) return VGL_OK;
    if (!site_status || !n_alleles || !offsets) return vgl_pack_set_error(VGL_E_ARG, "vgl_pack_records_device: null argument");
    if (hipSetDevice(device) != hipSuccess) return vgl_pack_set_error(VGL_E_NODEVICE, "vgl_pack_records_device: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t stride = (size_t)n_sites + 1;
    if (index_out) hipLaunchKernelGGL(k_pack_index, dim3((unsigned)((n_sites + 255) / 256)), dim3(256), 0, st, (int)n_sites, site_status, n_alleles, offsets, index_out);
    for (int f = 0; f < n_fields; ++f) {
        const vgl_pack_field& F = fields[f];
        if (!F.src || !F.dst || F.row_bytes <= 0 || F.planes < 1 || F.kind < VGL_PACK_ROW || F.kind > VGL_PACK_ROWS_A || (F.kind == VGL_PACK_ROW && F.planes != 1))
            return vgl_pack_set_error(VGL_E_ARG, "vgl_pack_records_device: bad field descriptor");
        const int32_t* off = offsets + (size_t)F.kind * stride;                          // (VGL_PACK_ROW = 0, _ROWS_G = 1, _ROWS_A = 2: the scan's three rows)
        const int64_t rows = (int64_t)n_sites * F.planes;
        const uint8_t* s = (const uint8_t*)F.src;
        uint8_t* d = (uint8_t*)F.dst;
        if (F.row_bytes <= 64) {
            if (rows > 0x7FFFFFFFLL * 256) return vgl_pack_set_error(VGL_E_ARG, "vgl_pack_records_device: too many rows");
            hipLaunchKernelGGL(k_pack_small, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (int)n_sites, (int)F.planes, (int)F.kind, (int)F.row_bytes, site_status, n_alleles, off, s, d);
        } else {
            const dim3 g((unsigned)((rows + 3) / 4));
            const bool a16 = (F.row_bytes % 16 == 0) && ((((uintptr_t)s) | ((uintptr_t)d)) % 16 == 0);
            const bool a4 = (F.row_bytes % 4 == 0) && ((((uintptr_t)s) | ((uintptr_t)d)) % 4 == 0);
            if (a16) hipLaunchKernelGGL(k_pack_rows<16>, g, dim3(256), 0, st, (int)n_sites, (int)F.planes, (int)F.kind, (int64_t)F.row_bytes, site_status, n_alleles, off, s, d);
            else if (a4) hipLaunchKernelGGL(k_pack_rows<4>, g, dim3(256), 0, st, (int)n_sites, (int)F.planes, (int)F.kind, (int64_t)F.row_bytes, site_status, n_alleles, off, s, d);
            else hipLaunchKernelGGL(k_pack_rows<1>, g, dim3(256), 0, st, (int)n_sites, (int)F.planes, (int)F.kind, (int64_t)F.row_bytes, site_status, n_alleles, off, s, d);
        }
    }
    if (hipGetLastError() != hipSuccess) return vgl_pack_set_error(VGL_E_NODEVICE, "vgl_pack_records_device: a launch failed");
    return VGL_OK;
}
