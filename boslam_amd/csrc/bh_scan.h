// bh_scan.h - exclusive prefix sums on the device (the set-up paths of ba.hip and bow.hip: slot numbering, list starts).
//
// Three small kernels instead of a library scan: a 1024-thread workgroup scans 4096 consecutive elements (four per thread, wave scan by
// DPP-free __shfl_up, the 16 wave totals through LDS) and writes its total to `tops`; the totals are scanned the same way (recursively:
// three levels reach 2^36 elements) and added back.  out[i] = in[0] + ... + in[i - 1]; callers that want the grand total pass n + 1
// elements with in[n] = 0.  `in` may equal `out`.  Everything is enqueued on `st`; `scratch` grows to ceil(n / 4096) + ceil(n / 4096^2) + 2
// elements of T.
#pragma once
#include "bh_internal.h"

#define BH_SCAN_T 1024
#define BH_SCAN_PER 4
#define BH_SCAN_BLK (BH_SCAN_T * BH_SCAN_PER)

template <typename TI, typename TO>
static __global__ __launch_bounds__(BH_SCAN_T) void k_bh_scan_blocks(const TI* __restrict__ in, TO* __restrict__ out, long long n, TO* __restrict__ tops)
{
    __shared__ TO s_w[BH_SCAN_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long i0 = (long long)blockIdx.x * BH_SCAN_BLK + (long long)tid * BH_SCAN_PER;
    TO v[BH_SCAN_PER], sum = 0;
#pragma unroll
    for (int k = 0; k < BH_SCAN_PER; k++) { v[k] = i0 + k < n ? (TO)in[i0 + k] : (TO)0; sum += v[k]; }
    TO inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const TO o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    TO base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BH_SCAN_T / 64; w++) { const TO t = s_w[w]; if (w < wv) base += t; tot += t; }
    TO run = base + inc - sum;
#pragma unroll
    for (int k = 0; k < BH_SCAN_PER; k++) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
    if (tid == 0 && tops) tops[blockIdx.x] = tot;
}
template <typename TO>
static __global__ __launch_bounds__(256) void k_bh_scan_add(TO* __restrict__ out, long long n, const TO* __restrict__ tops)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += tops[i / BH_SCAN_BLK];
}

template <typename TO>
static inline size_t bh_exscan_scratch(long long n)
{
    size_t tot = 2;
    while (n > BH_SCAN_BLK) { n = (n + BH_SCAN_BLK - 1) / BH_SCAN_BLK; tot += (size_t)n; }
    return tot * sizeof(TO);
}
// scratch: at least bh_exscan_scratch<TO>(n) bytes
template <typename TI, typename TO>
static inline int bh_exscan(bh_ctx* ctx, hipStream_t st, const TI* in, TO* out, long long n, void* scratch)
{
    if (n <= 0) return BH_OK;
    const long long nb = (n + BH_SCAN_BLK - 1) / BH_SCAN_BLK;
    TO* tops = (TO*)scratch;
    hipLaunchKernelGGL((k_bh_scan_blocks<TI, TO>), dim3((unsigned)nb), dim3(BH_SCAN_T), 0, st, in, out, n, nb > 1 ? tops : (TO*)nullptr);
    if (nb > 1) {
        const int rc = bh_exscan<TO, TO>(ctx, st, tops, tops, nb, tops + nb);
        if (rc != BH_OK) return rc;
        hipLaunchKernelGGL((k_bh_scan_add<TO>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, (const TO*)tops);
    }
    BH_HIP(ctx, hipGetLastError());
    return BH_OK;
}
