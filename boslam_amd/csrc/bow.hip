// bow.hip — (1) local-map visibility gate, (2) DBoW3 L1 database query, for gfx950.
//
// (1) bh_visibility_gate replaces the per-point Python loop of slam/tracking.py:97-111
//     (pixel = cam.cam_map(SE3Quat(frame.R, frame.t) * mp.pt3d); in-image test; dot(see_vector, n) < cos60):
//     one workgroup per frame, f64, ordered compaction of the surviving indices (ballot + prefix sums).
// (2) bh_bow_* replaces Database::query(des, [], max_results, -1) — slam/bow_db.py:37-42 ->
//     docker/pydbow3/src/dbow3.cpp:147-152 — the L1 score of one query BoW vector against every database entry.
//     HBM-streaming design: the database is a padded SoA CSR (u32 word ids / f64 weights, rows padded to 4 entries so
//     that every lane issues aligned 16-B loads); a persistent grid of 1024-lane workgroups keeps a bitmap of the query's
//     words in LDS (vocabulary 10^6 words -> 125 KB) and walks whole entries one wave at a time; a bitmap hit is confirmed
//     by binary search in the query (also in LDS), and hits are accumulated in ascending word order (lane order via
//     ballot, then in-lane order), i.e. the exact summation order of DBoW3's queryL1 — scores are bit-identical to the
//     CPU restatement.  Only the word-id stream is read for non-matching entries (weights are touched on hits only).
#include "bh_internal.h"
#include <math.h>
#include <vector>

// ================================================================================================ visibility gate
__global__ __launch_bounds__(1024) void k_gate(const double* __restrict__ pts, const double* __restrict__ nrm,
                                               const int* __restrict__ M_arr, int stride, const double* __restrict__ Rm,
                                               const double* __restrict__ tv, const double* __restrict__ seev,
                                               double f, double cx, double cy, int width, int height, double cos_thr,
                                               int* __restrict__ idx, int* __restrict__ n_out)
{
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t b = blockIdx.x;
    const int M = M_arr[b];
    const double* R = Rm + 9 * b; const double* t = tv + 3 * b; const double* see = seev + 3 * b;
    const double* P = pts + b * (size_t)stride * 3; const double* N = nrm + b * (size_t)stride * 3;
    int* out = idx + b * (size_t)stride;
    int base = 0;
    for (int c0 = 0; c0 < M; c0 += 1024) {
        const int i = c0 + tid;
        bool keep = false;
        if (i < M) {
            const double px = P[3 * i], py = P[3 * i + 1], pz = P[3 * i + 2];
            const double qx = R[0] * px + R[1] * py + R[2] * pz + t[0];
            const double qy = R[3] * px + R[4] * py + R[5] * pz + t[1];
            const double qz = R[6] * px + R[7] * py + R[8] * pz + t[2];
            const double u = qx / qz * f + cx, v = qy / qz * f + cy;
            if (0 <= u && u < width && 0 <= v && v < height) {
                const double d = see[0] * N[3 * i] + see[1] * N[3 * i + 1] + see[2] * N[3 * i + 2];
                keep = d < cos_thr;
            }
        }
        const unsigned long long m = __ballot(keep);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) s_w[wv] = __popcll(m);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < 16; w++) { const int c = s_w[w]; if (w < wv) wbase += c; tot += c; }
        if (keep) out[base + wbase + pre] = i;
        base += tot;
    }
    if (tid == 0) n_out[b] = base;
}

extern "C" int bh_visibility_gate(bh_ctx* ctx, const double* pts, const double* normals, const int* M, int B, int stride,
                                  const double* R, const double* t, const double* see, double f, double cx, double cy,
                                  int width, int height, double cos_thr, int* idx, int* n_out, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!pts || !normals || !M || !R || !t || !see || !idx || !n_out || B < 0 || stride < 1 ||
        (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_visibility_gate: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t np = (size_t)B * stride;
    const double *d_p = pts, *d_n = normals, *d_R = R, *d_t = t, *d_s = see; const int* d_M = M; int *d_i = idx, *d_o = n_out;
    if (mem == BH_MEM_HOST) {
        for (int b = 0; b < B; b++) if (M[b] < 0 || M[b] > stride) return bh_set_err(ctx, BH_ERR_INVALID, "bh_visibility_gate: M[b] outside [0, stride]");
        BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(double) * 3 * np)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(double) * 3 * np));
        BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(double) * 15 * B)); BH_TRY(bh_reserve(ctx, &ctx->st_d, sizeof(int) * (np + 2 * (size_t)B)));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, pts, sizeof(double) * 3 * np, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, normals, sizeof(double) * 3 * np, hipMemcpyHostToDevice, st));
        double* c = (double*)ctx->st_c.p;
        BH_HIP(ctx, hipMemcpyAsync(c, R, sizeof(double) * 9 * B, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(c + 9 * (size_t)B, t, sizeof(double) * 3 * B, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(c + 12 * (size_t)B, see, sizeof(double) * 3 * B, hipMemcpyHostToDevice, st));
        int* di = (int*)ctx->st_d.p;
        BH_HIP(ctx, hipMemcpyAsync(di + np + B, M, sizeof(int) * B, hipMemcpyHostToDevice, st));
        d_p = (const double*)ctx->st_a.p; d_n = (const double*)ctx->st_b.p; d_R = c; d_t = c + 9 * (size_t)B; d_s = c + 12 * (size_t)B;
        d_i = di; d_o = di + np; d_M = di + np + B;
    }
    BH_LAUNCH(ctx, BH_K_GATE, k_gate, dim3(B), dim3(1024), st, d_p, d_n, d_M, stride, d_R, d_t, d_s, f, cx, cy, width, height, cos_thr, d_i, d_o);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(idx, d_i, sizeof(int) * np, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_out, d_o, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}

// ================================================================================================ BoW L1 query
#define BOW_T        1024
#define BOW_BITS     (1u << 20)                    // LDS bitmap: 2^20 bits = 128 KiB (vocabulary 10^6 words: exact)
#define BOW_QMAX     1536                          // query words kept in LDS beside the bitmap (12 B each)
#define BOW_PAD_WORD 0xffffffffu

struct bh_bow {
    bh_ctx* ctx = nullptr;
    long long n_entries = 0, n_elems = 0;          // padded element count
    std::vector<long long> h_pstart;               // host mirror of the padded row offsets (for appends)
    bh_buf pstart, words, vals, acc, qw, qv, res;
};

// grow a device buffer keeping its first `keep` bytes
static int bow_grow(bh_ctx* ctx, bh_buf* b, size_t need, size_t keep)
{
    if (need <= b->cap) return BH_OK;
    size_t want = b->cap ? b->cap : 4096;
    while (want < need) want *= 2;
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) return bh_set_err(ctx, BH_ERR_NOMEM, "hipMalloc(%zu) failed", want);
    if (b->p) {
        if (keep && hipMemcpyAsync(p, b->p, keep, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { (void)hipFree(p); return bh_set_err(ctx, BH_ERR_HIP, "device copy failed"); }
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b->p);
    }
    b->p = p; b->cap = want;
    return BH_OK;
}

__global__ __launch_bounds__(256) void k_bow_pack(const long long* __restrict__ start, const uint32_t* __restrict__ w,
                                                  const double* __restrict__ v, const long long* __restrict__ pstart,
                                                  long long n_entries, uint32_t* __restrict__ pw, double* __restrict__ pv)
{
    const long long e = blockIdx.x;
    if (e >= n_entries) return;
    const long long a = start[e], n = start[e + 1] - a, pa = pstart[e], pn = pstart[e + 1] - pa;
    for (long long i = threadIdx.x; i < pn; i += 256) {
        pw[pa + i] = i < n ? w[a + i] : BOW_PAD_WORD;
        pv[pa + i] = i < n ? v[a + i] : 0.0;
    }
}

__device__ __forceinline__ double bow_shfl(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(BOW_T) void k_bow_scan(const long long* __restrict__ pstart, const uint32_t* __restrict__ pw,
                                                    const double* __restrict__ pv, long long n_entries,
                                                    const uint32_t* __restrict__ qw, const double* __restrict__ qv, int nq,
                                                    double* __restrict__ acc_out)
{
    extern __shared__ uint32_t smem[];
    uint32_t* s_bits = smem;                                   // BOW_BITS / 32 words
    uint32_t* s_qw = smem + BOW_BITS / 32;                     // BOW_QMAX
    double*   s_qv = reinterpret_cast<double*>(s_qw + BOW_QMAX);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < (int)(BOW_BITS / 32); i += BOW_T) s_bits[i] = 0;
    __syncthreads();
    for (int i = tid; i < nq; i += BOW_T) {
        const uint32_t w = qw[i];
        atomicOr(&s_bits[(w & (BOW_BITS - 1)) >> 5], 1u << (w & 31));
        s_qw[i] = w; s_qv[i] = qv[i];
    }
    __syncthreads();
    const long long wstride = (long long)gridDim.x * (BOW_T / 64);
    for (long long e = (long long)blockIdx.x * (BOW_T / 64) + wv; e < n_entries; e += wstride) {
        const long long a = pstart[e], b = pstart[e + 1];
        double acc = 0.0; bool any = false;
        for (long long c1 = a; c1 < b; c1 += 1024) {              // four 256-word chunks' loads in flight (an entry of configs[3] is exactly one round)
            uint4 wq[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const long long i = c1 + 256 * u + 4 * lane;
                wq[u] = make_uint4(BOW_PAD_WORD, BOW_PAD_WORD, BOW_PAD_WORD, BOW_PAD_WORD);
                if (i < b) wq[u] = *reinterpret_cast<const uint4*>(pw + i);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
            const long long i = c1 + 256 * u + 4 * lane;
            const uint4 w4 = wq[u];
            const uint32_t ws[4] = { w4.x, w4.y, w4.z, w4.w };
            double val[4] = { 0, 0, 0, 0 };
            int hitm = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t w = ws[k];
                if (w != BOW_PAD_WORD && (s_bits[(w & (BOW_BITS - 1)) >> 5] >> (w & 31)) & 1u) {
                    int lo = 0, hi = nq;                       // confirm + locate in the sorted query
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_qw[mid] < w) lo = mid + 1; else hi = mid; }
                    if (lo < nq && s_qw[lo] == w) {
                        const double q = s_qv[lo], d = pv[i + k];
                        val[k] = fabs(q - d) - fabs(q) - fabs(d);
                        hitm |= 1 << k;
                    }
                }
            }
            unsigned long long m = __ballot(hitm != 0);
            while (m) {                                        // ascending word order: lanes in order, then in-lane order
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int hm = __shfl(hitm, src, 64);
#pragma unroll
                for (int k = 0; k < 4; k++) { const double v = bow_shfl(val[k], src); if (hm & (1 << k)) acc += v; }
                any = true;
            }
        }
        }
        if (lane == 0) acc_out[e] = any ? acc : INFINITY;        // +inf: no common word, entry absent from the results
    }
}

// Stage 1 of the top-k for large databases: workgroup b finds the k best of its slice of TOPK_SLICE accumulators (8 per thread, in
// registers) and writes them as candidates (value, entry id; value = +inf when the slice runs out).  Ties -> smaller id throughout.
#define TOPK_SLICE 8192
#define TOPK_MAXK 64
__global__ __launch_bounds__(1024) void k_bow_topk_slices(const double* __restrict__ acc, long long n_entries, int k,
                                                          double* __restrict__ cand_v, long long* __restrict__ cand_i)
{
    __shared__ double s_v[16];
    __shared__ long long s_i[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long base = (long long)blockIdx.x * TOPK_SLICE;
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { const long long e = base + tid + 1024 * j; v[j] = e < n_entries ? acc[e] : INFINITY; }
    for (int r = 0; r < k; r++) {
        double bv = INFINITY; long long bi = -1;
#pragma unroll
        for (int j = 0; j < 8; j++) { const long long e = base + tid + 1024 * j; if (v[j] < bv) { bv = v[j]; bi = e; } }    // ascending e: first minimum
        for (int m = 32; m > 0; m >>= 1) {
            int lo = __double2loint(bv), hi = __double2hiint(bv);
            const double ov = __hiloint2double(__shfl_xor(hi, m, 64), __shfl_xor(lo, m, 64));
            const long long oi = ((long long)__shfl_xor((int)(bi >> 32), m, 64) << 32) | (unsigned)__shfl_xor((int)(bi & 0xffffffff), m, 64);
            if (oi >= 0 && (ov < bv || (ov == bv && (bi < 0 || oi < bi)))) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { s_v[wv] = bv; s_i[wv] = bi; }
        __syncthreads();
        bv = INFINITY; bi = -1;
        for (int w = 0; w < 16; w++) { const double ov = s_v[w]; const long long oi = s_i[w]; if (oi >= 0 && (ov < bv || (ov == bv && (bi < 0 || oi < bi)))) { bv = ov; bi = oi; } }
        if (tid == 0) { cand_v[(size_t)blockIdx.x * k + r] = bi >= 0 ? bv : INFINITY; cand_i[(size_t)blockIdx.x * k + r] = bi; }
#pragma unroll
        for (int j = 0; j < 8; j++) if (base + tid + 1024 * j == bi) v[j] = INFINITY;          // taken
    }
}

// single workgroup: k rounds of argmin (ties -> smaller id), results best first.  cand_i == NULL: over the accumulators themselves
// (entry id = index); else over stage-1 candidates (entry id = cand_i[index], -1 = empty).
__global__ __launch_bounds__(1024) void k_bow_topk(double* __restrict__ acc, long long n_entries, int k, const long long* __restrict__ cand_i,
                                                   int* __restrict__ ids, double* __restrict__ scores, int* __restrict__ n_res)
{
    __shared__ double s_v[16];
    __shared__ long long s_i[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int found = 0;
    for (int r = 0; r < k; r++) {
        double bv = INFINITY; long long bi = -1;
        for (long long e = tid; e < n_entries; e += 1024) {
            const double v = acc[e];
            const long long id = cand_i ? cand_i[e] : e;
            if (id >= 0 && (v < bv || (v == bv && bi >= 0 && id < bi))) { bv = v; bi = id; }
        }
        for (int m = 32; m > 0; m >>= 1) {
            int lo = __double2loint(bv), hi = __double2hiint(bv);
            const double ov = __hiloint2double(__shfl_xor(hi, m, 64), __shfl_xor(lo, m, 64));
            const long long oi = ((long long)__shfl_xor((int)(bi >> 32), m, 64) << 32) | (unsigned)__shfl_xor((int)(bi & 0xffffffff), m, 64);
            if (oi >= 0 && (ov < bv || (ov == bv && (bi < 0 || oi < bi)))) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { s_v[wv] = bv; s_i[wv] = bi; }
        __syncthreads();
        bv = INFINITY; bi = -1;
        for (int w = 0; w < 16; w++) { const double ov = s_v[w]; const long long oi = s_i[w]; if (oi >= 0 && (ov < bv || (ov == bv && (bi < 0 || oi < bi)))) { bv = ov; bi = oi; } }
        if (bi < 0) break;
        if (cand_i) { for (long long e = tid; e < n_entries; e += 1024) if (cand_i[e] == bi) acc[e] = INFINITY; }      // taken
        if (tid == 0) { ids[r] = (int)bi; scores[r] = -bv / 2.0; if (!cand_i) acc[bi] = INFINITY; }
        found = r + 1;
        __syncthreads();
    }
    if (tid == 0) *n_res = found;
}

extern "C" int bh_bow_create(bh_ctx* ctx, bh_bow** out)
{
    if (!ctx || !out) return BH_ERR_INVALID;
    bh_bow* b = new bh_bow(); b->ctx = ctx; *out = b;
    return BH_OK;
}

extern "C" void bh_bow_destroy(bh_bow* bow)
{
    if (!bow) return;
    (void)hipSetDevice(bow->ctx->device);
    (void)hipStreamSynchronize(bow->ctx->stream);
    bh_buf* all[] = { &bow->pstart, &bow->words, &bow->vals, &bow->acc, &bow->qw, &bow->qv, &bow->res };
    for (bh_buf* b : all) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    delete bow;
}

extern "C" int bh_bow_set_db(bh_bow* bow, long long n_entries, const int64_t* start, const uint32_t* words,
                             const double* vals, int mem)
{
    if (!bow) return BH_ERR_INVALID;
    bh_ctx* ctx = bow->ctx;
    if (n_entries < 0 || (n_entries && (!start || !words || !vals)) || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_set_db: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<long long> hstart(n_entries + 1, 0);
    if (n_entries) {
        if (mem == BH_MEM_HOST) for (long long e = 0; e <= n_entries; e++) hstart[e] = start[e];
        else { BH_HIP(ctx, hipMemcpyAsync(hstart.data(), start, sizeof(long long) * (n_entries + 1), hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipStreamSynchronize(st)); }
    }
    std::vector<long long> pst(n_entries + 1, 0);
    for (long long e = 0; e < n_entries; e++) {
        const long long n = hstart[e + 1] - hstart[e];
        if (n < 0) return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_set_db: start[] not ascending");
        pst[e + 1] = pst[e] + ((n + 3) & ~3ll);
    }
    const long long nel = hstart[n_entries] - hstart[0], npad = pst[n_entries];
    bow->n_entries = n_entries; bow->n_elems = npad; bow->h_pstart = pst;
    BH_TRY(bh_reserve(ctx, &bow->pstart, sizeof(long long) * (n_entries + 1)));
    BH_TRY(bh_reserve(ctx, &bow->words, sizeof(uint32_t) * (npad + 4))); BH_TRY(bh_reserve(ctx, &bow->vals, sizeof(double) * (npad + 4)));
    BH_TRY(bh_reserve(ctx, &bow->acc, sizeof(double) * (n_entries + 1)));
    BH_HIP(ctx, hipMemcpyAsync(bow->pstart.p, pst.data(), sizeof(long long) * (n_entries + 1), hipMemcpyHostToDevice, st));
    if (n_entries == 0) { BH_HIP(ctx, hipStreamSynchronize(st)); return BH_OK; }
    const long long* d_start; const uint32_t* d_w = words; const double* d_v = vals;
    BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(long long) * (n_entries + 1)));
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, hstart.data(), sizeof(long long) * (n_entries + 1), hipMemcpyHostToDevice, st));
    d_start = (const long long*)ctx->st_a.p;
    if (mem == BH_MEM_HOST) {
        BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(uint32_t) * (nel + 1))); BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(double) * (nel + 1)));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, words + hstart[0], sizeof(uint32_t) * nel, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_c.p, vals + hstart[0], sizeof(double) * nel, hipMemcpyHostToDevice, st));
        d_w = (const uint32_t*)ctx->st_b.p - hstart[0]; d_v = (const double*)ctx->st_c.p - hstart[0];
    }
    hipLaunchKernelGGL(k_bow_pack, dim3((unsigned)n_entries), dim3(256), 0, st, d_start, d_w, d_v, (const long long*)bow->pstart.p,
                       n_entries, (uint32_t*)bow->words.p, (double*)bow->vals.p);
    BH_HIP(ctx, hipGetLastError());
    BH_HIP(ctx, hipStreamSynchronize(st));
    return BH_OK;
}

// ---- query restricted to a subset of entries (the fork's Database::query(features, target_inds, ...), slam/bow_db.py:37-39):
// accumulators of entries outside target_inds are set to +inf (= absent from the results) between the scan and the top-k.
__global__ void k_bow_flag_set(const int* __restrict__ subset, int n_subset, long long n_entries, uint8_t* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_subset) { const int e = subset[i]; if (e >= 0 && e < n_entries) flag[e] = 1; }
}
__global__ void k_bow_flag_apply(const uint8_t* __restrict__ flag, long long n_entries, double* __restrict__ acc)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < n_entries && !flag[e]) acc[e] = INFINITY;
}

static int bow_query_impl(bh_bow* bow, const uint32_t* q_words, const double* q_vals, int nq, const int* subset, int n_subset,
                          int max_results, int* ids, double* scores, int* n_results, int mem);


extern "C" int bh_bow_query_l1(bh_bow* bow, const uint32_t* q_words, const double* q_vals, int nq, int max_results,
                               int* ids, double* scores, int* n_results, int mem)
{
    if (!bow) return BH_ERR_INVALID;
    return bow_query_impl(bow, q_words, q_vals, nq, nullptr, 0, max_results, ids, scores, n_results, mem);
}

static int bow_query_impl(bh_bow* bow, const uint32_t* q_words, const double* q_vals, int nq, const int* subset, int n_subset,
                          int max_results, int* ids, double* scores, int* n_results, int mem)
{
    bh_ctx* ctx = bow->ctx;
    if (!q_words || !q_vals || !ids || !scores || !n_results || nq < 0 || max_results < 1 || max_results > 1024 ||
        (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_query_l1: bad argument");
    if (nq > BOW_QMAX) return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_query_l1: more than %d query words", BOW_QMAX);
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (bow->n_entries == 0 || nq == 0) {
        if (mem == BH_MEM_HOST) *n_results = 0; else BH_HIP(ctx, hipMemsetAsync(n_results, 0, sizeof(int), st));
        return BH_OK;
    }
    const uint32_t* d_qw = q_words; const double* d_qv = q_vals; int* d_ids = ids; double* d_sc = scores; int* d_n = n_results;
    if (mem == BH_MEM_HOST) {
        for (int i = 1; i < nq; i++) if (q_words[i] <= q_words[i - 1]) return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_query_l1: query words must be strictly ascending");
        BH_TRY(bh_reserve(ctx, &bow->qw, sizeof(uint32_t) * (nq + 1))); BH_TRY(bh_reserve(ctx, &bow->qv, sizeof(double) * (nq + 1)));
        BH_TRY(bh_reserve(ctx, &bow->res, (sizeof(double) + sizeof(int)) * max_results + 64));
        BH_HIP(ctx, hipMemcpyAsync(bow->qw.p, q_words, sizeof(uint32_t) * nq, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(bow->qv.p, q_vals, sizeof(double) * nq, hipMemcpyHostToDevice, st));
        d_qw = (const uint32_t*)bow->qw.p; d_qv = (const double*)bow->qv.p;
        d_sc = (double*)bow->res.p; d_ids = (int*)(d_sc + max_results); d_n = d_ids + max_results;
    }
    const size_t lds = BOW_BITS / 8 + BOW_QMAX * (sizeof(uint32_t) + sizeof(double));
    static bool attr_set = false;
    if (!attr_set) { BH_HIP(ctx, hipFuncSetAttribute((const void*)k_bow_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
    int grid = 256;                                            // one persistent workgroup per CU (LDS-limited to 1/CU anyway)
    const long long need = (bow->n_entries + BOW_T / 64 - 1) / (BOW_T / 64);
    if (need < grid) grid = (int)need;
    bh_time_begin(ctx, BH_K_BOW_SCAN);
    hipLaunchKernelGGL(k_bow_scan, dim3(grid), dim3(BOW_T), lds, st, (const long long*)bow->pstart.p, (const uint32_t*)bow->words.p,
                       (const double*)bow->vals.p, bow->n_entries, d_qw, d_qv, nq, (double*)bow->acc.p);
    bh_time_end(ctx);
    if (n_subset > 0) {                                        // restrict to target_inds: everything else becomes "no common word"
        BH_TRY(bh_reserve(ctx, &ctx->st_g, (size_t)bow->n_entries + sizeof(int) * (size_t)n_subset + 16));
        uint8_t* flag = (uint8_t*)ctx->st_g.p; int* d_sub = (int*)(flag + (((size_t)bow->n_entries + 15) & ~(size_t)15));
        BH_HIP(ctx, hipMemsetAsync(flag, 0, (size_t)bow->n_entries, st));
        BH_HIP(ctx, hipMemcpyAsync(d_sub, subset, sizeof(int) * (size_t)n_subset, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_bow_flag_set, dim3((n_subset + 255) / 256), dim3(256), 0, st, (const int*)d_sub, n_subset, bow->n_entries, flag);
        hipLaunchKernelGGL(k_bow_flag_apply, dim3((unsigned)((bow->n_entries + 255) / 256)), dim3(256), 0, st, (const uint8_t*)flag, bow->n_entries, (double*)bow->acc.p);
    }
    if (bow->n_entries > 4 * TOPK_SLICE && max_results <= TOPK_MAXK) {           // two stages: per-slice candidates, then the merge
        const int ns = (int)((bow->n_entries + TOPK_SLICE - 1) / TOPK_SLICE);
        BH_TRY(bh_reserve(ctx, &ctx->st_f, (sizeof(double) + sizeof(long long)) * (size_t)ns * max_results));
        double* cv = (double*)ctx->st_f.p; long long* ci = (long long*)(cv + (size_t)ns * max_results);
        hipLaunchKernelGGL(k_bow_topk_slices, dim3(ns), dim3(1024), 0, st, (const double*)bow->acc.p, bow->n_entries, max_results, cv, ci);
        hipLaunchKernelGGL(k_bow_topk, dim3(1), dim3(1024), 0, st, cv, (long long)ns * max_results, max_results, (const long long*)ci, d_ids, d_sc, d_n);
    } else
        hipLaunchKernelGGL(k_bow_topk, dim3(1), dim3(1024), 0, st, (double*)bow->acc.p, bow->n_entries, max_results, (const long long*)nullptr, d_ids, d_sc, d_n);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(scores, d_sc, sizeof(double) * max_results, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(ids, d_ids, sizeof(int) * max_results, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_results, d_n, sizeof(int), hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}

extern "C" int bh_bow_query_l1_subset(bh_bow* bow, const uint32_t* q_words, const double* q_vals, int nq, const int* subset,
                                      int n_subset, int max_results, int* ids, double* scores, int* n_results)
{
    if (!bow) return BH_ERR_INVALID;
    if (n_subset < 0 || (n_subset && !subset)) return bh_set_err(bow->ctx, BH_ERR_INVALID, "bh_bow_query_l1_subset: bad subset");
    return bow_query_impl(bow, q_words, q_vals, nq, subset, n_subset, max_results, ids, scores, n_results, BH_MEM_HOST);
}

// ---- L1 comparison of two stored entries (the fork's Database::compareBowsL1 behind Dbow.compareFrames, slam/bow_db.py:47-57 ->
// dbow3.cpp:154-162): number of common words and Score = -sum(|a - b| - |a| - |b|) / 2 over them.  One wave: lane l takes words
// l, l + 64, ... of entry i and binary-searches entry j; the partial sums are added in lane order.
__global__ __launch_bounds__(64) void k_bow_compare(const long long* __restrict__ pstart, const uint32_t* __restrict__ words,
                                                    const double* __restrict__ vals, int ei, int ej, double* __restrict__ out)
{
    const int lane = threadIdx.x;
    const long long a0 = pstart[ei], a1 = pstart[ei + 1], b0 = pstart[ej], b1 = pstart[ej + 1];
    double acc = 0; int cnt = 0;
    for (long long a = a0 + lane; a < a1; a += 64) {
        const uint32_t w = words[a];
        if (w == BOW_PAD_WORD) continue;
        long long lo = b0, hi = b1;
        while (lo < hi) { const long long mid = (lo + hi) >> 1; if (words[mid] < w) lo = mid + 1; else hi = mid; }
        if (lo < b1 && words[lo] == w) { const double x = vals[a], y = vals[lo]; acc += fabs(x - y) - fabs(x) - fabs(y); cnt++; }
    }
    __shared__ double s_a[64]; __shared__ int s_c[64];
    s_a[lane] = acc; s_c[lane] = cnt;
    __syncthreads();
    if (lane == 0) { double t = 0; int c = 0; for (int i = 0; i < 64; i++) { t += s_a[i]; c += s_c[i]; } out[0] = (double)c; out[1] = -t / 2.0; }
}

extern "C" int bh_bow_compare_l1(bh_bow* bow, int entry_i, int entry_j, int* n_common, double* score)
{
    if (!bow) return BH_ERR_INVALID;
    bh_ctx* ctx = bow->ctx;
    if (!n_common || !score || entry_i < 0 || entry_j < 0 || entry_i >= bow->n_entries || entry_j >= bow->n_entries)
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_compare_l1: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    BH_TRY(bh_reserve(ctx, &bow->res, 64));
    hipLaunchKernelGGL(k_bow_compare, dim3(1), dim3(64), 0, ctx->stream, (const long long*)bow->pstart.p, (const uint32_t*)bow->words.p,
                       (const double*)bow->vals.p, entry_i, entry_j, (double*)bow->res.p);
    BH_HIP(ctx, hipGetLastError());
    double h[2];
    BH_HIP(ctx, hipMemcpyAsync(h, bow->res.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_common = (int)h[0]; *score = h[1];
    return BH_OK;
}

// Database::add at the BoW-vector level (slam/bow_db.py:26 -> dbow3.cpp:138-140): appends one entry, returns its id.
extern "C" int bh_bow_add_entry(bh_bow* bow, const uint32_t* words, const double* vals, int n, int* entry_id)
{
    if (!bow) return BH_ERR_INVALID;
    bh_ctx* ctx = bow->ctx;
    if (n < 0 || (n && (!words || !vals))) return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_add_entry: bad argument");
    for (int i = 1; i < n; i++) if (words[i] <= words[i - 1]) return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_add_entry: words must be strictly ascending");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (bow->h_pstart.empty()) bow->h_pstart.push_back(0);
    const long long a = bow->h_pstart.back(), npad = (n + 3) & ~3;
    std::vector<uint32_t> w(npad, BOW_PAD_WORD); std::vector<double> v(npad, 0.0);
    for (int i = 0; i < n; i++) { w[i] = words[i]; v[i] = vals[i]; }
    BH_TRY(bow_grow(ctx, &bow->words, sizeof(uint32_t) * (size_t)(a + npad + 4), sizeof(uint32_t) * (size_t)a));
    BH_TRY(bow_grow(ctx, &bow->vals, sizeof(double) * (size_t)(a + npad + 4), sizeof(double) * (size_t)a));
    BH_TRY(bow_grow(ctx, &bow->pstart, sizeof(long long) * (size_t)(bow->n_entries + 2), sizeof(long long) * (size_t)(bow->n_entries + 1)));
    BH_TRY(bow_grow(ctx, &bow->acc, sizeof(double) * (size_t)(bow->n_entries + 2), 0));
    if (npad) {
        BH_HIP(ctx, hipMemcpyAsync((uint32_t*)bow->words.p + a, w.data(), sizeof(uint32_t) * npad, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync((double*)bow->vals.p + a, v.data(), sizeof(double) * npad, hipMemcpyHostToDevice, st));
    }
    const long long ends[2] = { a, a + npad };
    BH_HIP(ctx, hipMemcpyAsync((long long*)bow->pstart.p + bow->n_entries, ends, sizeof(ends), hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    bow->h_pstart.push_back(a + npad);
    if (entry_id) *entry_id = (int)bow->n_entries;
    bow->n_entries += 1; bow->n_elems = a + npad;
    return BH_OK;
}

// Database::erase (the fork's method used at slam/bow_db.py:33 -> dbow3.cpp:142-145): the entry keeps its id but loses all
// its words (its row becomes padding), so it can never appear in a result again.
extern "C" int bh_bow_erase_entry(bh_bow* bow, int entry_id)
{
    if (!bow) return BH_ERR_INVALID;
    bh_ctx* ctx = bow->ctx;
    if (entry_id < 0 || entry_id >= bow->n_entries || (long long)bow->h_pstart.size() <= entry_id + 1)
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_erase_entry: no such entry");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    const long long a = bow->h_pstart[entry_id], e = bow->h_pstart[entry_id + 1];
    if (e > a) BH_HIP(ctx, hipMemsetAsync((uint32_t*)bow->words.p + a, 0xff, sizeof(uint32_t) * (size_t)(e - a), ctx->stream));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BH_OK;
}
