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
#include <chrono>
#include "bh_scan.h"

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
    // inverted file (DBoW3's own structure: word -> (entry, weight) rows), rebuilt lazily after the database changed
    bh_buf inv_start, inv_entry, inv_val, inv_tmp, bcnt, bkey, bval, cand_v, cand_i;
    int inv_ready = 0;             // rows are current
    int dirty_queries = 0;         // queries answered by the scan since the database last changed (see inv_ensure)
    int inv_usable = 1;            // every word id < 2^20 (else only the scan can answer)
    int mode = 0;                  // 0 = inverted file where possible, 1 = always scan (bh_bow_set_mode)
    long long inv_nb = 0;          // buckets of 64 entries
    int bcap = 0;                  // hit records a bucket can hold
    char* hq = nullptr;            // pinned, device-visible: per query slot the query going in and the results coming out
    size_t hq_cap = 0;
    int inv_nslots = 1;            // query slots (scratch sets): 1 for bh_bow_query_l1, INV_QSLOTS once bh_bow_query_l1_batch has been used
    int slots_ready = 0;
    bh_buf inv_slots, inv_acc;     // device copy of the slot table; the slots' accumulator arrays
    bh_buf dq;                     // device-resident queries (BH_MEM_DEVICE): a slot of their own (slot 0's scratch, the caller's query arrays), its in / out block and the "answered" flag
    char   slot0_host[128] = {0};  // host copy of slot 0 (an inv_slot), the template of that slot
    int q_seq = 0;                 // tag of the last single host query (the result block carries it: the host polls for it instead of a stream wake-up)
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
                                                    double* __restrict__ acc_out, const int* __restrict__ skip)
{
    if (skip && *skip) return;                                 // a device-resident query the inverted file has already answered (uniform)
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
                                                          double* __restrict__ cand_v, long long* __restrict__ cand_i, const int* __restrict__ skip)
{
    if (skip && *skip) return;
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
                                                   int* __restrict__ ids, double* __restrict__ scores, int* __restrict__ n_res, const int* __restrict__ skip)
{
    if (skip && *skip) return;
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

#define BOW_SCAN_LDS  (BOW_BITS / 8 + BOW_QMAX * (sizeof(uint32_t) + sizeof(double)))
#define INV_HEAVY_LDS ((sizeof(uint32_t) + sizeof(double)) * 2 * INV_HCAP + sizeof(uint32_t) * (4096 + 4096 + 64))
static int bow_set_func_attrs(bh_ctx* ctx);      // (defined below the kernels it names)

extern "C" int bh_bow_create(bh_ctx* ctx, bh_bow** out)
{
    if (!ctx || !out) return BH_ERR_INVALID;
    // The large-LDS attributes of k_bow_scan / k_inv_heavy are per DEVICE: set them here, on this context's device, for every handle (a
    // process-wide "already set" flag left a second GPU of the same process without them and let two threads race on the flag).
    BH_HIP(ctx, hipSetDevice(ctx->device));
    BH_TRY(bow_set_func_attrs(ctx));
    bh_bow* b = new bh_bow(); b->ctx = ctx; *out = b;
    return BH_OK;
}

extern "C" void bh_bow_destroy(bh_bow* bow)
{
    if (!bow) return;
    (void)hipSetDevice(bow->ctx->device);
    (void)hipStreamSynchronize(bow->ctx->stream);
    bh_buf* all[] = { &bow->pstart, &bow->words, &bow->vals, &bow->acc, &bow->qw, &bow->qv, &bow->res,
                      &bow->inv_start, &bow->inv_entry, &bow->inv_val, &bow->inv_tmp, &bow->bcnt, &bow->bkey, &bow->bval, &bow->cand_v, &bow->cand_i, &bow->inv_slots, &bow->inv_acc, &bow->dq };
    for (bh_buf* b : all) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    if (bow->hq) (void)hipHostFree(bow->hq);
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
    bow->inv_ready = 0; bow->dirty_queries = 1 << 20;          // a bulk load: the first query builds the inverted file
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

// ================================================================================================ inverted-file query
// DBoW3 answers a query through its inverted file: for every word of the query it walks that word's row of (entry, weight) items
// (Database::queryL1, behind slam/bow_db.py:37-42 -> docker/pydbow3/src/dbow3.cpp:147-152).  A query of ~1000 words against configs[3]
// touches ~100 items per word, 1.2 MB, where the scan above streams the word ids of the whole database (400 MB).
//   build (lazy, after the database changed)  word histogram -> exclusive scan -> scatter of (entry, weight) into the rows (order
//          inside a row is arbitrary);
//   k_inv_gather   one wave per query word: every item of the row becomes a hit record (entry & 63 << 11 | rank of the word in the
//          query, term |q - d| - |q| - |d|) appended to one of the four lists of its bucket of 64 entries (four, picked by the rank's
//          low bits: a relocalisation query hits its near-duplicate entry ~1000 times, and returning atomics on ONE counter serialise);
//          what a list cannot hold goes to a global overflow list with the full entry id;
//   k_inv_finish   every workgroup: a wave per bucket ranks the bucket's records by (entry, word rank) and adds them IN THAT ORDER - per
//          entry the ascending word order of DBoW3's accumulation, so scores are bit-identical to the scan and to the oracle; buckets
//          that overflowed are then taken by the whole workgroup - its 1024 lanes rank a bucket's lists + its share of the overflow
//          list through a (slot, rank) bitmap, the same ordered sums.  The LAST workgroup to arrive (agent-scope counter)
//          selects: a threshold from the 16 wave minima of the bucket minima (k <= 16 of them are k actual values, so at least k
//          accumulators lie at or below it), the few accumulators at or below it collected and ordered by (value, id), results to
//          pinned host memory.  (Rounds 3 - 4 ran these as three launches, k_inv_reduce / k_inv_heavy / k_inv_select.)
// Anything that does not fit (overflow list, a heavy bucket beyond 4096 records, more than 1024 accumulators at the selection threshold)
// raises a flag: the scan answers.
#define INV_WORDS BOW_BITS
#define INV_KMAX 16                                           // results the select kernel can deliver (k <= 16 wave minima)
#define INV_NSUB 4                                            // lists per bucket
#define INV_RANK_BITS 11                                      // query ranks < 2048 (BOW_QMAX = 1536)
#define INV_OVF_CAP 65536                                     // records of the global overflow list
#define INV_HCAP 4096                                         // records a heavy bucket's workgroup can rank
#define INV_SELCAP 1024                                       // accumulators at or below the threshold

struct inv_ctl { int ovf_cnt, flags, tick, pad; };            // tick: workgroups of k_inv_finish that have summed their buckets
// One query slot: its own scratch (bucket lists, overflow list, accumulators, control block) and its pinned in / out block
//   [int nq][int k][pad 8] [u32 words[BOW_QMAX]] [f64 vals[BOW_QMAX]] [int n][int fallback][f64 scores[INV_KMAX]][int ids[INV_KMAX]].
// bh_bow_query_l1 uses slot 0; bh_bow_query_l1_batch runs up to INV_QSLOTS queries per launch (blockIdx.y = slot).
#define INV_QSLOTS 16
#define INV_PIN_W 16
#define INV_PIN_V (INV_PIN_W + 4 * BOW_QMAX)
#define INV_PIN_R (INV_PIN_V + 8 * BOW_QMAX)
#define INV_PIN_BYTES (INV_PIN_R + 8 + 12 * INV_KMAX + 56)
struct inv_slot { const int* hdr; const uint32_t* qw; const double* qv; int* bcnt; uint32_t* bkey; double* bval; inv_ctl* ctl; uint32_t* okey; double* oval; double* acc; double* bmin; char* hout; };

__global__ __launch_bounds__(256) void k_inv_count(const uint32_t* __restrict__ pw, long long n, uint32_t* __restrict__ cnt, int* __restrict__ bad)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const uint32_t w = pw[i];
        if (w == BOW_PAD_WORD) continue;
        if (w >= INV_WORDS) { *bad = 1; continue; }
        atomicAdd(&cnt[w], 1u);
    }
}
__global__ __launch_bounds__(256) void k_inv_fill(const long long* __restrict__ pstart, const uint32_t* __restrict__ pw, const double* __restrict__ pv,
                                                  long long n_entries, const long long* __restrict__ inv_start, uint32_t* __restrict__ cursor,
                                                  uint32_t* __restrict__ inv_entry, double* __restrict__ inv_val)
{
    const long long e = blockIdx.x;
    if (e >= n_entries) return;
    const long long a = pstart[e], b = pstart[e + 1];
    for (long long i = a + threadIdx.x; i < b; i += 256) {
        const uint32_t w = pw[i];
        if (w >= INV_WORDS) continue;                          // padding (and out-of-range ids: inv_usable is off then)
        const long long pos = inv_start[w] + atomicAdd(&cursor[w], 1u);
        inv_entry[pos] = (uint32_t)e; inv_val[pos] = pv[i];
    }
}
__global__ void k_inv_widen(const uint32_t* __restrict__ cnt, long long* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = cnt[i];
}

__global__ __launch_bounds__(256) void k_inv_gather(const inv_slot* __restrict__ slots, const long long* __restrict__ inv_start,
                                                    const uint32_t* __restrict__ inv_entry, const double* __restrict__ inv_val, int scap, int nq1)
{
    const inv_slot S = slots[blockIdx.y];
    // (nq1 >= 0: a single host query - its length rides in the launch instead of behind a read of the pinned block, one PCIe round trip less)
    const uint32_t* __restrict__ qw = S.qw; const double* __restrict__ qv = S.qv; const int nq = nq1 >= 0 ? nq1 : S.hdr[0];
    int* __restrict__ bcnt = S.bcnt; uint32_t* __restrict__ bkey = S.bkey; double* __restrict__ bval = S.bval; inv_ctl* __restrict__ ctl = S.ctl;
    uint32_t* __restrict__ okey = S.okey; double* __restrict__ oval = S.oval;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= nq) return;
    const uint32_t w = qw[r];
    if (w >= INV_WORDS) return;                                // no row: no database entry holds such a word
    const double q = qv[r];
    const long long a = inv_start[w], b = inv_start[w + 1];
    for (long long i = a + lane; i < b; i += 64) {
        const uint32_t e = inv_entry[i];
        const double d = inv_val[i];
        const double term = fabs(q - d) - fabs(q) - fabs(d);
        const size_t li = (size_t)(e >> 6) * INV_NSUB + (r & (INV_NSUB - 1));
        const int pos = atomicAdd(&bcnt[li], 1);
        if (pos < scap) { bkey[li * scap + pos] = ((e & 63u) << INV_RANK_BITS) | (uint32_t)r; bval[li * scap + pos] = term; }
        else {
            const int p2 = atomicAdd(&ctl->ovf_cnt, 1);
            if (p2 < INV_OVF_CAP) { okey[p2] = (e << INV_RANK_BITS) | (uint32_t)r; oval[p2] = term; }
            else ctl->flags = 1;
        }
    }
}

// rank sort of n records (keys unique) held in s_key / s_val by `nthr` cooperating lanes -> s_okey / s_oval; caller synchronises around it
__device__ __forceinline__ void inv_rank_sort(const uint32_t* s_key, const double* s_val, uint32_t* s_okey, double* s_oval, int n, int t, int nthr)
{
    for (int i = t; i < n; i += nthr) {
        const uint32_t key = s_key[i];
        int pos = 0;
        for (int j = 0; j < n; j++) pos += s_key[j] < key ? 1 : 0;
        s_okey[pos] = key; s_oval[pos] = s_val[i];
    }
}
// the ordered sum of entry-slot l's records in a list sorted by (slot << INV_RANK_BITS | rank).  Both ends of the slot's segment are
// found by binary search first: a loop that tests every loaded key before it adds serialises on the LDS latency (57 us for the 800
// hits of a relocalisation query's near-duplicate entry).
__device__ __forceinline__ int inv_lower_bound(const uint32_t* s_okey, int n, uint32_t k0)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_okey[mid] < k0) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ double inv_ordered_sum(const uint32_t* s_okey, const double* s_oval, int n, int l, bool* any)
{
    const int a = inv_lower_bound(s_okey, n, (uint32_t)l << INV_RANK_BITS), b = inv_lower_bound(s_okey, n, (uint32_t)(l + 1) << INV_RANK_BITS);
    // the additions are one dependent chain (the order is the result), so the LDS reads must not sit in it: eight values are fetched
    // while the previous eight are added (a read-wait-add loop was 25 of the heavy kernel's 35 us for a relocalisation query's ~1000 hits)
    double acc = 0.0;
    int i = a;
    if (b - a >= 8) {
        double c[8], nx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = s_oval[i + u];
        for (; i + 16 <= b; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) nx[u] = s_oval[i + 8 + u];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += c[u];
#pragma unroll
            for (int u = 0; u < 8; u++) c[u] = nx[u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += c[u];
        i += 8;
    }
    for (; i < b; i++) acc += s_oval[i];
    *any = b > a;
    return acc;
}
// minimum of the wave's 64 values -> lane 0
__device__ __forceinline__ double inv_wave_min(double v, int lane)
{
    for (int m = 32; m > 0; m >>= 1) v = fmin(v, bow_shfl(v, lane ^ m));
    return v;
}

#define INV_LCAP 256                                           // records a light bucket's wave ranks (4 lists x 64)
#define INV_FIN_T 1024                                         // k_inv_finish: 16 waves = 16 buckets per workgroup
#define INV_FIN_BK (INV_FIN_T / 64)

// one heavy bucket (its lists + its share of the overflow list) by the whole workgroup: O(n) sort by a (slot, rank) bitmap, ordered sums
__device__ __forceinline__ void inv_heavy_bucket(const inv_slot& S, long long bk, const int* cnt4, int scap, long long n_entries, const uint8_t* __restrict__ keep, uint32_t* hsm, int* s_n)
{
    const uint32_t* __restrict__ bkey = S.bkey; const double* __restrict__ bval = S.bval; double* acc_out = S.acc; inv_ctl* ctl = S.ctl;
    const uint32_t* __restrict__ okey = S.okey; const double* __restrict__ oval = S.oval; double* bmin = S.bmin;
    uint32_t* s_key = hsm; uint32_t* s_okey = hsm + INV_HCAP;
    double* s_val = reinterpret_cast<double*>(hsm + 2 * INV_HCAP); double* s_oval = s_val + INV_HCAP;
    const int tid = threadIdx.x;
    int n0 = 0;
    for (int u = 0; u < INV_NSUB; u++) {
        const int cu = cnt4[u];
        for (int i = tid; i < cu; i += INV_FIN_T) { s_key[n0 + i] = bkey[((size_t)bk * INV_NSUB + u) * scap + i]; s_val[n0 + i] = bval[((size_t)bk * INV_NSUB + u) * scap + i]; }
        n0 += cu;
    }
    if (tid == 0) *s_n = n0;
    __syncthreads();
    const int novf = min(ctl->ovf_cnt, INV_OVF_CAP);
    for (int i0 = 0; i0 < novf; i0 += INV_FIN_T) {             // (whole waves iterate: the append below is aggregated per wave - 800 returning
        const int i = i0 + tid;                                //  LDS atomics on one counter serialise: 38 us)
        const uint32_t k = i < novf ? okey[i] : 0xffffffffu;
        const double ov = i < novf ? oval[i] : 0.0;            // (asked for with the key: not a second round trip behind the test)
        const bool mine = i < novf && (long long)(k >> (INV_RANK_BITS + 6)) == bk;
        const unsigned long long bm = __ballot(mine);
        if (bm) {
            const int lane = tid & 63, leader = __ffsll((long long)bm) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(s_n, __popcll(bm));
            base = __shfl(base, leader, 64);
            const int pos = base + __popcll(bm & ((1ull << lane) - 1ull));
            if (mine && pos < INV_HCAP) { s_key[pos] = k & ((64u << INV_RANK_BITS) - 1u); s_val[pos] = ov; }
        }
    }
    __syncthreads();
    const int n = *s_n;
    if (n > INV_HCAP) { if (tid == 0) ctl->flags = 1; __syncthreads(); return; }      // (uniform: the scan answers this query)
    // Sort by (slot, rank) in O(n): the (slot, rank) pairs are distinct, so a bitmap of 64 slots x 2048 ranks says where every record
    // goes: position = records of earlier slots + set bits of the slot below the rank.  (Ranking every record against every other,
    // as the light buckets do, is 740 k comparisons on one CU for a relocalisation query's near-duplicate entry: 28 us.)
    uint32_t* s_bm = reinterpret_cast<uint32_t*>(s_oval + INV_HCAP);       // [64][64] words
    int* s_wp = reinterpret_cast<int*>(s_bm + 4096);                        // [64][64] set bits of the slot before word w
    int* s_so = s_wp + 4096;                                                // [64] records of earlier slots
    for (int i = tid; i < 4096; i += INV_FIN_T) s_bm[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += INV_FIN_T) { const uint32_t key = s_key[i]; atomicOr(&s_bm[key >> 5], 1u << (key & 31)); }      // key = slot << 11 | rank: word = key >> 5
    __syncthreads();
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int sl = wv; sl < 64; sl += INV_FIN_BK) {         // a wave per slot: exclusive prefix of its 64 words' popcounts
            const int c = __popc(s_bm[sl * 64 + lane]);
            int incl = c;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
            s_wp[sl * 64 + lane] = incl - c;
            if (lane == 63) s_so[sl] = incl;                   // the slot's total for now
        }
    }
    __syncthreads();
    if (tid < 64) {                                            // exclusive prefix over the slots
        const int c = s_so[tid];
        int incl = c;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (tid >= d) incl += o; }
        s_so[tid] = incl - c;
    }
    __syncthreads();
    for (int i = tid; i < n; i += INV_FIN_T) {
        const uint32_t key = s_key[i];
        const int pos = s_so[key >> INV_RANK_BITS] + s_wp[key >> 5] + __popc(s_bm[key >> 5] & ((1u << (key & 31)) - 1u));
        s_okey[pos] = key; s_oval[pos] = s_val[i];
    }
    __syncthreads();
    if (tid < 64) {
        bool any;
        const double acc = inv_ordered_sum(s_okey, s_oval, n, tid, &any);
        const long long e = bk * 64 + tid;
        const double v = (any && e < n_entries && (!keep || keep[e])) ? acc : INFINITY;
        if (e < n_entries) acc_out[e] = v;
        const double bm = inv_wave_min(v, tid);
        if (tid == 0) bmin[bk] = bm;
    }
    __syncthreads();                                           // (the staging arrays are free for the next bucket; acc / bmin are read after a fence)
}

// results -> pinned host memory: [int n][int fallback][double scores[k]][int ids[k]].  Works from the buckets' minima: the k-th
// smallest of the 16 wave minima T is an upper bound of the k-th best accumulator (k <= 16 distinct buckets hold a value <= T); only
// buckets whose minimum is <= T can hold a result, and there are few of them.
#define INV_PIN_SEQ (2 + 3 * INV_KMAX)                       // int index of the tag inside the result block (behind n, fallback, scores, ids)
__device__ __forceinline__ void inv_select(const inv_slot& S, long long nb, long long n_entries, int k, uint32_t* hsm, int* s_cnt, int seq)
{
    const double* __restrict__ acc = S.acc; const double* bmin = S.bmin; inv_ctl* ctl = S.ctl;
    char* __restrict__ hout = S.hout;
    double* s_wmin = reinterpret_cast<double*>(hsm);           // [16]
    double* s_cv = s_wmin + 16;                                // [INV_SELCAP]
    int* s_ci = reinterpret_cast<int*>(s_cv + INV_SELCAP);     // [INV_SELCAP]
    int* s_bl = s_ci + INV_SELCAP;                             // [INV_SELCAP / 16]
    double* s_T = reinterpret_cast<double*>(s_bl + INV_SELCAP / 16);
    int& s_m = s_cnt[0]; int& s_nb = s_cnt[1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // buckets are dealt to the 16 waves round-robin (bucket b -> wave b % 16): a database of a few hundred buckets still gives every wave a
    // minimum, so T is finite for k <= 16 (dealt in runs of 64, anything below 65 536 entries left waves empty and went to the scan)
    const int tb = ((tid & 63) << 4) | wv;                     // this thread's first bucket; the others follow every 1024
    double mn = INFINITY, mine[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const long long b = tb + INV_FIN_T * u; mine[u] = b < nb ? bmin[b] : INFINITY; mn = fmin(mn, mine[u]); }
    for (long long b = tb + 4 * INV_FIN_T; b < nb; b += INV_FIN_T) mn = fmin(mn, bmin[b]);
    mn = inv_wave_min(mn, lane);
    if (lane == 0) s_wmin[wv] = mn;
    if (tid == 0) { s_m = 0; s_nb = 0; }
    __syncthreads();
    if (tid == 0) {
        double v[16];
        for (int i = 0; i < 16; i++) v[i] = s_wmin[i];
        for (int i = 0; i < 16; i++) for (int j = i + 1; j < 16; j++) if (v[j] < v[i]) { const double t = v[i]; v[i] = v[j]; v[j] = t; }
        *s_T = v[k - 1];
    }
    __syncthreads();
    const double T = *s_T;                                     // +inf when fewer than k waves saw a hit: then every bucket with a hit is listed
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long long b = tb + INV_FIN_T * u;
        if (mine[u] <= T && mine[u] < INFINITY) { const int p = atomicAdd(&s_nb, 1); if (p < INV_SELCAP / 16) s_bl[p] = (int)b; }
    }
    for (long long b = tb + 4 * INV_FIN_T; b < nb; b += INV_FIN_T) {
        const double v = bmin[b];
        if (v <= T && v < INFINITY) { const int p = atomicAdd(&s_nb, 1); if (p < INV_SELCAP / 16) s_bl[p] = (int)b; }
    }
    __syncthreads();
    const int nbl = s_nb;
    int* hi = reinterpret_cast<int*>(hout);
    bool bad = nbl > INV_SELCAP / 16;
    if (!bad) {
        for (int i = wv; i < nbl; i += INV_FIN_BK) {           // a wave per listed bucket: one accumulator per lane
            const long long e = (long long)s_bl[i] * 64 + lane;
            const double v = e < n_entries ? acc[e] : INFINITY;
            if (v <= T && v < INFINITY) { const int p = atomicAdd(&s_m, 1); if (p < INV_SELCAP) { s_cv[p] = v; s_ci[p] = (int)e; } }
        }
    }
    __syncthreads();
    const int m = s_m;
    bad = bad || m > INV_SELCAP;
    if (!bad && tid < m) {                                     // order by (value, id): position = number of candidates that come first
        const double v = s_cv[tid]; const int id = s_ci[tid];
        int pos = 0;
        for (int j = 0; j < m; j++) { const double ov = s_cv[j]; pos += (ov < v || (ov == v && s_ci[j] < id)) ? 1 : 0; }
        if (pos < k) { reinterpret_cast<double*>(hout + 8)[pos] = -v / 2.0; reinterpret_cast<int*>(hout + 8 + 8 * INV_KMAX)[pos] = id; }
    }
    if (seq) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __syncthreads(); }      // the scores / ids of the other lanes before the tag
    if (tid == 0) {
        hi[0] = bad ? 0 : min(m, k); hi[1] = bad ? 1 : ctl->flags;
        ctl->ovf_cnt = 0; ctl->flags = 0;      // counters ready for the next query
        // a single host query polls for this tag (pinned, coherent memory) instead of waiting for the stream: the wake-up costs more than the kernels
        if (seq) __hip_atomic_store(&hi[INV_PIN_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// k_inv_finish (round 5): everything behind the gather in ONE launch (was k_inv_reduce -> k_inv_heavy -> k_inv_select: three dependent
// launches of 7 + 35 + 21 us for a relocalisation query, most of it launch, drain and first-load latency of one-workgroup kernels).
//   every workgroup   a wave per bucket of 64 entries: rank its records by (entry, word rank), add them in that order, bucket minimum;
//                     then the buckets of this workgroup whose lists overflowed (one for a relocalisation query, none for most others),
//                     one after the other with all 1024 lanes: lists + the bucket's share of the overflow list, ranked through a
//                     (slot, rank) bitmap, the same ordered sums;
//   arrival           release fence, one agent-scope add on ctl->tick per workgroup;
//   the last arriver  (acquire fence) selects the k best and writes the slot's result block.  No workgroup waits for another.
__global__ __launch_bounds__(INV_FIN_T) void k_inv_finish(const inv_slot* __restrict__ slots, int scap, long long n_entries, const uint8_t* __restrict__ keep,
                                                          long long nb, int k1, int seq)
{
    extern __shared__ uint32_t hsm[];
    __shared__ int s_cnt[4];
    __shared__ int s_hn, s_hb[INV_FIN_BK], s_hc[INV_FIN_BK][INV_NSUB];      // this workgroup's heavy buckets: wave index, list lengths
    if (threadIdx.x == 0) s_hn = 0;
    __syncthreads();
    const inv_slot S = slots[blockIdx.y];
    const int k = k1 >= 0 ? k1 : S.hdr[1];                      // (asked for now, used by the last arriver only)
    inv_ctl* ctl = S.ctl;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    {
        int* __restrict__ bcnt = S.bcnt; const uint32_t* __restrict__ bkey = S.bkey; const double* __restrict__ bval = S.bval;
        double* acc_out = S.acc; double* bmin = S.bmin;
        // per wave: key[256] okey[256] (u32), val[256] oval[256] (f64) = 6 KB
        uint32_t* w_key = hsm + (size_t)wv * (6 * INV_LCAP); uint32_t* w_okey = w_key + INV_LCAP;
        double* w_val = reinterpret_cast<double*>(w_okey + INV_LCAP); double* w_oval = w_val + INV_LCAP;
        const long long bk = (long long)blockIdx.x * INV_FIN_BK + wv;
        if (bk < nb) {
            int c[INV_NSUB], n = 0; bool heavy = false;
#pragma unroll
            for (int u = 0; u < INV_NSUB; u++) { c[u] = bcnt[bk * INV_NSUB + u]; heavy = heavy || c[u] > scap; }
            if (lane < INV_NSUB) bcnt[bk * INV_NSUB + lane] = 0;       // ready for the next query
            if (heavy) {                                               // some of its records sit in the overflow list: the whole workgroup takes the bucket below
                if (lane == 0) {
                    const int hp = atomicAdd(&s_hn, 1);
                    s_hb[hp] = wv; for (int u = 0; u < INV_NSUB; u++) s_hc[hp][u] = min(c[u], scap);
                }
            } else {
#pragma unroll
                for (int u = 0; u < INV_NSUB; u++) {
                    for (int i = lane; i < c[u]; i += 64) { w_key[n + i] = bkey[((size_t)bk * INV_NSUB + u) * scap + i]; w_val[n + i] = bval[((size_t)bk * INV_NSUB + u) * scap + i]; }
                    n += c[u];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                inv_rank_sort(w_key, w_val, w_okey, w_oval, n, lane, 64);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                bool any;
                const double acc = inv_ordered_sum(w_okey, w_oval, n, lane, &any);   // lane l owns entry 64 bk + l: its hits in ascending word order
                const long long e = bk * 64 + lane;
                const double v = (any && e < n_entries && (!keep || keep[e])) ? acc : INFINITY;
                if (e < n_entries) acc_out[e] = v;
                const double bm = inv_wave_min(v, lane);
                if (lane == 0) bmin[bk] = bm;
            }
        }
    }
    // ---- this workgroup's heavy buckets, one after the other, all 1024 lanes on each (the waves' staging areas are free now)
    __syncthreads();
    {
        const int nh = s_hn;
        for (int hb = 0; hb < nh; hb++) inv_heavy_bucket(S, (long long)blockIdx.x * INV_FIN_BK + s_hb[hb], s_hc[hb], scap, n_entries, keep, hsm, &s_cnt[3]);
    }
    // ---- arrival: this workgroup's accumulators and minima are out before its count is
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int prev = __hip_atomic_fetch_add(&ctl->tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_cnt[2] = prev == (int)gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!s_cnt[2]) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid == 0) ctl->tick = 0;                                // (the next query's launch is ordered behind this one)
    inv_select(S, nb, n_entries, k, hsm, s_cnt, seq);
}

// Scratch, pinned in / out blocks and the device slot table for `nslots` concurrent queries (sizes follow the current database).
static int inv_alloc_slots(bh_bow* bow, int nslots)
{
    bh_ctx* ctx = bow->ctx;
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t nl = (size_t)bow->inv_nb * INV_NSUB;
    const size_t cb = al(sizeof(int) * (nl + 64) + sizeof(inv_ctl)), kb = al(sizeof(uint32_t) * nl * bow->bcap + 64), vb = al(sizeof(double) * nl * bow->bcap + 64);
    const size_t ob = al(sizeof(uint32_t) * INV_OVF_CAP), ovb = al(sizeof(double) * (INV_OVF_CAP + (size_t)bow->inv_nb + 8)), ab = al(sizeof(double) * ((size_t)bow->n_entries + 2));
    BH_TRY(bh_reserve(ctx, &bow->bcnt, cb * nslots)); BH_TRY(bh_reserve(ctx, &bow->bkey, kb * nslots)); BH_TRY(bh_reserve(ctx, &bow->bval, vb * nslots));
    BH_TRY(bh_reserve(ctx, &bow->cand_i, ob * nslots)); BH_TRY(bh_reserve(ctx, &bow->cand_v, ovb * nslots));      // the overflow lists; bucket minima behind their values
    BH_TRY(bh_reserve(ctx, &bow->inv_acc, ab * nslots)); BH_TRY(bh_reserve(ctx, &bow->inv_slots, sizeof(inv_slot) * nslots));
    const size_t need = (size_t)INV_PIN_BYTES * nslots;
    if (need > bow->hq_cap) {
        BH_HIP(ctx, hipStreamSynchronize(st));
        if (bow->hq) (void)hipHostFree(bow->hq);
        bow->hq = nullptr; bow->hq_cap = 0;
        if (hipHostMalloc((void**)&bow->hq, need, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); bow->hq = nullptr; bow->inv_usable = 0; return BH_OK; }
        bow->hq_cap = need;
        memset(bow->hq, 0, need);                               // (the result tags start at 0: no query carries that tag)
    }
    BH_HIP(ctx, hipMemsetAsync(bow->bcnt.p, 0, cb * nslots, st));
    std::vector<inv_slot> tab(nslots);
    for (int q = 0; q < nslots; q++) {
        inv_slot& S = tab[q];
        char* hp = bow->hq + (size_t)INV_PIN_BYTES * q;
        S.hdr = (const int*)hp; S.qw = (const uint32_t*)(hp + INV_PIN_W); S.qv = (const double*)(hp + INV_PIN_V); S.hout = hp + INV_PIN_R;
        S.bcnt = (int*)((char*)bow->bcnt.p + cb * q); S.ctl = (inv_ctl*)(S.bcnt + nl + 64);
        S.bkey = (uint32_t*)((char*)bow->bkey.p + kb * q); S.bval = (double*)((char*)bow->bval.p + vb * q);
        S.okey = (uint32_t*)((char*)bow->cand_i.p + ob * q); S.oval = (double*)((char*)bow->cand_v.p + ovb * q); S.bmin = S.oval + INV_OVF_CAP;
        S.acc = (double*)((char*)bow->inv_acc.p + ab * q);
    }
    BH_HIP(ctx, hipMemcpyAsync(bow->inv_slots.p, tab.data(), sizeof(inv_slot) * nslots, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    static_assert(sizeof(inv_slot) <= sizeof(bow->slot0_host), "slot template");
    memcpy(bow->slot0_host, &tab[0], sizeof(inv_slot));
    bow->inv_nslots = nslots;
    return BH_OK;
}

static int inv_build(bh_bow* bow)
{
    bh_ctx* ctx = bow->ctx;
    hipStream_t st = ctx->stream;
    const long long nel = bow->n_elems, ne = bow->n_entries;
    BH_TRY(bh_reserve(ctx, &bow->inv_start, sizeof(long long) * (INV_WORDS + 2)));
    BH_TRY(bh_reserve(ctx, &bow->inv_entry, sizeof(uint32_t) * (size_t)(nel + 4))); BH_TRY(bh_reserve(ctx, &bow->inv_val, sizeof(double) * (size_t)(nel + 4)));
    const size_t scan_bytes = bh_exscan_scratch<long long>(INV_WORDS + 1);
    const size_t off_cnt = 0, off_wide = off_cnt + sizeof(uint32_t) * (INV_WORDS + 64), off_scan = off_wide + sizeof(long long) * (INV_WORDS + 2);
    BH_TRY(bh_reserve(ctx, &bow->inv_tmp, off_scan + scan_bytes + 256));
    uint32_t* cnt = (uint32_t*)((char*)bow->inv_tmp.p + off_cnt); long long* wide = (long long*)((char*)bow->inv_tmp.p + off_wide);
    int* bad = (int*)(cnt + INV_WORDS + 8);
    BH_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(uint32_t) * (INV_WORDS + 64), st));
    const int g = (int)std::min<long long>(4096, (nel + 255) / 256 + 1);
    hipLaunchKernelGGL(k_inv_count, dim3(g), dim3(256), 0, st, (const uint32_t*)bow->words.p, nel, cnt, bad);
    hipLaunchKernelGGL(k_inv_widen, dim3((INV_WORDS + 1 + 255) / 256), dim3(256), 0, st, (const uint32_t*)cnt, wide, INV_WORDS + 1);
    BH_TRY((bh_exscan<long long, long long>(ctx, st, (const long long*)wide, (long long*)bow->inv_start.p, INV_WORDS + 1, (char*)bow->inv_tmp.p + off_scan)));
    int hbad = 0;
    BH_HIP(ctx, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(uint32_t) * INV_WORDS, st));                      // now the per-row fill cursors
    hipLaunchKernelGGL(k_inv_fill, dim3((unsigned)ne), dim3(256), 0, st, (const long long*)bow->pstart.p, (const uint32_t*)bow->words.p, (const double*)bow->vals.p,
                       ne, (const long long*)bow->inv_start.p, cnt, (uint32_t*)bow->inv_entry.p, (double*)bow->inv_val.p);
    BH_HIP(ctx, hipGetLastError());
    BH_HIP(ctx, hipStreamSynchronize(st));
    bow->inv_usable = (hbad || ne > (1ll << 21)) ? 0 : 1;      // overflow records pack (entry << 11 | rank) into 32 bits: entry ids stay below 2^21, beyond that the scan answers
    // buckets of 64 entries, INV_NSUB lists each: a list holds 3 x the hits a full 1536-word query expects of it, 16 .. 64
    bow->inv_nb = (ne + 63) / 64;
    const double per_entry = ne ? (double)nel / (double)ne : 0.0;                               // words per entry (padded)
    const long long cap = (long long)(3.0 * 64.0 * per_entry * (double)BOW_QMAX / (double)INV_WORDS / INV_NSUB) + 16;
    bow->bcap = (int)std::min<long long>(INV_LCAP / INV_NSUB, (cap + 15) / 16 * 16);          // per list
    BH_TRY(inv_alloc_slots(bow, bow->inv_nslots));
    bow->inv_ready = 1;
    return BH_OK;
}

// The two launches of the inverted-file path for the first G slots (grid.y = slot), then one synchronisation.
static int inv_launch(bh_bow* bow, int G, int max_nq, const uint8_t* keep, const inv_slot* dev_slot = nullptr, int nq1 = -1, int k1 = -1)
{
    // a single host query (nq1 >= 0) is tagged: 1, 2, ... (0 = untagged)
    const int seq = (!dev_slot && G == 1 && nq1 >= 0 && !getenv("BH_BOW_SYNC")) ? (bow->q_seq = bow->q_seq == 0x7fffffff ? 1 : bow->q_seq + 1) : 0;
    bh_ctx* ctx = bow->ctx;
    hipStream_t st = ctx->stream;
    const size_t hlds = INV_HEAVY_LDS;
    const inv_slot* slots = dev_slot ? dev_slot : (const inv_slot*)bow->inv_slots.p;      // dev_slot: one device-resident query, nothing read back here
    bh_time_begin(ctx, BH_K_BOW_INV);
    hipLaunchKernelGGL(k_inv_gather, dim3((max_nq + 3) / 4, G), dim3(256), 0, st, slots, (const long long*)bow->inv_start.p, (const uint32_t*)bow->inv_entry.p,
                       (const double*)bow->inv_val.p, bow->bcap, nq1);
    hipLaunchKernelGGL(k_inv_finish, dim3((unsigned)((bow->inv_nb + INV_FIN_BK - 1) / INV_FIN_BK), G), dim3(INV_FIN_T), hlds, st, slots, bow->bcap, bow->n_entries, keep,
                       bow->inv_nb, k1, seq);
    bh_time_end(ctx);
    BH_HIP(ctx, hipGetLastError());
    if (dev_slot) return BH_OK;
    if (seq) {                                                 // bounded by time; a kernel fault never delivers the tag and the stream reports it below
        const volatile int* tag = reinterpret_cast<const volatile int*>(bow->hq + INV_PIN_R) + INV_PIN_SEQ;
        const auto t0 = std::chrono::steady_clock::now();
        for (long spin = 0; ; spin++) {
            if (__atomic_load_n(const_cast<const int*>(tag), __ATOMIC_ACQUIRE) == seq) return BH_OK;
            if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
    }
    BH_HIP(ctx, hipStreamSynchronize(st));
    return BH_OK;
}
// results of a device-resident query: out of the slot's block into the caller's device arrays, if the inverted file could answer
__global__ void k_inv_export(const char* __restrict__ hout, int* __restrict__ ids, double* __restrict__ scores, int* __restrict__ n_results, int* __restrict__ answered)
{
    const int* hi = reinterpret_cast<const int*>(hout);
    const int n = hi[0], fb = hi[1], i = threadIdx.x;
    if (!fb && i < n) { scores[i] = reinterpret_cast<const double*>(hout + 8)[i]; ids[i] = reinterpret_cast<const int*>(hout + 8 + 8 * INV_KMAX)[i]; }
    if (i == 0) { if (!fb) n_results[0] = n; answered[0] = fb ? 0 : 1; }
}
static void inv_fill_slot(bh_bow* bow, int q, const uint32_t* q_words, const double* q_vals, int nq, int k)
{
    char* hp = bow->hq + (size_t)INV_PIN_BYTES * q;
    reinterpret_cast<int*>(hp)[0] = nq; reinterpret_cast<int*>(hp)[1] = k;
    memcpy(hp + INV_PIN_W, q_words, sizeof(uint32_t) * (size_t)nq); memcpy(hp + INV_PIN_V, q_vals, sizeof(double) * (size_t)nq);
}
static int inv_read_slot(bh_bow* bow, int q, int* ids, double* scores, int* n_results)      // 1: answered, 0: the scan must
{
    const char* hr = bow->hq + (size_t)INV_PIN_BYTES * q + INV_PIN_R;
    const int nres = reinterpret_cast<const int*>(hr)[0], fallback = reinterpret_cast<const int*>(hr)[1];
    if (fallback) return 0;                                    // something did not fit this query
    *n_results = nres;
    memcpy(scores, hr + 8, sizeof(double) * (size_t)nres);
    memcpy(ids, hr + 8 + 8 * INV_KMAX, sizeof(int) * (size_t)nres);
    return 1;
}

// The inverted file after the database changed.  The reference alternates add and query per keyframe (covisibility_graph.add_kf ->
// dbow.add, then the loop closer queries from bow_queue): rebuilding the file - an O(database) count / scan / scatter, 1 - 2 ms at configs[3] -
// for every such query would cost more than the streaming scan it replaces (0.29 ms).  So while the file is stale the scan answers, and the
// file is rebuilt once INV_DIRTY_QUERIES queries in a row have found the database unchanged (a phase of pure querying pays one rebuild).
#define INV_DIRTY_QUERIES 3
static int inv_ensure(bh_bow* bow, int* ready)
{
    *ready = 1;
    if (bow->inv_ready) return BH_OK;
    if (bow->dirty_queries < INV_DIRTY_QUERIES) { bow->dirty_queries++; *ready = 0; return BH_OK; }
    BH_TRY(inv_build(bow));
    bow->dirty_queries = 0;
    return BH_OK;
}

// The inverted-file query, host arrays in and out.  Returns BH_OK with *used = 1 when it answered, *used = 0 when the scan must.
static int inv_query(bh_bow* bow, const uint32_t* q_words, const double* q_vals, int nq, const int* subset, int n_subset, int max_results,
                     int* ids, double* scores, int* n_results, int* used)
{
    bh_ctx* ctx = bow->ctx;
    hipStream_t st = ctx->stream;
    *used = 0;
    if (max_results > INV_KMAX) return BH_OK;
    int ready = 0;
    BH_TRY(inv_ensure(bow, &ready));
    if (!ready || !bow->inv_usable || !bow->hq) return BH_OK;
    inv_fill_slot(bow, 0, q_words, q_vals, nq, max_results);  // pinned, device-visible block: the query in, the results out - no staging copies, one synchronisation
    const uint8_t* keep = nullptr;
    if (n_subset > 0) {
        BH_TRY(bh_reserve(ctx, &ctx->st_g, (size_t)bow->n_entries + sizeof(int) * (size_t)n_subset + 16));
        uint8_t* flag = (uint8_t*)ctx->st_g.p; int* d_sub = (int*)(flag + (((size_t)bow->n_entries + 15) & ~(size_t)15));
        BH_HIP(ctx, hipMemsetAsync(flag, 0, (size_t)bow->n_entries, st));
        BH_HIP(ctx, hipMemcpyAsync(d_sub, subset, sizeof(int) * (size_t)n_subset, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_bow_flag_set, dim3((n_subset + 255) / 256), dim3(256), 0, st, (const int*)d_sub, n_subset, bow->n_entries, flag);
        keep = flag;
    }
    BH_TRY(inv_launch(bow, 1, nq, keep, nullptr, nq, max_results));
    *used = inv_read_slot(bow, 0, ids, scores, n_results);
    return BH_OK;
}

extern "C" int bh_bow_set_mode(bh_bow* bow, int mode)
{
    if (!bow || (mode != 0 && mode != 1)) return BH_ERR_INVALID;
    bow->mode = mode;
    return BH_OK;
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
    if (mem == BH_MEM_HOST && bow->mode == 0) {                 // DBoW3's own structure first; the scan when it cannot answer
        for (int i = 1; i < nq; i++) if (q_words[i] <= q_words[i - 1]) return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_query_l1: query words must be strictly ascending");
        int used = 0;
        BH_TRY(inv_query(bow, q_words, q_vals, nq, subset, n_subset, max_results, ids, scores, n_results, &used));
        if (used) return BH_OK;
    }
    const int* skip = nullptr;
    if (mem == BH_MEM_DEVICE && bow->mode == 0 && n_subset == 0 && max_results <= INV_KMAX) {
        // A device-resident query through the inverted file too (a relocaliser that keeps its BoW vectors in HBM): the four kernels run on a
        // slot whose query arrays are the caller's, k_inv_export moves the results into the caller's arrays and raises `answered`; the
        // scan that follows is enqueued all the same and returns at once when the flag is up - no host decision, no synchronisation.
        int ready = 0;
        BH_TRY(inv_ensure(bow, &ready));
        if (ready && bow->inv_usable && bow->hq) {
            const size_t off_hdr = 256, off_out = 512, off_flag = 512 + 1024;
            BH_TRY(bh_reserve(ctx, &bow->dq, 2048));
            inv_slot S; memcpy(&S, bow->slot0_host, sizeof(S));
            char* dq = (char*)bow->dq.p;
            S.hdr = (const int*)(dq + off_hdr); S.qw = q_words; S.qv = q_vals; S.hout = dq + off_out;
            const int hdr[2] = { nq, max_results };
            BH_HIP(ctx, hipMemcpyAsync(dq, &S, sizeof(S), hipMemcpyHostToDevice, st));
            BH_HIP(ctx, hipMemcpyAsync(dq + off_hdr, hdr, sizeof(hdr), hipMemcpyHostToDevice, st));
            BH_TRY(inv_launch(bow, 1, nq, nullptr, (const inv_slot*)dq));
            hipLaunchKernelGGL(k_inv_export, dim3(1), dim3(64), 0, st, (const char*)(dq + off_out), ids, scores, n_results, (int*)(dq + off_flag));
            skip = (const int*)(dq + off_flag);
        }
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
    const size_t lds = BOW_SCAN_LDS;
    int grid = 256;                                            // one persistent workgroup per CU (LDS-limited to 1/CU anyway)
    const long long need = (bow->n_entries + BOW_T / 64 - 1) / (BOW_T / 64);
    if (need < grid) grid = (int)need;
    bh_time_begin(ctx, BH_K_BOW_SCAN);
    hipLaunchKernelGGL(k_bow_scan, dim3(grid), dim3(BOW_T), lds, st, (const long long*)bow->pstart.p, (const uint32_t*)bow->words.p,
                       (const double*)bow->vals.p, bow->n_entries, d_qw, d_qv, nq, (double*)bow->acc.p, skip);
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
        hipLaunchKernelGGL(k_bow_topk_slices, dim3(ns), dim3(1024), 0, st, (const double*)bow->acc.p, bow->n_entries, max_results, cv, ci, skip);
        hipLaunchKernelGGL(k_bow_topk, dim3(1), dim3(1024), 0, st, cv, (long long)ns * max_results, max_results, (const long long*)ci, d_ids, d_sc, d_n, skip);
    } else
        hipLaunchKernelGGL(k_bow_topk, dim3(1), dim3(1024), 0, st, (double*)bow->acc.p, bow->n_entries, max_results, (const long long*)nullptr, d_ids, d_sc, d_n, skip);
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
    bow->inv_ready = 0; bow->dirty_queries = 0;
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
    bow->inv_ready = 0; bow->dirty_queries = 0;
    return BH_OK;
}

// Q queries at once through the inverted file: up to INV_QSLOTS per round of four launches (blockIdx.y = query slot, every slot with its
// own scratch), so the ~0.1 ms of launch / dependency latency a single query pays is shared.  A loop closer or relocaliser that works
// through a queue of keyframes (slam/loop_closing.py:29-30 takes them one by one from bow_queue) can hand them over together.
extern "C" int bh_bow_query_l1_batch(bh_bow* bow, const uint32_t* q_words, const double* q_vals, const int* nq, int Q, int q_stride, int max_results,
                                     int* ids, double* scores, int* n_results)
{
    if (!bow) return BH_ERR_INVALID;
    bh_ctx* ctx = bow->ctx;
    if (!q_words || !q_vals || !nq || !ids || !scores || !n_results || Q < 0 || q_stride < 1 || max_results < 1 || max_results > 1024)
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_query_l1_batch: bad argument");
    for (int q = 0; q < Q; q++) {
        if (nq[q] < 0 || nq[q] > q_stride || nq[q] > BOW_QMAX) return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_query_l1_batch: query %d has %d words (stride %d, limit %d)", q, nq[q], q_stride, BOW_QMAX);
        const uint32_t* w = q_words + (size_t)q * q_stride;
        for (int i = 1; i < nq[q]; i++) if (w[i] <= w[i - 1]) return bh_set_err(ctx, BH_ERR_INVALID, "bh_bow_query_l1: query words must be strictly ascending");
    }
    BH_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<char> done(Q, 0);
    if (bow->mode == 0 && max_results <= INV_KMAX && bow->n_entries > 0 && Q > 0) {
        int ready = 0;
        if (!bow->inv_ready && Q >= 2 * INV_DIRTY_QUERIES) bow->dirty_queries = INV_DIRTY_QUERIES;      // a queue of queries is a querying phase by itself
        BH_TRY(inv_ensure(bow, &ready));
        if (ready && bow->inv_usable && bow->inv_nslots < INV_QSLOTS) BH_TRY(inv_alloc_slots(bow, INV_QSLOTS));
        if (ready && bow->inv_usable && bow->hq) {
            for (int g0 = 0; g0 < Q; g0 += INV_QSLOTS) {
                const int G = std::min(INV_QSLOTS, Q - g0);
                int max_nq = 1;
                for (int s = 0; s < G; s++) { const int q = g0 + s; inv_fill_slot(bow, s, q_words + (size_t)q * q_stride, q_vals + (size_t)q * q_stride, nq[q], max_results); max_nq = std::max(max_nq, nq[q]); }
                BH_TRY(inv_launch(bow, G, max_nq, nullptr));
                for (int s = 0; s < G; s++) { const int q = g0 + s; if (nq[q] > 0) done[q] = (char)inv_read_slot(bow, s, ids + (size_t)q * max_results, scores + (size_t)q * max_results, n_results + q); }
            }
        }
    }
    for (int q = 0; q < Q; q++)                                 // whatever the inverted file could not answer (or scan mode): one by one
        if (!done[q]) BH_TRY(bow_query_impl(bow, q_words + (size_t)q * q_stride, q_vals + (size_t)q * q_stride, nq[q], nullptr, 0, max_results,
                                            ids + (size_t)q * max_results, scores + (size_t)q * max_results, n_results + q, BH_MEM_HOST));
    return BH_OK;
}

static int bow_set_func_attrs(bh_ctx* ctx)
{
    BH_HIP(ctx, hipFuncSetAttribute((const void*)k_bow_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BOW_SCAN_LDS));
    BH_HIP(ctx, hipFuncSetAttribute((const void*)k_inv_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)INV_HEAVY_LDS));
    return BH_OK;
}
