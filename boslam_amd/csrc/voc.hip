// voc.hip — vocabulary-tree quantisation of ORB descriptors into a BoW vector (gfx950).
//
// Replaces DBoW3 Vocabulary::transform as reached from slam/bow_db.py:26 (Database.add(frame.des)) and :39
// (Database.query(frame.des, ...)) through docker/pydbow3/src/dbow3.cpp:138-152.  SURVEY.md §8(f) rank 1: it makes the BoW
// path end-to-end from descriptors.
//
// One 1024-lane workgroup per descriptor set (frame).  Phase 1: one lane per descriptor walks the tree (children of a
// node are contiguous 32-B descriptors: k xor+popcount distances per level, first minimum in stored order).  Phase 2: the
// (word, descriptor index) keys are bitonic-sorted in LDS.  Phase 3: run heads sum their run's weights sequentially in
// descriptor order — the order DBoW3's addWeight uses — and are compacted by a block scan.  Phase 4: L1 normalisation with
// the norm summed in ascending word order (one lane), so the result is bit-identical to the CPU restatement.
#include "bh_internal.h"
#include <math.h>
#include <vector>

#define VOC_T   1024
#define VOC_MAX 4096              // descriptors per set

struct bh_voc {
    bh_ctx* ctx = nullptr;
    int n_nodes = 0;
    bh_buf child_start, child_idx, node_desc, node_weight, node_word;
};

__global__ __launch_bounds__(VOC_T) void k_voc_transform(const int* __restrict__ child_start, const int* __restrict__ child_idx,
                                                         const uint4* __restrict__ node_desc, const double* __restrict__ node_weight,
                                                         const int* __restrict__ node_word, const uint8_t* __restrict__ desc,
                                                         const int* __restrict__ n_arr, int stride, uint32_t* __restrict__ words,
                                                         double* __restrict__ vals, int* __restrict__ n_words)
{
    __shared__ uint32_t s_key[VOC_MAX];
    __shared__ double   s_w[VOC_MAX];
    __shared__ int      s_scan[VOC_T / 64];
    __shared__ double   s_norm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t b = blockIdx.x;
    const int n = min(n_arr[b], VOC_MAX);
    int npad = 64; while (npad < n) npad <<= 1;
    // phase 1: tree walk
    for (int r = tid; r < npad; r += VOC_T) {
        uint32_t key = 0xffffffffu;
        double w = 0.0;
        if (r < n) {
            const uint4* d = reinterpret_cast<const uint4*>(desc + (b * (size_t)stride + r) * 32);
            const uint4 d0 = d[0], d1 = d[1];
            int node = 0;
            while (true) {
                const int c0 = child_start[node], c1 = child_start[node + 1];
                if (c1 <= c0) break;
                int best = -1, best_d = 1 << 30;
                for (int c = c0; c < c1; c++) {
                    const int id = child_idx[c];
                    const uint4 a = node_desc[2 * (size_t)id], e = node_desc[2 * (size_t)id + 1];
                    const int dist = __popc(a.x ^ d0.x) + __popc(a.y ^ d0.y) + __popc(a.z ^ d0.z) + __popc(a.w ^ d0.w) +
                                     __popc(e.x ^ d1.x) + __popc(e.y ^ d1.y) + __popc(e.z ^ d1.z) + __popc(e.w ^ d1.w);
                    if (dist < best_d) { best_d = dist; best = id; }
                }
                node = best;
            }
            w = node_weight[node];
            if (w > 0) key = ((uint32_t)node_word[node] << 12) | (uint32_t)r;        // word < 2^20, r < 4096
        }
        s_key[r] = key; s_w[r] = w;
    }
    __syncthreads();
    // phase 2: bitonic sort of the keys (ascending); invalid keys (0xffffffff) sink to the end
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npad; i += VOC_T) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = s_key[i], c = s_key[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { s_key[i] = c; s_key[l] = a; }
                }
            }
            __syncthreads();
        }
    // phase 3: run heads accumulate their run in descriptor order; heads are compacted in ascending word order
    int base = 0;
    uint32_t* ow = words + b * (size_t)stride; double* ov = vals + b * (size_t)stride;
    for (int c0 = 0; c0 < npad; c0 += VOC_T) {
        const int i = c0 + tid;
        bool head = false; uint32_t word = 0; double acc = 0;
        if (i < npad) {
            const uint32_t key = s_key[i];
            if (key != 0xffffffffu) {
                word = key >> 12;
                head = i == 0 || (s_key[i - 1] >> 12) != word;
                if (head) for (int j = i; j < npad && s_key[j] != 0xffffffffu && (s_key[j] >> 12) == word; j++) acc += s_w[s_key[j] & 4095u];
            }
        }
        const unsigned long long m = __ballot(head);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) s_scan[wv] = __popcll(m);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < VOC_T / 64; w++) { const int c = s_scan[w]; if (w < wv) wbase += c; tot += c; }
        if (head) { ow[base + wbase + pre] = word; ov[base + wbase + pre] = acc; }
        base += tot;
    }
    __syncthreads();
    // phase 4: L1 normalisation, norm summed in ascending word order
    if (tid == 0) { double nrm = 0; for (int i = 0; i < base; i++) nrm += fabs(ov[i]); s_norm = nrm; n_words[b] = base; }
    __syncthreads();
    const double nrm = s_norm;
    if (nrm > 0) for (int i = tid; i < base; i += VOC_T) ov[i] = ov[i] / nrm;
}

extern "C" int bh_voc_create(bh_ctx* ctx, int n_nodes, const int* child_start, const int* child_idx, const uint8_t* node_desc,
                             const double* node_weight, const int* node_word, bh_voc** out)
{
    if (!ctx || !out) return BH_ERR_INVALID;
    if (n_nodes < 1 || !child_start || !child_idx || !node_desc || !node_weight || !node_word)
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_voc_create: bad argument");
    const int nchild = child_start[n_nodes];
    if (child_start[0] != 0 || nchild < 0) return bh_set_err(ctx, BH_ERR_INVALID, "bh_voc_create: child_start must start at 0");
    for (int i = 0; i < n_nodes; i++) if (child_start[i + 1] < child_start[i]) return bh_set_err(ctx, BH_ERR_INVALID, "bh_voc_create: child_start not ascending");
    for (int c = 0; c < nchild; c++) if (child_idx[c] <= 0 || child_idx[c] >= n_nodes) return bh_set_err(ctx, BH_ERR_INVALID, "bh_voc_create: child index out of range");
    for (int i = 0; i < n_nodes; i++) if (child_start[i + 1] == child_start[i] && (node_word[i] < 0 || node_word[i] >= (1 << 20)))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_voc_create: leaf word id outside [0, 2^20)");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    bh_voc* v = new bh_voc(); v->ctx = ctx; v->n_nodes = n_nodes;
    hipStream_t st = ctx->stream;
    int rc = BH_OK;
    auto up = [&](bh_buf* b, const void* src, size_t bytes) {
        if (rc != BH_OK) return;
        rc = bh_reserve(ctx, b, bytes + 64);
        if (rc == BH_OK && bytes && hipMemcpyAsync(b->p, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) rc = bh_set_err(ctx, BH_ERR_HIP, "bh_voc_create: upload failed");
    };
    up(&v->child_start, child_start, sizeof(int) * (n_nodes + 1)); up(&v->child_idx, child_idx, sizeof(int) * nchild);
    up(&v->node_desc, node_desc, (size_t)32 * n_nodes); up(&v->node_weight, node_weight, sizeof(double) * n_nodes);
    up(&v->node_word, node_word, sizeof(int) * n_nodes);
    if (rc == BH_OK && hipStreamSynchronize(st) != hipSuccess) rc = bh_set_err(ctx, BH_ERR_HIP, "bh_voc_create: sync failed");
    if (rc != BH_OK) { delete v; return rc; }
    *out = v;
    return BH_OK;
}

extern "C" void bh_voc_destroy(bh_voc* v)
{
    if (!v) return;
    (void)hipSetDevice(v->ctx->device);
    (void)hipStreamSynchronize(v->ctx->stream);
    bh_buf* all[] = { &v->child_start, &v->child_idx, &v->node_desc, &v->node_weight, &v->node_word };
    for (bh_buf* b : all) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    delete v;
}

extern "C" int bh_voc_transform(bh_voc* voc, const uint8_t* desc, const int* n, int B, int stride, uint32_t* words, double* vals,
                                int* n_words, int mem)
{
    if (!voc) return BH_ERR_INVALID;
    bh_ctx* ctx = voc->ctx;
    if (!desc || !n || !words || !vals || !n_words || B < 0 || stride < 1 || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_voc_transform: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t ne = (size_t)B * stride;
    const uint8_t* d_desc = desc; const int* d_n = n; uint32_t* d_w = words; double* d_v = vals; int* d_nw = n_words;
    if (mem == BH_MEM_HOST) {
        for (int b = 0; b < B; b++) if (n[b] < 0 || n[b] > stride || n[b] > VOC_MAX) return bh_set_err(ctx, BH_ERR_INVALID, "bh_voc_transform: n[b] outside [0, min(stride, 4096)]");
        BH_TRY(bh_reserve(ctx, &ctx->st_a, ne * 32)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(int) * 2 * (size_t)B));
        BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(uint32_t) * ne)); BH_TRY(bh_reserve(ctx, &ctx->st_d, sizeof(double) * ne));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, desc, ne * 32, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, n, sizeof(int) * B, hipMemcpyHostToDevice, st));
        d_desc = (const uint8_t*)ctx->st_a.p; d_n = (const int*)ctx->st_b.p; d_nw = (int*)ctx->st_b.p + B;
        d_w = (uint32_t*)ctx->st_c.p; d_v = (double*)ctx->st_d.p;
    }
    BH_LAUNCH(ctx, BH_K_VOC, k_voc_transform, dim3(B), dim3(VOC_T), st, (const int*)voc->child_start.p, (const int*)voc->child_idx.p,
              (const uint4*)voc->node_desc.p, (const double*)voc->node_weight.p, (const int*)voc->node_word.p, d_desc, d_n, stride,
              d_w, d_v, d_nw);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(words, d_w, sizeof(uint32_t) * ne, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(vals, d_v, sizeof(double) * ne, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_words, d_nw, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}
