// loop.hip - the two geometric stages behind bh_align_se3_ransac in the loop-closure thread: GlobalMapManager._loop_fusion and
// GlobalMapManager._essential_graph_optimization (slam/loop_closing.py:21-29).
//
// Reference: both bodies are `pass`: PARITY UNPINNED BY CONSTRUCTION - there is no reference behaviour, the
// algorithms are specified by this project and checked against CPU restatements of that specification
// (oracle.loop_fuse_pairs, oracle.pose_graph_optimize).
//
// bh_loop_fuse_pairs ("finds duplicated mappoints and fuses them together", :21-26): the map points seen from the loop's two ends - side A
// around the current keyframe, side B around the loop candidate - with (R, t) of _similiarity_transformation (a ~ R b + t).  A pair
// (i, j) is a duplicate when the cross-check Hamming matcher pairs feat_a[i] with feat_b[j] (cv2.BFMatcher(NORM_HAMMING, crossCheck=True), the
// matcher every other stage of the reference uses), their distance is below d_max (config.d_hamming_max, `<` as slam/tracking.py:57) and
// |pt_a[i] - (R pt_b[j] + t)| <= radius.  The survivor of a pair is the point with more observations (A on a tie): keep_a.  Pairs come
// out in ascending i.  The graph surgery itself (re-pointing edges_kf2mps / edges_mp2kfs) stays with the host's graph.
#include "bh_internal.h"

__global__ __launch_bounds__(256) void k_fuse_filter(const int* __restrict__ rq, const int* __restrict__ rt, const int* __restrict__ rd, const int* __restrict__ rn, int raw_cap,
                                                     const double* __restrict__ pt_a, const double* __restrict__ pt_b, const int* __restrict__ nobs_a,
                                                     const int* __restrict__ nobs_b, const double* __restrict__ Rt /* R[9] t[3] */, int d_max, double radius2,
                                                     int* __restrict__ pair_a, int* __restrict__ pair_b, uint8_t* __restrict__ keep_a, int* __restrict__ n_pairs, int cap)
{
    __shared__ int s_wsum[4], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(rn[0], raw_cap);
    const double R0 = Rt[0], R1 = Rt[1], R2 = Rt[2], R3 = Rt[3], R4 = Rt[4], R5 = Rt[5], R6 = Rt[6], R7 = Rt[7], R8 = Rt[8], t0 = Rt[9], t1 = Rt[10], t2 = Rt[11];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        int a = 0, b = 0;
        bool keep = false;
        if (i < n) {
            a = rq[i]; b = rt[i];
            if (rd[i] < d_max) {
                const double bx = pt_b[3 * (size_t)b], by = pt_b[3 * (size_t)b + 1], bz = pt_b[3 * (size_t)b + 2];
                const double dx = pt_a[3 * (size_t)a] - (((R0 * bx) + (R1 * by)) + (R2 * bz) + t0);
                const double dy = pt_a[3 * (size_t)a + 1] - (((R3 * bx) + (R4 * by)) + (R5 * bz) + t1);
                const double dz = pt_a[3 * (size_t)a + 2] - (((R6 * bx) + (R7 * by)) + (R8 * bz) + t2);
                keep = ((dx * dx) + (dy * dy)) + (dz * dz) <= radius2;
            }
        }
        const unsigned long long bm = __ballot(keep);
        const int before = __popcll(bm & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[wv] = __popcll(bm);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; w++) off += s_wsum[w];
        const int pos = off + before;
        if (keep && pos < cap) { pair_a[pos] = a; pair_b[pos] = b; keep_a[pos] = nobs_a[a] >= nobs_b[b] ? 1 : 0; }
        __syncthreads();
        if (tid == 0) s_base += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        __syncthreads();
    }
    if (tid == 0) n_pairs[0] = min(s_base, cap);
}

extern "C" int bh_loop_fuse_pairs(bh_ctx* ctx, const uint8_t* feat_a, const double* pt_a, const int* nobs_a, int Na,
                                  const uint8_t* feat_b, const double* pt_b, const int* nobs_b, int Nb,
                                  const double R[9], const double t[3], int d_max, double radius,
                                  int* pair_a, int* pair_b, uint8_t* keep_a, int* n_pairs, int cap, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!feat_a || !pt_a || !nobs_a || !feat_b || !pt_b || !nobs_b || !R || !t || !pair_a || !pair_b || !keep_a || !n_pairs || Na < 0 || Nb < 0 || cap < 1 ||
        !(radius >= 0) || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_loop_fuse_pairs: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t na = (size_t)(Na > 0 ? Na : 1), nb = (size_t)(Nb > 0 ? Nb : 1), raw = na < nb ? na : nb;
    const size_t w_b = al(sizeof(int) * 2) + 3 * al(sizeof(int) * raw) + al(sizeof(int)) + al(sizeof(double) * 12);
    const size_t in_b = mem == BH_MEM_HOST ? al(na * 32) + al(nb * 32) + al(sizeof(double) * 3 * na) + al(sizeof(double) * 3 * nb) + al(sizeof(int) * na) + al(sizeof(int) * nb) +
                                             2 * al(sizeof(int) * (size_t)cap) + al((size_t)cap) + al(sizeof(int)) : 0;
    BH_TRY(bh_reserve(ctx, &ctx->st_h, w_b + in_b));
    char* p = (char*)ctx->st_h.p;
    int* d_cnt = (int*)p; p += al(sizeof(int) * 2);                // nq, nt of the one matcher pair
    int* rq = (int*)p; p += al(sizeof(int) * raw); int* rt = (int*)p; p += al(sizeof(int) * raw); int* rd = (int*)p; p += al(sizeof(int) * raw);
    int* rn = (int*)p; p += al(sizeof(int));
    double* d_Rt = (double*)p; p += al(sizeof(double) * 12);
    const int hc[2] = { Na, Nb };
    double hRt[12]; memcpy(hRt, R, sizeof(double) * 9); memcpy(hRt + 9, t, sizeof(double) * 3);
    BH_HIP(ctx, hipMemcpyAsync(d_cnt, hc, sizeof(hc), hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(d_Rt, hRt, sizeof(hRt), hipMemcpyHostToDevice, st));
    const uint8_t *d_fa = feat_a, *d_fb = feat_b; const double *d_pa = pt_a, *d_pb = pt_b; const int *d_na = nobs_a, *d_nb = nobs_b;
    int *d_qa = pair_a, *d_qb = pair_b, *d_np = n_pairs; uint8_t* d_ka = keep_a;
    if (mem == BH_MEM_HOST) {
        auto up = [&](const void* src, size_t bytes, size_t room) -> char* { char* d = p; p += al(room); return (bytes == 0 || hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st) == hipSuccess) ? d : nullptr; };
        char* q;
#define LF_UP(dst, type, src, bytes, room) if (!(q = up(src, bytes, room))) return bh_set_err(ctx, BH_ERR_HIP, "bh_loop_fuse_pairs: upload failed"); dst = (type)q
        LF_UP(d_fa, const uint8_t*, feat_a, (size_t)Na * 32, na * 32); LF_UP(d_fb, const uint8_t*, feat_b, (size_t)Nb * 32, nb * 32);
        LF_UP(d_pa, const double*, pt_a, sizeof(double) * 3 * (size_t)Na, sizeof(double) * 3 * na); LF_UP(d_pb, const double*, pt_b, sizeof(double) * 3 * (size_t)Nb, sizeof(double) * 3 * nb);
        LF_UP(d_na, const int*, nobs_a, sizeof(int) * (size_t)Na, sizeof(int) * na); LF_UP(d_nb, const int*, nobs_b, sizeof(int) * (size_t)Nb, sizeof(int) * nb);
#undef LF_UP
        d_qa = (int*)p; p += al(sizeof(int) * (size_t)cap); d_qb = (int*)p; p += al(sizeof(int) * (size_t)cap); d_ka = (uint8_t*)p; p += al((size_t)cap); d_np = (int*)p;
    }
    BH_TRY(bh_match_hamming_crosscheck(ctx, d_fa, d_cnt, (int)na, d_fb, d_cnt + 1, (int)nb, 1, rq, rt, rd, rn, (int)raw, BH_MEM_DEVICE));
    BH_LAUNCH(ctx, BH_K_MATCH, k_fuse_filter, dim3(1), dim3(256), st, (const int*)rq, (const int*)rt, (const int*)rd, (const int*)rn, (int)raw, d_pa, d_pb, d_na, d_nb,
              (const double*)d_Rt, d_max, radius * radius, d_qa, d_qb, d_ka, d_np, cap);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(n_pairs, d_np, sizeof(int), hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(pair_a, d_qa, sizeof(int) * (size_t)cap, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(pair_b, d_qb, sizeof(int) * (size_t)cap, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(keep_a, d_ka, (size_t)cap, hipMemcpyDeviceToHost, st));
        return bh_ctx_status(ctx);
    }
    return BH_OK;
}
