// ctx.hip — context, error reporting and workspace management of libboslam_hip.so.
#include "bh_internal.h"
#include <stdarg.h>

int bh_set_err(bh_ctx* ctx, int code, const char* fmt, ...)
{
    if (ctx) {
        va_list ap; va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

int bh_reserve(bh_ctx* ctx, bh_buf* b, size_t bytes)
{
    if (bytes <= b->cap) return BH_OK;
    // a grow frees the old block: make sure nothing enqueued still uses it
    if (b->p) {
        BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int p = 0; p < BH_MAX_PIPES - 1; p++)                // chunk pipelines of an earlier call that returned early (error path) may still run
            if (ctx->aux_stream[p]) BH_HIP(ctx, hipStreamSynchronize(ctx->aux_stream[p]));
        BH_HIP(ctx, hipFree(b->p)); b->p = nullptr; b->cap = 0;
    }
    size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b->p, want);
    if (e != hipSuccess) { b->p = nullptr; return bh_set_err(ctx, BH_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    b->cap = want;
    return BH_OK;
}

int bh_pin_reserve(bh_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->pin_cap) return BH_OK;
    if (ctx->pin) { BH_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->pin); ctx->pin = nullptr; ctx->pin_cap = 0; }
    const size_t want = (bytes + (bytes >> 1) + 4095) & ~(size_t)4095;
    hipError_t e = hipHostMalloc(&ctx->pin, want, hipHostMallocDefault);
    if (e != hipSuccess) { ctx->pin = nullptr; return bh_set_err(ctx, BH_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    ctx->pin_cap = want;
    return BH_OK;
}

// The device status word is STICKY: kernels OR capacity flags into it, nothing clears it except bh_ctx_status after reading it, so
// a BH_MEM_DEVICE caller that checks once after many enqueued steps sees every flag raised in between.
int bh_status_ready(bh_ctx* ctx)
{
    if (ctx->status.p) return BH_OK;
    BH_TRY(bh_reserve(ctx, &ctx->status, 256));
    BH_HIP(ctx, hipMemsetAsync(ctx->status.p, 0, 256, ctx->stream));
    return BH_OK;
}

static int drain_timing(bh_ctx* ctx);
static void release(bh_buf* b) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }

extern "C" const char* bh_version(void) { return "boslam_hip 0.1 (gfx950)"; }

extern "C" int bh_ctx_create(int device, bh_ctx** out)
{
    if (!out) return BH_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return BH_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return BH_ERR_HIP;
    bh_ctx* ctx = new bh_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return BH_ERR_HIP; }
    ctx->stream = ctx->own_stream;
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int p = 0; p < BH_MAX_PIPES - 1 && ok; p++)
        ok = hipStreamCreateWithFlags(&ctx->aux_stream[p], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->ev_join[p], hipEventDisableTiming) == hipSuccess;
    if (!ok) { bh_ctx_destroy(ctx); return BH_ERR_HIP; }
    *out = ctx;
    return BH_OK;
}

extern "C" void bh_ctx_destroy(bh_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int p = 0; p < BH_MAX_PIPES - 1; p++) if (ctx->aux_stream[p]) (void)hipStreamSynchronize(ctx->aux_stream[p]);
    for (int p = 0; p < BH_MAX_PIPES; p++) { bh_orb_ws& W = ctx->ows[p]; bh_buf* wsb[] = { &W.pyr, &W.mpyr, &W.clist, &W.blur, &W.hist, &W.sel_xy, &W.sel_resp, &W.sel_cnt }; for (bh_buf* b : wsb) release(b); }
    bh_buf* all[] = { &ctx->tab, &ctx->rtab, &ctx->status, &ctx->tr_w, &ctx->tr_in, &ctx->tl_w, &ctx->tl_in,
                      &ctx->st_img, &ctx->st_mask, &ctx->st_depth, &ctx->st_kps, &ctx->st_desc, &ctx->st_nkp,
                      &ctx->st_a, &ctx->st_b, &ctx->st_c, &ctx->st_d, &ctx->st_e, &ctx->st_f, &ctx->st_g, &ctx->st_h, &ctx->pnp_p0 };
    for (bh_buf* b : all) release(b);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    (void)drain_timing(ctx);
    for (int i = 0; i < ctx->n_epool; i++) (void)hipEventDestroy(ctx->epool[i]);
    free(ctx->epool); free(ctx->trecs);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int p = 0; p < BH_MAX_PIPES - 1; p++) { if (ctx->ev_join[p]) (void)hipEventDestroy(ctx->ev_join[p]); if (ctx->aux_stream[p]) (void)hipStreamDestroy(ctx->aux_stream[p]); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" const char* bh_last_error(const bh_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int bh_ctx_set_stream(bh_ctx* ctx, void* hip_stream)
{
    if (!ctx) return BH_ERR_INVALID;
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return BH_OK;
}

extern "C" int bh_ctx_synchronize(bh_ctx* ctx)
{
    if (!ctx) return BH_ERR_INVALID;
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BH_OK;
}

extern "C" int bh_ctx_copy(bh_ctx* ctx, void* dst, const void* src, size_t bytes, int to_device)
{
    if (!ctx || (bytes && (!dst || !src))) return BH_ERR_INVALID;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) BH_HIP(ctx, hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
    return BH_OK;
}

extern "C" int bh_ctx_status(bh_ctx* ctx)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!ctx->status.p) return BH_OK;
    uint32_t st = 0;
    BH_HIP(ctx, hipMemcpyAsync(&st, ctx->status.p, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    BH_HIP(ctx, hipMemsetAsync(ctx->status.p, 0, 256, ctx->stream));           // read-and-clear, in stream order
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st & BH_ST_KP_OVERFLOW) return bh_set_err(ctx, BH_ERR_CAPACITY, "more keypoints than max_kp (ties at the retainBest cut); output truncated");
    if (st & 4u) return bh_set_err(ctx, BH_ERR_INVALID, "matcher: descriptor count above the supported maximum (nq <= 8192, nt <= 2^20)");
    if (st & 8u) return bh_set_err(ctx, BH_ERR_CAPACITY, "matcher: more matches than out_stride");
    if (st & BH_ST_CAND_OVERFLOW) return bh_set_err(ctx, BH_ERR_CAPACITY, "FAST candidate list overflow in k_select; output truncated");
    return BH_OK;
}

// ------------------------------------------------------------------------------------------------ per-kernel timing
static const char* k_names[BH_K_COUNT] = { "k_gray_mask", "k_pyr_down", "k_fast", "k_blur", "k_select", "k_describe",
                                           "k_match", "k_kp_backproject", "k_pose_ba", "k_ba_linearize", "k_ba_schur",
                                           "k_ba_solve", "k_ba_update", "k_ba_other", "k_gate", "k_bow_scan", "k_pnp_ransac", "k_voc_transform", "k_bow_inverted", "k_align_se3" };

static hipEvent_t take_event(bh_ctx* ctx)
{
    if (ctx->n_epool > 0) return ctx->epool[--ctx->n_epool];
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static void give_event(bh_ctx* ctx, hipEvent_t e)
{
    if (ctx->n_epool == ctx->cap_epool) {
        ctx->cap_epool = ctx->cap_epool ? ctx->cap_epool * 2 : 256;
        ctx->epool = (hipEvent_t*)realloc(ctx->epool, sizeof(hipEvent_t) * ctx->cap_epool);
    }
    ctx->epool[ctx->n_epool++] = e;
}

void bh_time_begin(bh_ctx* ctx, int kid)
{
    if (!ctx->timing_on) return;
    if (ctx->n_trecs == ctx->cap_trecs) {
        ctx->cap_trecs = ctx->cap_trecs ? ctx->cap_trecs * 2 : 256;
        ctx->trecs = (bh_time_rec*)realloc(ctx->trecs, sizeof(bh_time_rec) * ctx->cap_trecs);
    }
    bh_time_rec& r = ctx->trecs[ctx->n_trecs++];
    r.kid = kid; r.e0 = take_event(ctx); r.e1 = take_event(ctx);
    (void)hipEventRecord(r.e0, ctx->tstream ? ctx->tstream : ctx->stream);
}
void bh_time_end(bh_ctx* ctx)
{
    if (!ctx->timing_on) return;
    (void)hipEventRecord(ctx->trecs[ctx->n_trecs - 1].e1, ctx->tstream ? ctx->tstream : ctx->stream);
}

static int drain_timing(bh_ctx* ctx)
{
    if (ctx->n_trecs == 0) return BH_OK;
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < BH_MAX_PIPES - 1; p++) if (ctx->aux_stream[p]) BH_HIP(ctx, hipStreamSynchronize(ctx->aux_stream[p]));
    for (int i = 0; i < ctx->n_trecs; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->trecs[i].e0, ctx->trecs[i].e1) == hipSuccess) {
            ctx->t_ms[ctx->trecs[i].kid] += ms; ctx->t_cnt[ctx->trecs[i].kid]++;
        }
        give_event(ctx, ctx->trecs[i].e0); give_event(ctx, ctx->trecs[i].e1);
    }
    ctx->n_trecs = 0;
    return BH_OK;
}

extern "C" int bh_ctx_timing(bh_ctx* ctx, int enable)
{
    if (!ctx) return BH_ERR_INVALID;
    BH_TRY(drain_timing(ctx));
    for (int i = 0; i < BH_K_COUNT; i++) { ctx->t_ms[i] = 0; ctx->t_cnt[i] = 0; }
    ctx->timing_on = enable ? 1 : 0;
    return BH_OK;
}

extern "C" int bh_ctx_timing_read(bh_ctx* ctx, int max_k, double* total_ms, long long* launches, const char** names, int* n_k)
{
    if (!ctx || !total_ms || !launches || !n_k) return BH_ERR_INVALID;
    BH_TRY(drain_timing(ctx));
    int n = BH_K_COUNT < max_k ? BH_K_COUNT : max_k;
    for (int i = 0; i < n; i++) { total_ms[i] = ctx->t_ms[i]; launches[i] = ctx->t_cnt[i]; if (names) names[i] = k_names[i]; }
    *n_k = n;
    return BH_OK;
}
