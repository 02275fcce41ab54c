// Internal declarations shared by the translation units of libboslam_hip.so (gfx950 only).
#ifndef BH_INTERNAL_H
#define BH_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/boslam_hip.h"

#define BH_MAX_LEVELS 16
#define BH_MAX_PIPES  4

struct bh_buf {                      // grow-only device allocation
    void*  p = nullptr;
    size_t cap = 0;
};

struct bh_level_geom {
    int   w, h, pitch;               // pitch: bytes per row, multiple of 64
    int   off;                       // byte offset of the level inside one frame's pyramid (multiple of 64)
    float scale;                     // (float)pow(scaleFactor, l)
    int   nfeat;                     // n_l
    int   tiles_x, tiles_y, tile_base;   // 64x32 tiles
    int   tabx_off, taby_off;        // offsets (in u32) into the resize coefficient table
    int   nsx_in, n_edge, nrb;       // blur: interior / edge 4-px strips per row, 16-row blocks
    int   bblk_base, eblk_base;      // blur: first workgroup of the level in the interior / edge launch
    int   list_off, list_cap;        // corner list of the level inside one frame's list block (entries of 8 B)
    int   moff, mpitch;              // bit-packed mask level: first word inside one frame's block, words per row (pitch / 32 + 1 pad word)
};

struct bh_orb_geom {
    int nlevels, W, H;
    int pyr_stride;                  // bytes per frame
    int ntiles;                      // over all levels
    int fast_thr, edge_thr;
    int tab_len;
    int bblk_total, eblk_total;
    int list_stride;                 // corner-list entries per frame (all levels)
    int mstride;                     // words per frame of the bit-packed mask pyramid
    bh_level_geom lv[BH_MAX_LEVELS];
};

// kernel ids for the optional per-kernel hipEvent timing (bh_ctx_timing / bh_ctx_timing_read)
enum bh_kid { BH_K_GRAY = 0, BH_K_PYR, BH_K_FAST, BH_K_BLUR, BH_K_SELECT, BH_K_BRIEF, BH_K_MATCH, BH_K_BACKPROJ,
              BH_K_POSE_BA, BH_K_BA_LINEARIZE, BH_K_BA_SCHUR, BH_K_BA_SOLVE, BH_K_BA_UPDATE, BH_K_BA_OTHER, BH_K_GATE, BH_K_BOW_SCAN, BH_K_PNP, BH_K_VOC, BH_K_BOW_INV, BH_K_ALIGN, BH_K_COUNT };

struct bh_time_rec { int kid; hipEvent_t e0, e1; };

// k_pyr_mfma's host-built tables for one pyramid level (offsets in u32 words into bh_ctx::rtab): window start per 32-column block,
// tap fragments of the two passes, the mask's run masks per 32-pixel word, the per-tile lists of special mask positions
struct bh_pm_level { int xinfo, xtap, yinfo, ytap, mtab, mwb, fidx, fix, tiles_x, tiles_y; };

struct bh_orb_ws { bh_buf pyr, mpyr, clist, blur, hist, sel_xy, sel_resp, sel_cnt; };   // one chunk pipeline's workspace

struct bh_ctx {
    int         device = 0;
    hipStream_t stream = nullptr;    // stream in use
    hipStream_t own_stream = nullptr;
    hipStream_t aux_stream[BH_MAX_PIPES - 1] = {};   // extra chunk pipelines of bh_orb_detect_describe
    hipStream_t tstream = nullptr;     // stream the per-kernel timing events are recorded on (null: `stream`)
    hipEvent_t  ev_fork = nullptr, ev_join[BH_MAX_PIPES - 1] = {};
    char        err[512] = {0};
    // ---- ORB workspace ----
    bh_orb_geom geom = {};
    bh_orb_params geom_params = {};
    int         geom_valid = 0;
    int         ws_frames = 0;       // frames the pyramid workspace holds
    bh_orb_ws   ows[BH_MAX_PIPES];
    bh_buf      tab, status;
    bh_buf      rtab;                // matrix-core resize tables (k_pyr_mfma)
    bh_pm_level pm[BH_MAX_LEVELS] = {};
    int         pm_ok = 0;           // every level's windows fit the kernel's tile (else k_pyr_down_bits)
    int         gl_sr = 0;           // k_gray_l1: level-1 rows per strip for the current geometry (0: level 0 and level 1 as two launches)
    int         pp_sr[BH_MAX_LEVELS] = {};      // k_pyr_pair: rows of level l + 1 per strip when levels l, l + 1 are made by one launch (0: not as a pair)
    int         last_pipe = 0;
    bh_buf      st_img, st_mask, st_depth, st_kps, st_desc, st_nkp;     // host-mode staging
    bh_buf      st_a, st_b, st_c, st_d, st_e, st_f, st_g, st_h;         // generic staging (match / BA)
    bh_buf      pnp_p0;                                                  // four-point PnP problems: the P3P pose per problem (k_pnp_p3p -> k_pnp_ransac)
    bh_buf      tl_w, tl_in;                                              // bh_track_local_map: gathered train side / see vectors, host-mode staging
    bh_buf      tr_w, tr_in;                                              // bh_track_refkf: intermediates (raw matches, gathered BA inputs, chi2), host-mode inputs
    void*       pin = nullptr;                                            // pinned host staging (grow-only): small problems go up in one copy and come back in one
    size_t      pin_cap = 0;
    int         last_B = 0;
    int         last_has_mask = 0, last_mask_bits = 0;
    // ---- per-kernel timing ----
    int         timing_on = 0;
    bh_time_rec* trecs = nullptr;
    int         n_trecs = 0, cap_trecs = 0;
    hipEvent_t* epool = nullptr;
    int         n_epool = 0, cap_epool = 0;
    double      t_ms[BH_K_COUNT] = {0};
    long long   t_cnt[BH_K_COUNT] = {0};
};

// Bracket one kernel launch with events when timing is on.
void bh_time_begin(bh_ctx* ctx, int kid);
void bh_time_end(bh_ctx* ctx);
#define BH_LAUNCH(ctx, kid, kernel, grid, block, st, ...)                                          \
    do {                                                                                           \
        bh_time_begin((ctx), (kid));                                                               \
        hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                               \
        bh_time_end(ctx);                                                                          \
    } while (0)

int  bh_set_err(bh_ctx* ctx, int code, const char* fmt, ...);
int  bh_reserve(bh_ctx* ctx, bh_buf* b, size_t bytes);
int  bh_pin_reserve(bh_ctx* ctx, size_t bytes);     // ctx->pin holds at least `bytes` (synchronises the stream before it grows)
int  bh_status_ready(bh_ctx* ctx);      // allocates + zeroes the sticky device status word on first use

#define BH_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return bh_set_err((ctx), BH_ERR_HIP, "%s failed: %s (%s:%d)", #call,                   \
                              hipGetErrorString(e_), __FILE__, __LINE__);                          \
    } while (0)

#define BH_TRY(expr)                                                                               \
    do { int rc_ = (expr); if (rc_ != BH_OK) return rc_; } while (0)

// device status word bits (bh_ctx::status[0])
#define BH_ST_KP_OVERFLOW    1u   // more keypoints than max_kp
#define BH_ST_CAND_OVERFLOW  2u   // more FAST candidates above the retainBest threshold than the LDS list holds

#endif
