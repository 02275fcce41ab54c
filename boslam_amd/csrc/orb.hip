// orb.hip — batched ORB detect + describe for gfx950 (MI355X).
//
// Replaces cv2.ORB_create(**orb_params).detectAndCompute(frame, mask) — reference camera.py:49-62,141 — for a batch of
// frames resident in HBM.  Stages (one kernel each, all frames and all pyramid levels per launch where the data
// dependence allows):
//   k_gray_mask   BGR -> gray (level 0) and depth -> level-0 mask BITS (mask bytes when the caller passes them)   streaming, 4 px / lane
//   k_pyr_down_bits / k_pyr_down   level l from level l-1 (bit-exact fixed-point bilinear), image and mask   (nlevels-1 launches)
//   k_fast        FAST-9/16 pretest + ring score + 3x3 NMS + mask/border filter on 128x32 LDS tiles -> per-level corner lists and
//                 score histograms; the 7x7 sigma-2 Gaussian of the same tile as two int8 MFMA products (BRIEF's input)
//   k_select      one workgroup per (level, frame): retainBest(2 n_l) threshold from the histogram, Harris response of the listed
//                 corners, retainBest(n_l) by rank counting in LDS -> per-level staging slots in canonical (y, x) order
//   k_describe    64 keypoints per workgroup: intensity-centroid sums (wave per keypoint), angle / sin / cos (lane per keypoint),
//                 256 rotated pixel-pair tests from an LDS-staged window, 4 ballots -> 32 B
// Every integer stage is exact; the float formulas (Harris response, fastAtan2, cos/sin of the angle) are written
// one IEEE operation at a time and the file is compiled with -ffp-contract=off.
#include "bh_internal.h"
#include <type_traits>
#include <math.h>
#include <float.h>
#include <vector>
#include <algorithm>

#ifndef TILE_W
#define TILE_W 128
#endif
#define TILE_H 32

// ------------------------------------------------------------------------------------------------ geometry (host)
static int round_even_f(float v) { return (int)lrintf(v); }
static int round_even_d(double v) { return (int)lrint(v); }
static int align_up(int v, int a) { return (v + a - 1) / a * a; }

static int build_geom(const bh_orb_params* p, int H, int W, bh_orb_geom* g)
{
    if (!p || p->nlevels < 1 || p->nlevels > BH_MAX_LEVELS || p->firstLevel != 0 || p->patchSize != 31 ||
        p->nfeatures < 0 || p->scaleFactor <= 1.0f || p->scaleFactor > 1.3f ||   // k_pyr_down: 4 outputs span <= 3 source words
        H < 1 || W < 1 || W > 16383 || H > 16383 ||
        p->edgeThreshold < 19)   // Harris(4) / IC(15) / BRIEF(<=19) windows must stay inside the level image
        return BH_ERR_INVALID;
    memset(g, 0, sizeof(*g));
    g->nlevels = p->nlevels; g->W = W; g->H = H;
    g->fast_thr = p->fastThreshold < 0 ? 0 : (p->fastThreshold > 255 ? 255 : p->fastThreshold);
    g->edge_thr = p->edgeThreshold;
    // features per level
    float factor = (float)(1.0 / p->scaleFactor);
    float nd = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p->nlevels));
    int sum = 0;
    int off = 0, tiles = 0, tab = 0, bblk = 0, eblk = 0, lst = 0, moff = 0;
    for (int l = 0; l < p->nlevels; l++) {
        bh_level_geom& L = g->lv[l];
        L.scale = (float)pow((double)p->scaleFactor, (double)l);
        L.w = round_even_f((float)W / L.scale);
        L.h = round_even_f((float)H / L.scale);
        if (L.w < 1 || L.h < 1) return BH_ERR_INVALID;
        L.pitch = align_up(L.w, 64);
        L.off = off; off += L.pitch * L.h;
        L.mpitch = (L.pitch >> 5) + 1; L.moff = moff; moff += L.mpitch * L.h;
        L.tiles_x = (L.w + TILE_W - 1) / TILE_W;
        L.tiles_y = (L.h + TILE_H - 1) / TILE_H;
        L.tile_base = tiles; tiles += L.tiles_x * L.tiles_y;
        // blur strips: interior strips sx = 1 .. nsx_in (x0-4 >= 0 and x0+8 <= w), the rest are edge strips
        L.nsx_in = (L.w - 8) / 4 > 0 ? (L.w - 8) / 4 : 0;
        L.n_edge = (L.w + 3) / 4 - L.nsx_in;
        L.nrb = (L.h + 15) / 16;
        L.bblk_base = bblk; bblk += (L.nsx_in * L.nrb + 255) / 256;
        L.eblk_base = eblk; eblk += (L.n_edge * L.nrb + 255) / 256;
        // strict 3x3 maxima: at most one per 2x2 block in theory; w*h/8 entries leave a wide margin over any real image (overflow is flagged)
        L.list_off = lst; L.list_cap = align_up((L.w * L.h + 7) / 8, 8); lst += L.list_cap;
        L.tabx_off = tab; tab += align_up(L.w, 4);
        L.taby_off = tab; tab += align_up(L.h, 4);
        if (l < p->nlevels - 1) { L.nfeat = round_even_f(nd); sum += L.nfeat; nd *= factor; }
        else L.nfeat = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    }
    g->list_stride = lst; g->mstride = align_up(moff + 4, 16);
    g->pyr_stride = off; g->ntiles = tiles; g->tab_len = tab; g->bblk_total = bblk; g->eblk_total = eblk;
    return BH_OK;
}

// Fixed-point bilinear coefficients per axis: entry = (source offset << 16) | c1, c0 = 256 - c1.
static void lin_coeffs(int src, int dst, uint32_t* out)
{
    double inv_scale = (double)dst / (double)src;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; d++) {
        double fval = scale * ((double)d + 0.5) - 0.5;
        int ival = (int)floor(fval);
        uint32_t ofs, c1;
        if (ival >= 0 && src > 1) {
            if (ival < src - 1) { ofs = (uint32_t)ival; c1 = (uint32_t)round_even_d((fval - (double)ival) * 256.0); }
            else { ofs = (uint32_t)(src - 1); c1 = 0; }
        } else { ofs = 0; c1 = 0; }
        out[d] = (ofs << 16) | c1;
    }
}

// One u32 per 8 lanes from 4 bits each: lane 8k+7 returns the nibbles of lanes 8k .. 8k+7 (lane 8k in bits 0-3).  DPP row shifts,
// zero fill at the row start; the 16 lanes of a DPP row must all be active.
template <int CTRL> __device__ __forceinline__ uint32_t u_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t nib_gather8(uint32_t nib, int lane)
{
    uint32_t v = nib << (4 * (lane & 7));
    v |= u_dpp<0x111>(v); v |= u_dpp<0x112>(v); v |= u_dpp<0x114>(v);
    return v;
}

// ------------------------------------------------------------------------------------------------ gray + mask, level 0
__global__ __launch_bounds__(256) void k_gray_mask(const uint8_t* __restrict__ img, int channels,
                                                   const uint8_t* __restrict__ mask,
                                                   const uint16_t* __restrict__ depth,
                                                   uint8_t* __restrict__ pyr, uint8_t* __restrict__ mpyr,
                                                   int W, int H, int pitch, int pyr_stride, uint32_t* __restrict__ mbits, int mpitch, int mstride)
{
    const int groups = pitch >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int y = idx / groups, x = (idx - y * groups) << 2;
    if (y >= H) return;
    const size_t f = blockIdx.y;
    const size_t px = (size_t)f * H * W + (size_t)y * W + x;
    uint32_t g4 = 0, m4 = 0;
    if (x + 3 < W && (W & 3) == 0) {
        if (channels == 3) {
            const uint32_t* s = reinterpret_cast<const uint32_t*>(img + px * 3);
            uint32_t a = s[0], b = s[1], c = s[2];          // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
            uint32_t bb[4] = { a & 255u, (a >> 24) & 255u, (b >> 16) & 255u, (c >> 8) & 255u };
            uint32_t gg[4] = { (a >> 8) & 255u, b & 255u, (b >> 24) & 255u, (c >> 16) & 255u };
            uint32_t rr[4] = { (a >> 16) & 255u, (b >> 8) & 255u, c & 255u, (c >> 24) & 255u };
#pragma unroll
            for (int i = 0; i < 4; i++)
                g4 |= ((bb[i] * 1868u + gg[i] * 9617u + rr[i] * 4899u + 8192u) >> 14) << (8 * i);
        } else {
            g4 = *reinterpret_cast<const uint32_t*>(img + px);
        }
        if (mask) m4 = *reinterpret_cast<const uint32_t*>(mask + px);
        else if (depth) {
            uint2 d = *reinterpret_cast<const uint2*>(depth + px);
            m4 = ((d.x & 0xffffu) ? 0xffu : 0u) | ((d.x >> 16) ? 0xff00u : 0u) |
                 ((d.y & 0xffffu) ? 0xff0000u : 0u) | ((d.y >> 16) ? 0xff000000u : 0u);
        }
    } else {
        for (int i = 0; i < 4; i++) {
            if (x + i >= W) break;
            uint32_t gv;
            if (channels == 3) {
                const uint8_t* s = img + (px + i) * 3;
                gv = ((uint32_t)s[0] * 1868u + (uint32_t)s[1] * 9617u + (uint32_t)s[2] * 4899u + 8192u) >> 14;
            } else gv = img[px + i];
            g4 |= gv << (8 * i);
            uint32_t mv = mask ? mask[px + i] : (depth ? (depth[px + i] ? 255u : 0u) : 0u);
            m4 |= mv << (8 * i);
        }
    }
    const size_t o = f * pyr_stride + (size_t)y * pitch + x;
    *reinterpret_cast<uint32_t*>(pyr + o) = g4;
    if (mpyr) *reinterpret_cast<uint32_t*>(mpyr + o) = m4;
    if (mbits) {                                               // depth-derived masks are binary: one bit per pixel, 8 lanes make a word
        const uint32_t nib = ((m4 & 0xffu) ? 1u : 0u) | ((m4 & 0xff00u) ? 2u : 0u) | ((m4 & 0xff0000u) ? 4u : 0u) | ((m4 >> 24) ? 8u : 0u);
        const uint32_t wd = nib_gather8(nib, threadIdx.x);
        if ((threadIdx.x & 7) == 7) mbits[f * mstride + (size_t)y * mpitch + (x >> 5)] = wd;
    }
}

// The reference call's case (camera.py:141: BGR frame, mask = (depth > 0) * 255) with a width that is a multiple of 4, specialised:
// the luma (B 1868 + G 9617 + R 4899 + 8192) >> 14 straight from the three packed BGR words by v_dot4_u32_u8 (weights split into
// high and low bytes: 1868 = 7 x 256 + 76, 9617 = 37 x 256 + 145, 4899 = 19 x 256 + 35; a pixel that straddles two words chains two
// dot products), the four depth values to a nibble by one packed min per register, frame-uniform 64-bit bases with 32-bit lane
// offsets.  50 vector instructions per lane where the general kernel above takes 118; both are at the HBM roof on their own, but the
// front end as a whole is bound by vector-instruction issue, and this kernel runs beside the others.
// four pixels (x .. x + 3 of row y, x < W) of the reference call's level 0: gray bytes g4 and the mask nibble (depth != 0)
__device__ __forceinline__ void gray_item4(const uint8_t* __restrict__ fimg, const uint16_t* __restrict__ fd, int W, int y, int x, uint32_t& g4, uint32_t& nib)
{
    const uint32_t px = __umul24((uint32_t)y, (uint32_t)W) + (uint32_t)x;
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(fimg + __umul24(px, 3u));
    const uint32_t a = sp[0], b = sp[1], c = sp[2];       // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
    const uint32_t h0 = __builtin_amdgcn_udot4(a, 7u | 37u << 8 | 19u << 16, 0u, false), l0 = __builtin_amdgcn_udot4(a, 76u | 145u << 8 | 35u << 16, 8192u, false);
    const uint32_t h1 = __builtin_amdgcn_udot4(a, 7u << 24, __builtin_amdgcn_udot4(b, 37u | 19u << 8, 0u, false), false);
    const uint32_t l1 = __builtin_amdgcn_udot4(a, 76u << 24, __builtin_amdgcn_udot4(b, 145u | 35u << 8, 8192u, false), false);
    const uint32_t h2 = __builtin_amdgcn_udot4(b, 7u << 16 | 37u << 24, __builtin_amdgcn_udot4(c, 19u, 0u, false), false);
    const uint32_t l2 = __builtin_amdgcn_udot4(b, 76u << 16 | 145u << 24, __builtin_amdgcn_udot4(c, 35u, 8192u, false), false);
    const uint32_t h3 = __builtin_amdgcn_udot4(c, 7u << 8 | 37u << 16 | 19u << 24, 0u, false), l3 = __builtin_amdgcn_udot4(c, 76u << 8 | 145u << 16 | 35u << 24, 8192u, false);
    g4 = __builtin_amdgcn_ubfe((h0 << 8) + l0, 14u, 8u) | __builtin_amdgcn_ubfe((h1 << 8) + l1, 14u, 8u) << 8 |
         __builtin_amdgcn_ubfe((h2 << 8) + l2, 14u, 8u) << 16 | __builtin_amdgcn_ubfe((h3 << 8) + l3, 14u, 8u) << 24;
    typedef unsigned short gm_us2 __attribute__((ext_vector_type(2)));
    const uint2 d = *reinterpret_cast<const uint2*>(fd + px);
    const gm_us2 one = gm_us2{ 1, 1 };
    const uint32_t nx = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(gm_us2, d.x), one));      // 1 / 0 in each half
    const uint32_t ny = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(gm_us2, d.y), one));
    nib = ((nx | nx >> 15) & 3u) | ((ny | ny >> 15) & 3u) << 2;
}

__global__ __launch_bounds__(256) void k_gray_bits(const uint8_t* __restrict__ img, const uint16_t* __restrict__ depth, uint8_t* __restrict__ pyr,
                                                   int W, int H, int pitch, int pyr_stride, uint32_t* __restrict__ mbits, int mpitch, int mstride)
{
    const int groups = pitch >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    int y;
    {   // idx / groups: both multiples of 16 apart from idx's low bits, the divisor at most 4096 / 4; a float quotient off by one is put right
        const int q = idx >> 4, d = groups >> 4;
        y = (int)((float)q * __builtin_amdgcn_rcpf((float)d));
        const int r = q - __mul24(y, d);
        y += r >= d ? 1 : (r < 0 ? -1 : 0);
    }
    const int x = (idx - __mul24(y, groups)) << 2;      // (24-bit multiplies throughout: a full 32-bit one runs at a quarter of the rate)
    if (y >= H) return;                                        // whole rows leave together (groups is a multiple of 16: DPP rows stay whole)
    const uint8_t* __restrict__ fimg = img + (size_t)blockIdx.y * H * W * 3;
    const uint16_t* __restrict__ fd = depth + (size_t)blockIdx.y * H * W;
    uint8_t* __restrict__ fp = pyr + (size_t)blockIdx.y * pyr_stride;
    uint32_t* __restrict__ fm = mbits + (size_t)blockIdx.y * mstride;
    uint32_t g4 = 0, nib = 0;
    if (x < W) gray_item4(fimg, fd, W, y, x, g4, nib);
    *reinterpret_cast<uint32_t*>(fp + (__umul24((uint32_t)y, (uint32_t)pitch) + (uint32_t)x)) = g4;
    const uint32_t wd = nib_gather8(nib, threadIdx.x);
    if ((threadIdx.x & 7) == 7) fm[__umul24((uint32_t)y, (uint32_t)mpitch) + (uint32_t)(x >> 5)] = wd;
}
// ------------------------------------------------------------------------------------------------ pyramid level l from l-1
// INTER_LINEAR_EXACT arithmetic: horizontal 8.8 fixed point, vertical 16.16, round to nearest.
// One lane produces 4 output pixels.  Its 4 coefficient entries come in one 16-B load; the <= 6 source columns they
// reference fit one 8-byte window starting at the first column, so each of the two source rows is ONE (unaligned) 8-byte
// load per plane (image, mask) — 6 loads per lane, all issued before the first use — and the byte pairs are picked with a
// 64-bit shift instead of 16 single-byte loads per plane.
__device__ __forceinline__ unsigned long long pyr_load8(const uint8_t* __restrict__ base, size_t off)
{   // 8 bytes from byte offset `off` of a 4-byte-aligned buffer as ONE aligned 12-byte load (global_load_dwordx3) + a funnel shift:
    // an unaligned 8-byte request is split by the memory pipeline, which is what bounded this kernel (measured)
    const uint32_t* q = reinterpret_cast<const uint32_t*>(base + (off & ~(size_t)3));
    const uint32_t sh = ((uint32_t)off & 3u) << 3;
    const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return ((unsigned long long)hi << 32) | lo;
}

// The depth-mask form (bit-packed mask pyramid).  With 0 / 255 inputs the reference arithmetic on the mask
//   v = ((256-yc1) h0 + yc1 h1 + 32768) >> 16,  h = 255 ((256-c1) b0 + c1 b1),  keep v == 255
// is  T = (256-yc1) H0 + yc1 H1 >= 65408,  H = (256-c1) b0 + c1 b1  (checked exhaustively over c1, yc1, the 16 bit patterns); a lane
// whose six source columns x two rows are all set / all clear needs none of it.  A lane produces 4 pixels of PD_R consecutive rows
// (one row per lane: waves of 0.7 us, 7.5 resident per CU; measured 1.23 / 1.11 / 1.13 / 1.23 ms per 1000 frames at PD_R = 1 / 2 / 4 / 8);
// all loads of the PD_R rows are issued before the first store (loads and stores go through the same pointer, which
// would otherwise serialise the rows).  Lanes in the padding columns compute on column 0 and are masked at the end: they must
// reach the DPP gather anyway.
#ifndef PD_R
#define PD_R 2
#endif
typedef unsigned short pd_us2 __attribute__((ext_vector_type(2)));

// ---- level l from level l-1, depth-mask form, round 3.  One lane = 4 output pixels x PD_R rows, lanes flattened over (row block,
// column group) so that every lane of every wave holds real work.  The kernel is bound by its vector instructions (measured in three
// forms this round: fewer vector-memory instructions through an LDS copy of the source rows, fewer dependent loads through a row loop
// per lane, and fused small levels all ran at the same ~36 lane-instructions per output pixel), so what is spent per pixel is what counts:
//   * everything that depends on x only - tap offsets, packed coefficients, v_perm selectors - is set up once per lane (pyr_xsetup);
//   * horizontal pass: per pixel ONE v_perm per tap builds (row a, row b) as two 16-bit halves from the rows' 4-byte windows, then
//     H = (256 - c1) P0 + c1 P1 is a packed multiply + multiply-add (<= 65 280: exact in 16 bits) for both rows at once;
//   * vertical pass: v = ((256 - yc) Ha + yc Hb + 32768) >> 16 is one v_dot2_u32_u16 of (Ha, Hb) with (256 - yc, yc);
//   * mask: with 0 / 255 inputs the reference's resize + threshold(254) is "the taps that are clear weigh <= 128 of 65 536"
//     (T >= 65 408, verified exhaustively in round 2).  When the smallest weight product min(c1, 256 - c1) min(yc, 256 - yc) exceeds 128
//     that is the AND of the four taps (one shift-and chain per lane).  A wave in which some lane has a possibly ignorable tap AND a
//     window with some but not all bits set evaluates T for its mixed lanes instead (bit-field extracts feeding 24-bit multiply-adds).
struct pyr_xs {
    uint32_t selP0[4];          // v_perm selector of tap 0 of pixel i: (row a byte s, 0, row b byte s, 0); tap 1 = + 0x00010001
    pd_us2 C0[4], C1[4];        // (256 - c1, 256 - c1), (c1, c1)
    uint32_t sv[4];             // tap-0 column of pixel i relative to the window's first column (0 .. 4)
    uint32_t mmin;              // min over the pixels of min(c1, 256 - c1)
    uint32_t boff, shb, shm;    // byte offset of the aligned 12-byte window inside a source row; funnel shifts (bits) to the window / its second word
    uint32_t mcol, mbit;        // mask: first word of the row, bit inside it
    int over, x;
};
struct pyr_ld { uint32_t a0, a1, a2, b0, b1, b2, ma0, ma1, mb0, mb1; };

__device__ __forceinline__ void pyr_xsetup(pyr_xs& X, const bh_level_geom& D, const uint32_t* __restrict__ tab, int x)
{
    X.x = x;
    X.over = min(max(x + 4 - D.w, 0), 4);                      // outputs of this lane beyond the row
    const __amdgpu_buffer_rsrc_t rs_tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tab), 0, 0x7fffffff, 0x00020000);
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v t4 = __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rs_tab, (X.over == 4 ? 0 : x) << 2, D.tabx_off << 2, 0));
    const uint4 tx4 = make_uint4(t4.x, t4.y, t4.z, t4.w);
    const uint32_t txs[4] = { tx4.x, tx4.y, tx4.z, tx4.w };
    const uint32_t base = tx4.x >> 16;
    X.mmin = 256u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t c1 = txs[i] & 0xffffu;
        X.sv[i] = (txs[i] >> 16) - base;
        X.C1[i] = pd_us2{ (unsigned short)c1, (unsigned short)c1 };
        X.C0[i] = pd_us2{ (unsigned short)(256u - c1), (unsigned short)(256u - c1) };
        X.mmin = min(X.mmin, min(c1, 256u - c1));
    }
    // pixels 0, 1 read the window word at byte 0 (their tap 0 is at byte <= 2), pixels 2, 3 the word at byte m = min(s_2, 2)
    // (s - m <= 2), so a tap pair (s, s + 1) never leaves its word
    const uint32_t m2 = min(X.sv[2], 2u);
#pragma unroll
    for (int i = 0; i < 4; i++) X.selP0[i] = 0x0c040c00u + (i < 2 ? X.sv[i] : X.sv[i] - m2) * 0x00010001u;
    X.boff = base & ~3u; X.shb = (base & 3u) << 3; X.shm = m2 << 3;
    X.mcol = base >> 5; X.mbit = base & 31u;
}

// Buffer loads (round 5): a row's offset is ONE 24-bit multiply-add per load (rows and pitches are far below 2^24) - as pointer arithmetic every
// load cost a full 32-bit multiply (quarter rate: ten of them were a sixth of the kernel's vector cycles) and a 64-bit add.
typedef uint32_t pyr_u3 __attribute__((ext_vector_type(3)));
typedef uint32_t pyr_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pyr_load(pyr_ld& L, __amdgpu_buffer_rsrc_t rs_src, __amdgpu_buffer_rsrc_t rs_ms, const bh_level_geom& S,
                                         const pyr_xs& X, int r0, int r1)
{
    const pyr_u3 a = __builtin_bit_cast(pyr_u3, __builtin_amdgcn_raw_buffer_load_b96(rs_src, (int)(__umul24((uint32_t)r0, (uint32_t)S.pitch) + X.boff), 0, 0));
    const pyr_u3 b = __builtin_bit_cast(pyr_u3, __builtin_amdgcn_raw_buffer_load_b96(rs_src, (int)(__umul24((uint32_t)r1, (uint32_t)S.pitch) + X.boff), 0, 0));
    L.a0 = a.x; L.a1 = a.y; L.a2 = a.z; L.b0 = b.x; L.b1 = b.y; L.b2 = b.z;
    const pyr_u2 ma = __builtin_bit_cast(pyr_u2, __builtin_amdgcn_raw_buffer_load_b64(rs_ms, (int)((__umul24((uint32_t)r0, (uint32_t)S.mpitch) + X.mcol) << 2), 0, 0));
    const pyr_u2 mb = __builtin_bit_cast(pyr_u2, __builtin_amdgcn_raw_buffer_load_b64(rs_ms, (int)((__umul24((uint32_t)r1, (uint32_t)S.mpitch) + X.mcol) << 2), 0, 0));
    L.ma0 = ma.x; L.ma1 = ma.y; L.mb0 = mb.x; L.mb1 = mb.y;
}

__device__ __forceinline__ void pyr_row(const pyr_ld& L, const pyr_xs& X, uint32_t yc, int lane, uint32_t* o4, uint32_t* nbw)
{
    const pd_us2 Wy = pd_us2{ (unsigned short)(256u - yc), (unsigned short)yc };
    const uint32_t alo = __builtin_amdgcn_alignbit(L.a1, L.a0, X.shb), ahi = __builtin_amdgcn_alignbit(L.a2, L.a1, X.shb);
    const uint32_t blo = __builtin_amdgcn_alignbit(L.b1, L.b0, X.shb), bhi = __builtin_amdgcn_alignbit(L.b2, L.b1, X.shb);
    const uint32_t wa[2] = { alo, __builtin_amdgcn_alignbit(ahi, alo, X.shm) }, wb[2] = { blo, __builtin_amdgcn_alignbit(bhi, blo, X.shm) };
    uint32_t r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const pd_us2 P0 = __builtin_bit_cast(pd_us2, __builtin_amdgcn_perm(wb[i >> 1], wa[i >> 1], X.selP0[i]));
        const pd_us2 P1 = __builtin_bit_cast(pd_us2, __builtin_amdgcn_perm(wb[i >> 1], wa[i >> 1], X.selP0[i] + 0x00010001u));
        const pd_us2 Hh = (pd_us2)(X.C1[i] * P1 + (pd_us2)(X.C0[i] * P0));          // (Ha, Hb)
        r[i] = __builtin_amdgcn_udot2(Hh, Wy, 32768u, false);                       // result byte = bits 16 .. 23
    }
    const uint32_t v4 = __builtin_amdgcn_perm(__builtin_amdgcn_perm(r[3], r[2], 0x0c0c0602u), __builtin_amdgcn_perm(r[1], r[0], 0x0c0c0602u), 0x05040100u);
    const uint32_t A8 = __builtin_amdgcn_alignbit(L.ma1, L.ma0, X.mbit), B8 = __builtin_amdgcn_alignbit(L.mb1, L.mb0, X.mbit);
    const uint32_t both = A8 & B8 & 0x3fu, any = (A8 | B8) & 0x3fu;
    const bool mixed = both != 0x3fu && any != 0u;
    uint32_t nib = both == 0x3fu ? 0xfu : 0u;
    if (__builtin_amdgcn_ballot_w64(mixed && X.mmin * min(yc, 256u - yc) <= 128u) == 0ull) {      // wave-uniform: no tap of any mixed lane can be ignored
        if (mixed) {                                             // every tap weighs more than 128: set <=> all four taps set
            const uint32_t a4 = A8 & (A8 >> 1) & B8 & (B8 >> 1);
#pragma unroll
            for (int i = 0; i < 4; i++) nib |= __builtin_amdgcn_ubfe(a4, X.sv[i], 1u) << i;
        }
    } else if (mixed) {
        const uint32_t wy0 = 256u - yc;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t w1 = X.C1[i].x, w0 = X.C0[i].x;
            const uint32_t H0 = __builtin_amdgcn_ubfe(A8, X.sv[i], 1u) * w0 + __builtin_amdgcn_ubfe(A8, X.sv[i] + 1u, 1u) * w1;
            const uint32_t H1 = __builtin_amdgcn_ubfe(B8, X.sv[i], 1u) * w0 + __builtin_amdgcn_ubfe(B8, X.sv[i] + 1u, 1u) * w1;
            nib |= (wy0 * H0 + yc * H1 >= 65408u ? 1u : 0u) << i;
        }
    }
    *o4 = X.over == 4 ? 0u : (v4 & (0xffffffffu >> (8 * X.over)));      // padding columns stay zero
    *nbw = nib_gather8(nib & (0xfu >> X.over), lane);
}

// item index -> (row block, column group) without the generic 32-bit division: both are multiples of 16 apart from idx's low bits, the
// divisor is at most 256, and a float quotient off by one is put right by the remainder
__device__ __forceinline__ int pyr_div_groups(int idx, int groups)
{
    const int q = idx >> 4, d = groups >> 4;
    int yb = (int)((float)q * __builtin_amdgcn_rcpf((float)d));
    const int r = q - __mul24(yb, d);
    yb += r >= d ? 1 : (r < 0 ? -1 : 0);
    return yb;
}

#ifndef PD_R
#define PD_R 2
#endif
__global__ __launch_bounds__(256) void k_pyr_down_bits(uint8_t* __restrict__ pyr, bh_orb_geom g, int level, const uint32_t* __restrict__ tab,
                                                       uint32_t* __restrict__ mbits)
{
    const bh_level_geom S = g.lv[level - 1], D = g.lv[level];
    const int groups = D.pitch >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int yb = pyr_div_groups(idx, groups);
    const int x = (idx - __mul24(yb, groups)) << 2, y0 = yb * PD_R;
    if (y0 >= D.h) return;                                     // whole rows (pitch / 4 is a multiple of 16 lanes: DPP rows leave together)
    // wave-uniform 64-bit bases, 32-bit lane offsets: the loads and stores take the scalar-base + vector-offset form
    const uint8_t* __restrict__ src = pyr + (size_t)blockIdx.y * g.pyr_stride + S.off;
    uint8_t* __restrict__ dst = pyr + (size_t)blockIdx.y * g.pyr_stride + D.off;
    const uint32_t* __restrict__ ms = mbits + (size_t)blockIdx.y * g.mstride + S.moff;
    uint32_t* __restrict__ md = mbits + (size_t)blockIdx.y * g.mstride + D.moff;
    pyr_xs X;
    pyr_xsetup(X, D, tab, x);
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ms = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ms), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_md = __builtin_amdgcn_make_buffer_rsrc(md, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tab), 0, 0x7fffffff, 0x00020000);
    pyr_ld L[PD_R]; uint32_t yc[PD_R];
#pragma unroll
    for (int k = 0; k < PD_R; k++) {                            // all loads of the PD_R rows before the first store (same pointer: they would serialise)
        const uint32_t ty = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_tab, min(y0 + k, D.h - 1) << 2, D.taby_off << 2, 0);      // rows beyond the level repeat the last one and are not stored
        const int r0 = (int)(ty >> 16), r1 = min(r0 + 1, S.h - 1);
        yc[k] = ty & 0xffffu;
        pyr_load(L[k], rs_src, rs_ms, S, X, r0, r1);
    }
    uint32_t o4[PD_R], nbw[PD_R];
#pragma unroll
    for (int k = 0; k < PD_R; k++) pyr_row(L[k], X, yc[k], threadIdx.x, &o4[k], &nbw[k]);
    const int od = (int)(__umul24((uint32_t)y0, (uint32_t)D.pitch) + (uint32_t)x), om = (int)((__umul24((uint32_t)y0, (uint32_t)D.mpitch) + (uint32_t)(x >> 5)) << 2);
#pragma unroll
    for (int k = 0; k < PD_R; k++) {
        const int y = y0 + k;
        if (y < D.h) {
            __builtin_amdgcn_raw_buffer_store_b32(o4[k], rs_dst, od, k * D.pitch, 0);
            if ((threadIdx.x & 7) == 7) __builtin_amdgcn_raw_buffer_store_b32(nbw[k], rs_md, om, k * D.mpitch * 4, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ level 0 and level 1 in one launch
// k_gray_l1 (round 5): the reference call's gray + mask-bit level 0 AND level 1 of the pyramid by one workgroup per strip of GL_SR level-1
// rows.  k_gray_bits streams 1.9 MB per frame at the HBM roof with 60 % of its vector-issue slots idle, and level 1 is a third of the
// pyramid's pixels at ~30 instructions each with little memory traffic: here the workgroups of a CU sit in different phases - some
// waiting for their BGR / depth words, some resizing - so the resize rides in the issue slots the stream leaves free.  A workgroup
//   1. computes the level-0 rows its level-1 rows read (src(y1a) .. src(y1b - 1) + 1: at most GL_R0) with gray_item4, writes them to
//      HBM (the rows two strips share are written twice, identically) and keeps them, with their mask words, in LDS;
//   2. resizes from LDS with the arithmetic of k_pyr_down_bits (pyr_xsetup / pyr_row: the same statements, so the same bytes) and
//      stores level 1.  Level 0 is not read back from HBM, and one launch per chunk is gone.
#ifndef GL_PR
#define GL_PR 4                // level-1 rows per lane item in phase 2 (the column set-up is paid once per item)
#endif
#ifndef GL_R0
#define GL_R0 26               // level-0 rows a workgroup stages (18.8 KB of LDS at 640 columns: eight workgroups per CU)
#endif
__device__ __forceinline__ void pyr_load_lds(pyr_ld& L, const uint8_t* __restrict__ sg, const uint32_t* __restrict__ sm, int pitch, int mpitch,
                                             const pyr_xs& X, int r0, int r1)
{
    const uint32_t* qa = reinterpret_cast<const uint32_t*>(sg + __umul24((uint32_t)r0, (uint32_t)pitch) + X.boff);
    const uint32_t* qb = reinterpret_cast<const uint32_t*>(sg + __umul24((uint32_t)r1, (uint32_t)pitch) + X.boff);
    L.a0 = qa[0]; L.a1 = qa[1]; L.a2 = qa[2]; L.b0 = qb[0]; L.b1 = qb[1]; L.b2 = qb[2];
    const uint32_t* pa = sm + __umul24((uint32_t)r0, (uint32_t)mpitch) + X.mcol; const uint32_t* pb = sm + __umul24((uint32_t)r1, (uint32_t)mpitch) + X.mcol;
    L.ma0 = pa[0]; L.ma1 = pa[1]; L.mb0 = pb[0]; L.mb1 = pb[1];
}
__global__ __launch_bounds__(256) void k_gray_l1(const uint8_t* __restrict__ img, const uint16_t* __restrict__ depth, uint8_t* __restrict__ pyr,
                                                 bh_orb_geom g, int sr, const uint32_t* __restrict__ tab, uint32_t* __restrict__ mbits, uint32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t gl_sm[];
    const bh_level_geom S = g.lv[0], D = g.lv[1];
    uint8_t* const sg = gl_sm;                                             // [GL_R0][S.pitch + 16] gray rows (+16: a 12-byte window may start at the last column group)
    const int gp = S.pitch + 16;
    uint32_t* const sm = reinterpret_cast<uint32_t*>(gl_sm + (size_t)GL_R0 * gp);      // [GL_R0][S.mpitch + 1] mask words
    const int mp = S.mpitch + 1;
    const int tid = threadIdx.x, W = g.W, H = g.H;
    const int y1a = blockIdx.x * sr, y1b = min(D.h, y1a + sr);
    const uint8_t* __restrict__ fimg = img + (size_t)blockIdx.y * H * W * 3;
    const uint16_t* __restrict__ fd = depth + (size_t)blockIdx.y * H * W;
    uint8_t* __restrict__ fp = pyr + (size_t)blockIdx.y * g.pyr_stride;
    uint32_t* __restrict__ fm = mbits + (size_t)blockIdx.y * g.mstride;
    const int ra = (int)(tab[D.taby_off + y1a] >> 16);
    int rb = min((int)(tab[D.taby_off + y1b - 1] >> 16) + 1, S.h - 1);
    if (blockIdx.x == gridDim.x - 1) rb = S.h - 1;                          // (the last strip finishes level 0 whatever the last source row is)
    const int nr = rb - ra + 1;
    if (nr > GL_R0) { if (tid == 0) atomicOr(status, BH_ST_CAND_OVERFLOW); return; }      // cannot happen: the host sizes sr for the scale factor
    // ---- level 0 rows ra .. rb: gray + mask bits -> HBM and LDS
    const int groups = S.pitch >> 2;                                       // a multiple of 16: the 16 lanes of a DPP row share an image row
    for (int idx = tid; idx < nr * groups; idx += 256) {
        int r;
        {
            const int q = idx >> 4, d = groups >> 4;
            r = (int)((float)q * __builtin_amdgcn_rcpf((float)d));
            const int rem = q - __mul24(r, d);
            r += rem >= d ? 1 : (rem < 0 ? -1 : 0);
        }
        const int x = (idx - __mul24(r, groups)) << 2, y = ra + r;
        uint32_t g4 = 0, nib = 0;
        if (x < W) gray_item4(fimg, fd, W, y, x, g4, nib);
        *reinterpret_cast<uint32_t*>(fp + (__umul24((uint32_t)y, (uint32_t)S.pitch) + (uint32_t)x)) = g4;
        *reinterpret_cast<uint32_t*>(sg + __umul24((uint32_t)r, (uint32_t)gp) + (uint32_t)x) = g4;
        const uint32_t wd = nib_gather8(nib, tid);
        if ((tid & 7) == 7) { fm[__umul24((uint32_t)y, (uint32_t)S.mpitch) + (uint32_t)(x >> 5)] = wd; sm[__umul24((uint32_t)r, (uint32_t)mp) + (uint32_t)(x >> 5)] = wd; }
    }
    for (int i = tid; i < nr * 6; i += 256) {                              // the words behind a row's last column group / mask word: zero (only padding columns meet them)
        const int r = i / 6, k = i - r * 6;
        if (k < 4) *reinterpret_cast<uint32_t*>(sg + __umul24((uint32_t)r, (uint32_t)gp) + (uint32_t)S.pitch + 4u * k) = 0u;
        else sm[__umul24((uint32_t)r, (uint32_t)mp) + (uint32_t)(S.pitch >> 5) + (uint32_t)(k - 4)] = 0u;
    }
    __syncthreads();
    // ---- level 1 rows y1a .. y1b - 1 from the LDS rows (two rows x four pixels per lane item, as k_pyr_down_bits)
    uint8_t* __restrict__ dst = fp + D.off;
    uint32_t* __restrict__ md = fm + D.moff;
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_md = __builtin_amdgcn_make_buffer_rsrc(md, 0, 0x7fffffff, 0x00020000);
    const int g1 = D.pitch >> 2, nblk = (y1b - y1a + GL_PR - 1) / GL_PR;
    for (int idx = tid; idx < nblk * g1; idx += 256) {
        int yb;
        {
            const int q = idx >> 4, d = g1 >> 4;
            yb = (int)((float)q * __builtin_amdgcn_rcpf((float)d));
            const int rem = q - __mul24(yb, d);
            yb += rem >= d ? 1 : (rem < 0 ? -1 : 0);
        }
        const int x = (idx - __mul24(yb, g1)) << 2, y0 = y1a + GL_PR * yb;
        pyr_xs X;
        pyr_xsetup(X, D, tab, x);                                          // once per GL_PR rows (LDS rows: no load latency to spread over more lanes)
        const int od = (int)(__umul24((uint32_t)y0, (uint32_t)D.pitch) + (uint32_t)x), om = (int)((__umul24((uint32_t)y0, (uint32_t)D.mpitch) + (uint32_t)(x >> 5)) << 2);
#pragma unroll
        for (int k = 0; k < GL_PR; k++) {
            const uint32_t ty = tab[D.taby_off + min(y0 + k, D.h - 1)];
            const int r0 = (int)(ty >> 16), r1 = min(r0 + 1, S.h - 1);
            pyr_ld L;
            pyr_load_lds(L, sg, sm, gp, mp, X, min(r0, rb) - ra, min(r1, rb) - ra);
            uint32_t o4, nbw;
            pyr_row(L, X, ty & 0xffffu, tid, &o4, &nbw);
            if (y0 + k < y1b) {
                __builtin_amdgcn_raw_buffer_store_b32(o4, rs_dst, od, k * D.pitch, 0);
                if ((tid & 7) == 7) __builtin_amdgcn_raw_buffer_store_b32(nbw, rs_md, om, k * D.mpitch * 4, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ levels l and l + 1 in one launch
// k_pyr_pair (round 5): what k_gray_l1 does for levels 0 / 1, for any later pair.  The six launches of levels 2 .. 7 are the front end's
// shortest kernels (6 - 20 us each, 48 per 1 000 frames) and each re-reads the level the one before has just written; here a workgroup
// takes a strip of `sr` rows of level l + 1,
//   1. resizes the <= GL_R0 rows of level l they read from level l - 1 in HBM (pyr_xsetup / pyr_load / pyr_row, two rows per lane item:
//      the statements of k_pyr_down_bits, hence the same bytes), writes them to HBM (rows two strips share are written twice, identically)
//      and keeps them with their mask words in LDS;
//   2. resizes level l + 1 from LDS exactly as k_gray_l1's second phase does.
// Three launches instead of six for an 8-level pyramid behind k_gray_l1, and levels 2 / 4 / 6 are not read back.  Measured at configs[1]:
// 0.674 -> 0.668 ms of kernel time per 1 000 frames, the step unchanged (317.9 / 317.8 against 316.0 / 317.1 k frames/s in an A / B / A / B run):
// what the six launches cost is their vector instructions and the latency of each workgroup's own load -> compute -> store chain, not the
// launch boundaries - so it runs on request only (BH_ORB_PYR_PAIR=1), tested like k_pyr_mfma.
__global__ __launch_bounds__(256) void k_pyr_pair(uint8_t* __restrict__ pyr, bh_orb_geom g, int level, int sr, const uint32_t* __restrict__ tab,
                                                  uint32_t* __restrict__ mbits, uint32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t gl_sm[];
    const bh_level_geom Z = g.lv[level - 1], S = g.lv[level], D = g.lv[level + 1];
    uint8_t* const sg = gl_sm;                                             // [GL_R0][S.pitch + 16] rows of level l
    const int gp = S.pitch + 16;
    uint32_t* const sm = reinterpret_cast<uint32_t*>(gl_sm + (size_t)GL_R0 * gp);      // [GL_R0][S.mpitch + 1] their mask words
    const int mp = S.mpitch + 1;
    const int tid = threadIdx.x;
    const int y1a = blockIdx.x * sr, y1b = min(D.h, y1a + sr);
    uint8_t* __restrict__ fp = pyr + (size_t)blockIdx.y * g.pyr_stride;
    uint32_t* __restrict__ fm = mbits + (size_t)blockIdx.y * g.mstride;
    const int ra = (int)(tab[D.taby_off + y1a] >> 16);
    int rb = min((int)(tab[D.taby_off + y1b - 1] >> 16) + 1, S.h - 1);
    if (blockIdx.x == gridDim.x - 1) rb = S.h - 1;                          // (the last strip finishes level l whatever the last source row is)
    const int nr = rb - ra + 1;
    if (nr > GL_R0) { if (tid == 0) atomicOr(status, BH_ST_CAND_OVERFLOW); return; }      // cannot happen: the host sizes sr from the row table
    // ---- level l rows ra .. rb from level l - 1 -> HBM and LDS
    {
        const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(fp + Z.off, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_ms = __builtin_amdgcn_make_buffer_rsrc(fm + Z.moff, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(fp + S.off, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_md = __builtin_amdgcn_make_buffer_rsrc(fm + S.moff, 0, 0x7fffffff, 0x00020000);
        const int groups = S.pitch >> 2, nblk = (nr + PD_R - 1) / PD_R;    // groups: a multiple of 16, the 16 lanes of a DPP row share their rows
        for (int idx = tid; idx < nblk * groups; idx += 256) {
            const int yb = pyr_div_groups(idx, groups);
            const int x = (idx - __mul24(yb, groups)) << 2, y0 = ra + yb * PD_R;
            pyr_xs X;
            pyr_xsetup(X, S, tab, x);
            pyr_ld L[PD_R]; uint32_t yc[PD_R];
#pragma unroll
            for (int k = 0; k < PD_R; k++) {
                const uint32_t ty = tab[S.taby_off + min(y0 + k, S.h - 1)];
                const int r0 = (int)(ty >> 16), r1 = min(r0 + 1, Z.h - 1);
                yc[k] = ty & 0xffffu;
                pyr_load(L[k], rs_src, rs_ms, Z, X, r0, r1);
            }
            uint32_t o4[PD_R], nbw[PD_R];
#pragma unroll
            for (int k = 0; k < PD_R; k++) pyr_row(L[k], X, yc[k], tid, &o4[k], &nbw[k]);
            const int od = (int)(__umul24((uint32_t)y0, (uint32_t)S.pitch) + (uint32_t)x), om = (int)((__umul24((uint32_t)y0, (uint32_t)S.mpitch) + (uint32_t)(x >> 5)) << 2);
#pragma unroll
            for (int k = 0; k < PD_R; k++) {
                if (y0 + k <= rb) {
                    const int r = y0 + k - ra;
                    __builtin_amdgcn_raw_buffer_store_b32(o4[k], rs_dst, od, k * S.pitch, 0);
                    *reinterpret_cast<uint32_t*>(sg + __umul24((uint32_t)r, (uint32_t)gp) + (uint32_t)x) = o4[k];
                    if ((tid & 7) == 7) {
                        __builtin_amdgcn_raw_buffer_store_b32(nbw[k], rs_md, om, k * S.mpitch * 4, 0);
                        sm[__umul24((uint32_t)r, (uint32_t)mp) + (uint32_t)(x >> 5)] = nbw[k];
                    }
                }
            }
        }
    }
    for (int i = tid; i < nr * 6; i += 256) {                              // the words behind a row's last column group / mask word: zero (only padding columns meet them)
        const int r = i / 6, k = i - r * 6;
        if (k < 4) *reinterpret_cast<uint32_t*>(sg + __umul24((uint32_t)r, (uint32_t)gp) + (uint32_t)S.pitch + 4u * k) = 0u;
        else sm[__umul24((uint32_t)r, (uint32_t)mp) + (uint32_t)(S.pitch >> 5) + (uint32_t)(k - 4)] = 0u;
    }
    __syncthreads();
    // ---- level l + 1 rows y1a .. y1b - 1 from the LDS rows (the second phase of k_gray_l1)
    uint8_t* __restrict__ dst = fp + D.off;
    uint32_t* __restrict__ md = fm + D.moff;
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_md = __builtin_amdgcn_make_buffer_rsrc(md, 0, 0x7fffffff, 0x00020000);
    const int g1 = D.pitch >> 2, nblk = (y1b - y1a + GL_PR - 1) / GL_PR;
    for (int idx = tid; idx < nblk * g1; idx += 256) {
        const int yb = pyr_div_groups(idx, g1);
        const int x = (idx - __mul24(yb, g1)) << 2, y0 = y1a + GL_PR * yb;
        pyr_xs X;
        pyr_xsetup(X, D, tab, x);
        const int od = (int)(__umul24((uint32_t)y0, (uint32_t)D.pitch) + (uint32_t)x), om = (int)((__umul24((uint32_t)y0, (uint32_t)D.mpitch) + (uint32_t)(x >> 5)) << 2);
#pragma unroll
        for (int k = 0; k < GL_PR; k++) {
            const uint32_t ty = tab[D.taby_off + min(y0 + k, D.h - 1)];
            const int r0 = (int)(ty >> 16), r1 = min(r0 + 1, S.h - 1);
            pyr_ld L;
            pyr_load_lds(L, sg, sm, gp, mp, X, min(r0, rb) - ra, min(r1, rb) - ra);
            uint32_t o4, nbw;
            pyr_row(L, X, ty & 0xffffu, tid, &o4, &nbw);
            if (y0 + k < y1b) {
                __builtin_amdgcn_raw_buffer_store_b32(o4, rs_dst, od, k * D.pitch, 0);
                if ((tid & 7) == 7) __builtin_amdgcn_raw_buffer_store_b32(nbw, rs_md, om, k * D.mpitch * 4, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ pyramid level l from l-1 on the matrix cores
// The resize is a separable linear map with at most two taps per output and axis, i.e. two products with banded matrices, and the
// arithmetic of INTER_LINEAR_EXACT (8.8 horizontal, 16.16 vertical, one rounding at the end) is exact in int32 in any order.  The
// vector-ALU form above spends ~0.53 wave-instructions per output pixel (the front end is bound by vector issue, tools/valu_rate.hip);
// here one wave makes a 32-column x 24-row output block with 8 v_mfma_i32_32x32x32_i8:
//   pass H   Hs[r][c'] = sum_k (-cx[k][c']) (127 - px[r][k]) - 1 = H - 32513  for the block's 32 source rows (24 x 1.3 + 2 <= 32) and a
//            64-byte source window starting at a multiple of 16 (32 outputs span <= 43 columns): A = the rows' bytes as they lie in
//            memory (xor 0x7f), B = tap fragments built on the host per column block.  A tap is 0 .. 256 - nine bits - so it enters as
//            two int8 halves (two products into one accumulator), and NEGATED, because 256 = 128 + 128 has no int8 form but -256 has;
//            the operand is negated with it (127 - px = -(px - 128) - 1, the -1 is a constant since the taps of an output sum to 256).
//   pass V   the accumulator tile is the next product's A operand (k = source row, in accumulator order; the host builds the row taps
//            in that order): Hs = 256 hi + lo8 -> (-hi - 1), (127 - lo8) as int8, 256 x (sum over hi) + constant is the C operand of
//            the lo product, whose result is the rounded sum; byte 2 is the pixel.
// The bit-packed mask (binary: depth > 0) follows the reference's resize + threshold(254) by bit logic: away from a sparse, static set
// of "special" positions the output bit is the AND of its four taps - per 32-pixel output word: AND the two taps' windows, then
// gather the bits by the runs of equal (source - destination) distance (six shift-and-mask steps per 16 outputs, masks from the
// host).  Where a tap weighs <= 128 of 65 536 (3.4 % of the positions; a property of the coefficients alone) the reference result is a
// 16-entry truth table of the four taps, built on the host: those positions are patched in LDS from a per-tile list.
#define PM_TW 128
#define PM_TH 24

#define PM_MH 64             // rows of a mask workgroup (4 words x 64 rows: one 32-pixel output word per lane)
#define PM_MSR 88            // source mask rows staged for them (64 x 1.3 + 2 <= 88)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_pyr_mfma(uint8_t* __restrict__ pyr, bh_orb_geom g, int level,
                                                  const uint32_t* __restrict__ tab, const uint32_t* __restrict__ rtab, bh_pm_level T, uint32_t* __restrict__ mbits)
{
    typedef int fv4 __attribute__((ext_vector_type(4)));
    typedef int fv16 __attribute__((ext_vector_type(16)));
    __shared__ uint32_t s_o[4 * 32 * 9];                          // image: per wave 32 rows x (8 + 1) words; mask: the staged source rows (88 x 8)
    __shared__ __attribute__((aligned(16))) uint32_t s_mt[64];    // mask: run masks of the tile's four output words
    __shared__ uint32_t s_m[256];                                 // mask: the tile's output words
    static_assert(PM_MSR * 8 <= 4 * 32 * 9, "the mask rows share the image's LDS");
    const bh_level_geom S = g.lv[level - 1], D = g.lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, h = lane >> 5;
    const int x0 = blockIdx.x * PM_TW;

    if ((int)blockIdx.y < T.tiles_y) {
        // ---- image: a wave makes its 32 x 24 block alone - no workgroup barrier, the transposition for the row stores goes through a
        // wave-private piece of LDS.  Every global load is issued before the first use.
        const int y0 = blockIdx.y * PM_TH;
        if (!(x0 + 32 * wv < D.w)) return;                        // wave-uniform
        const uint8_t* __restrict__ src = pyr + (size_t)blockIdx.z * g.pyr_stride + S.off;
        uint8_t* __restrict__ dst = pyr + (size_t)blockIdx.z * g.pyr_stride + D.off;
        const int cb = blockIdx.x * 4 + wv;
        const int r0 = (int)rtab[T.yinfo + blockIdx.y], ws = (int)rtab[T.xinfo + cb];
        const uint8_t* rp = src + (uint32_t)(min(r0 + n, S.h - 1) * S.pitch + ws + 16 * h);
        fv4 a0 = *reinterpret_cast<const fv4*>(rp), a1 = *reinterpret_cast<const fv4*>(rp + 32);
        const fv4* xt = reinterpret_cast<const fv4*>(rtab + (uint32_t)(T.xtap + 1024 * cb)) + lane;      // [column block][k-block][half][lane]
        const fv4 X00 = xt[0], X01 = xt[64], X10 = xt[128], X11 = xt[192];
        const fv4* yt = reinterpret_cast<const fv4*>(rtab + (uint32_t)(T.ytap + 512 * blockIdx.y)) + lane;      // [row block][half][lane]
        const fv4 Y0 = yt[0], Y1 = yt[64];
        a0 ^= (int)0x7f7f7f7fu; a1 ^= (int)0x7f7f7f7fu;
        fv16 acc = { -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1 };
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, X00, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, X01, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, X10, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, X11, acc, 0, 0, 0);
        fv4 Al, Ah;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t q01 = __builtin_amdgcn_perm((uint32_t)acc[4 * d + 1], (uint32_t)acc[4 * d], 0x05010400u);      // (r0.b0, r1.b0, r0.b1, r1.b1)
            const uint32_t q23 = __builtin_amdgcn_perm((uint32_t)acc[4 * d + 3], (uint32_t)acc[4 * d + 2], 0x05010400u);
            Al[d] = (int)(__builtin_amdgcn_perm(q23, q01, 0x05040100u) ^ 0x7f7f7f7fu);      // 127 - lo8
            Ah[d] = (int)~__builtin_amdgcn_perm(q23, q01, 0x07060302u);                      // -hi - 1
        }
        fv16 accH = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        accH = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, Y0, accH, 0, 0, 0);
        accH = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, Y1, accH, 0, 0, 0);
        const int KC = 256 * 32384 + 32768;                        // 256 (-256 + 127 + 32513) + the rounding
        fv16 accL;
#pragma unroll
        for (int i = 0; i < 16; i++) accL[i] = (accH[i] << 8) + KC;
        accL = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, Y0, accL, 0, 0, 0);
        accL = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, Y1, accL, 0, 0, 0);
        // lane = output row n; registers 4 q .. 4 q + 3 = columns 8 q + 4 h + (0..3) of the block: byte 2 of the sums
        uint32_t* so = s_o + wv * (32 * 9);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)accL[4 * q + 1], (uint32_t)accL[4 * q], 0x0c0c0602u);
            const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)accL[4 * q + 3], (uint32_t)accL[4 * q + 2], 0x0c0c0602u);
            so[n * 9 + 2 * q + h] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // LDS exchange inside one wave: ordered by the hardware, the fences keep the compiler from moving the reads up
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {   // 32-byte row segments (the four waves of the workgroup fill the 128-byte lines together); padding columns stay zero
            const int wd = lane & 7, r0s = lane >> 3, x = x0 + 32 * wv + 4 * wd;
            const int over = min(max(x + 4 - D.w, 0), 4);
            const uint32_t keep = over == 4 ? 0u : (0xffffffffu >> (8 * over));
            uint8_t* bt = dst + (uint32_t)(y0 * D.pitch + x0 + 32 * wv);
            const uint32_t o0 = (uint32_t)(r0s * D.pitch + 4 * wd);
#pragma unroll
            for (int k = 0; k < PM_TH / 8; k++) {
                const int rr = r0s + 8 * k;
                if (x < D.pitch && y0 + rr < D.h) *reinterpret_cast<uint32_t*>(bt + (o0 + (uint32_t)(8 * k * D.pitch))) = so[rr * 9 + wd] & keep;
            }
        }
        return;
    }

    // ---- mask: 4 output words x 64 rows per workgroup, one word per lane.  The source rows the block needs (<= 88, 8 words from the
    // tile's first source word) are staged in LDS by coalesced loads; the per-lane windows and the special positions read LDS.
    uint32_t* s_ms = s_o;
    const uint32_t* __restrict__ ms = mbits + (size_t)blockIdx.z * g.mstride + S.moff;
    uint32_t* __restrict__ md = mbits + (size_t)blockIdx.z * g.mstride + D.moff;
    const int by = blockIdx.y - T.tiles_y, y0 = by * PM_MH;
    const int r0 = (int)(tab[D.taby_off + y0] >> 16), mwb = (int)rtab[T.mwb + blockIdx.x];
    const int tile = by * T.tiles_x + blockIdx.x;
    const int f0 = (int)rtab[T.fidx + tile], f1 = (int)rtab[T.fidx + tile + 1];
    // special positions of this block: entry = (x in tile | y in block << 7 | truth table << 13, source column | source row << 16)
    uint2 fe0 = make_uint2(0, 0), fe1 = fe0;
    if (f0 + tid < f1) fe0 = *reinterpret_cast<const uint2*>(rtab + (uint32_t)(T.fix + 2 * (f0 + tid)));
    if (f0 + 256 + tid < f1) fe1 = *reinterpret_cast<const uint2*>(rtab + (uint32_t)(T.fix + 2 * (f0 + 256 + tid)));
    uint32_t msw[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int idx = min(tid + 256 * k, PM_MSR * 8 - 1);
        msw[k] = ms[(uint32_t)(min(r0 + (idx >> 3), S.h - 1) * S.mpitch + min(mwb + (idx & 7), S.mpitch - 1))];
    }
    uint32_t mtw = 0;
    if (tid < 64) mtw = rtab[(uint32_t)(T.mtab + 64 * blockIdx.x) + tid];
    const int mrow = tid >> 2, mw = tid & 3, mX = x0 + 32 * mw, my = y0 + mrow;
    const bool mact = mX < D.pitch && my < D.h;
    const uint32_t mty = tab[(uint32_t)(D.taby_off + min(my, D.h - 1))];
#pragma unroll
    for (int k = 0; k < 3; k++) if (tid + 256 * k < PM_MSR * 8) s_ms[tid + 256 * k] = msw[k];
    if (tid < 64) s_mt[tid] = mtw;
    __syncthreads();
    {
        const uint32_t* pa = s_ms + ((int)(mty >> 16) - r0) * 8 - mwb;
        const uint4* mt = reinterpret_cast<const uint4*>(s_mt + 16 * mw);
        const uint4 t0 = mt[0], t1 = mt[1], t2 = mt[2], t3 = mt[3];      // s0 s1 m0[0] m0[1] | m0[2..5] | m1[0..3] | m1[4] m1[5] - -
        uint32_t res[2];
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const uint32_t st = hf ? t0.y : t0.x, wi = st >> 5, sh = st & 31u;
            const uint32_t wa = __builtin_amdgcn_alignbit(pa[wi + 1], pa[wi], sh), wb = __builtin_amdgcn_alignbit(pa[wi + 9], pa[wi + 8], sh);
            const uint32_t Q = wa & (wa >> 1) & wb & (wb >> 1);
            const uint32_t m0 = hf ? t2.x : t0.z, m1 = hf ? t2.y : t0.w, m2 = hf ? t2.z : t1.x, m3 = hf ? t2.w : t1.y, m4 = hf ? t3.x : t1.z, m5 = hf ? t3.y : t1.w;
            res[hf] = (Q & m0) | ((Q >> 1) & m1) | ((Q >> 2) & m2) | ((Q >> 3) & m3) | ((Q >> 4) & m4) | ((Q >> 5) & m5);
        }
        s_m[tid] = res[0] | (res[1] << 16);
    }
    __syncthreads();
    for (int i = f0 + tid, k = 0; i < f1; i += 256, k++) {
        uint2 e = k == 0 ? fe0 : fe1;
        if (k >= 2) e = *reinterpret_cast<const uint2*>(rtab + (uint32_t)(T.fix + 2 * i));
        const int xr = e.x & 127u, yr = (e.x >> 7) & 63u;
        const int ox = e.y & 0xffffu, ri = (int)(e.y >> 16) - r0, wi = (ox >> 5) - mwb;
        const uint32_t* pa = s_ms + (ri * 8 + wi);
        const uint32_t a2 = __builtin_amdgcn_alignbit(pa[1], pa[0], ox & 31) & 3u, b2 = __builtin_amdgcn_alignbit(pa[9], pa[8], ox & 31) & 3u;
        uint32_t* wp = &s_m[yr * 4 + (xr >> 5)];
        if (((e.x >> 13) >> (a2 | (b2 << 2))) & 1u) atomicOr(wp, 1u << (xr & 31)); else atomicAnd(wp, ~(1u << (xr & 31)));
    }
    __syncthreads();
    if (mact) md[(uint32_t)(my * D.mpitch + (mX >> 5))] = s_m[tid];
}

__global__ __launch_bounds__(256) void k_pyr_down(uint8_t* __restrict__ pyr, uint8_t* __restrict__ mpyr,
                                                  bh_orb_geom g, int level, const uint32_t* __restrict__ tab)
{
    const bh_level_geom S = g.lv[level - 1], D = g.lv[level];
    const int groups = D.pitch >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int y = idx / groups, x = (idx - y * groups) << 2;
    if (y >= D.h) return;                                      // whole rows (pitch / 4 is a multiple of 16 lanes)
    const size_t fb = (size_t)blockIdx.y * g.pyr_stride;
    const size_t o = fb + D.off + (size_t)y * D.pitch + x;
    if (x >= D.w) {                                            // padding columns: keep them zero
        *reinterpret_cast<uint32_t*>(pyr + o) = 0;
        if (mpyr) *reinterpret_cast<uint32_t*>(mpyr + o) = 0;
        return;
    }
    const uint4 tx4 = *reinterpret_cast<const uint4*>(tab + D.tabx_off + x);      // tabx_off is a multiple of 4, padded
    const uint32_t ty = tab[D.taby_off + y];
    const int r0 = ty >> 16, r1 = min(r0 + 1, S.h - 1);
    const uint32_t yc1 = ty & 0xffffu;
    const uint32_t txs[4] = { tx4.x, tx4.y, tx4.z, tx4.w };
    // the 4 outputs reference source columns a0 .. a0+5 (scale <= 1.3): one 8-byte window per source row and plane
    const int base = (int)(tx4.x >> 16);
    const size_t s0 = fb + S.off + (size_t)r0 * S.pitch + base, s1 = fb + S.off + (size_t)r1 * S.pitch + base;
    const unsigned long long A = pyr_load8(pyr, s0), Bv = pyr_load8(pyr, s1);
    unsigned long long MA = 0, MB = 0;
    if (mpyr) { MA = pyr_load8(mpyr, s0); MB = pyr_load8(mpyr, s1); }
    uint32_t o4 = 0, mo4 = 0;
    if (mpyr) {
        // image and mask ride in the two 16-bit halves of one register through the horizontal pass: (p0 << 8) + c1*(p1 - p0) stays
        // below 2^16, so modular 16-bit arithmetic (v_pk_sub/mul/add_u16) is exact; v_perm_b32 builds the (image, mask) byte pairs
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (x + i < D.w) {
                const uint32_t tx = txs[i];
                const int sh = ((int)(tx >> 16) - base) << 3;
                const uint32_t c1 = tx & 0xffffu;
                const uint32_t pa = (uint32_t)(A >> sh), pb = (uint32_t)(Bv >> sh), ma = (uint32_t)(MA >> sh), mb = (uint32_t)(MB >> sh);
                const uint32_t P0 = __builtin_amdgcn_perm(ma, pa, 0x0c040c00u), P1 = __builtin_amdgcn_perm(ma, pa, 0x0c050c01u);
                const uint32_t Q0 = __builtin_amdgcn_perm(mb, pb, 0x0c040c00u), Q1 = __builtin_amdgcn_perm(mb, pb, 0x0c050c01u);
                const us2 C1 = { (unsigned short)c1, (unsigned short)c1 };
                const us2 p0v = __builtin_bit_cast(us2, P0), p1v = __builtin_bit_cast(us2, P1), q0v = __builtin_bit_cast(us2, Q0), q1v = __builtin_bit_cast(us2, Q1);
                const us2 h0v = (us2)(p0v << (us2){ 8, 8 }) + C1 * (p1v - p0v);
                const us2 h1v = (us2)(q0v << (us2){ 8, 8 }) + C1 * (q1v - q0v);
                const int h0 = h0v.x, h1 = h1v.x, g0 = h0v.y, g1 = h1v.y;
                const uint32_t v = (uint32_t)((h0 << 8) + (int)yc1 * (h1 - h0) + 32768) >> 16;
                const uint32_t w = (uint32_t)((g0 << 8) + (int)yc1 * (g1 - g0) + 32768) >> 16;
                o4 |= v << (8 * i);
                mo4 |= (w > 254u ? w : 0u) << (8 * i);                 // THRESH_TOZERO at 254
            }
        }
        *reinterpret_cast<uint32_t*>(pyr + o) = o4;
        *reinterpret_cast<uint32_t*>(mpyr + o) = mo4;
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (x + i < D.w) {
            const uint32_t tx = txs[i];
            const int sh = ((int)(tx >> 16) - base) << 3;       // 0 .. 32 bits
            const uint32_t c1 = tx & 0xffffu;
            // c0*p0 + c1*p1 with c0 = 256 - c1  ==  (p0 << 8) + c1*(p1 - p0); same for the vertical pass (exact integers)
            uint32_t pa = (uint32_t)(A >> sh), pb = (uint32_t)(Bv >> sh);
            int p0 = (int)(pa & 255u), q0 = (int)(pb & 255u);
            int h0 = (p0 << 8) + (int)c1 * ((int)((pa >> 8) & 255u) - p0);
            int h1 = (q0 << 8) + (int)c1 * ((int)((pb >> 8) & 255u) - q0);
            uint32_t v = (uint32_t)((h0 << 8) + (int)yc1 * (h1 - h0) + 32768) >> 16;
            o4 |= v << (8 * i);
            if (mpyr) {
                pa = (uint32_t)(MA >> sh); pb = (uint32_t)(MB >> sh);
                p0 = (int)(pa & 255u); q0 = (int)(pb & 255u);
                h0 = (p0 << 8) + (int)c1 * ((int)((pa >> 8) & 255u) - p0);
                h1 = (q0 << 8) + (int)c1 * ((int)((pb >> 8) & 255u) - q0);
                v = (uint32_t)((h0 << 8) + (int)yc1 * (h1 - h0) + 32768) >> 16;
                mo4 |= (v > 254u ? v : 0u) << (8 * i);                 // THRESH_TOZERO at 254
            }
        }
    }
    *reinterpret_cast<uint32_t*>(pyr + o) = o4;
    if (mpyr) *reinterpret_cast<uint32_t*>(mpyr + o) = mo4;
}

// ------------------------------------------------------------------------------------------------ tile decode
__device__ __forceinline__ int tile_level(const bh_orb_geom& g, int tile)
{
    int l = 0;
    for (int i = 1; i < g.nlevels; i++) if (tile >= g.lv[i].tile_base) l = i;
    return l;
}

// ------------------------------------------------------------------------------------------------ FAST-9/16 + NMS + 7x7 blur

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (i < 0) i = -i; else if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}

// Tile = 128 x 32 pixels.  LDS pixel tile: rows y0-4 .. y0+35, pixels x0-4 .. x0+131 as 34 aligned words per row, stored with a
// pitch of 36 words (one pad word on either side, so word+-1 never leaves the row).  The score tile (rows y0-1 .. y0+32) has the
// same word geometry, so one index serves both: item i = score row * 36 + word, its pixel row starts 3 rows further down.
#define FPX_H (TILE_H + 8)
#define FRW   (TILE_W / 4 + 2)  // real words per row (34)
#define FPW   (FRW + 2)         // LDS pitch in words (36)
#define FSC_H (TILE_H + 2)
#define FROW  (4 * FPW)         // bytes per LDS row (144)
#define FITEMS (FSC_H * FPW)    // 1224 (item, word) pairs of the score tile
#define FLIST_CAP (2 * FITEMS)   // candidate list entries: half of the score tile's pixels
#define FCL_CAP 1536             // corner list entries (more: phase 3 walks the candidate list instead)

typedef short fs2 __attribute__((ext_vector_type(2)));
// byte index of the score tile -> row: b / 144 for b < 2^15 as one 24-bit multiply and a shift (14564 = ceil(2^21 / 144): the error b x 0.44 / 2^21 stays
// below 1 / 144 there; the compiler's division by a constant is a 32-bit multiply-high, a quarter-rate instruction)
static_assert(FROW == 144, "frow_div assumes 144-byte tile rows");
__device__ __forceinline__ int frow_div(int b) { return (int)(__umul24((uint32_t)b, 14564u) >> 21); }
__device__ __forceinline__ fs2 fswap(fs2 a) { return fs2{ a.y, a.x }; }
__device__ __forceinline__ fs2 fmn(fs2 a, fs2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ fs2 fmx(fs2 a, fs2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ fs2 fmn3(fs2 a, fs2 b, fs2 c) { fs2 r; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ fs2 fmx3(fs2 a, fs2 b, fs2 c) { fs2 r; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// LDS fetch-add issued as ONE ds_add_rtn_u32 for all active lanes (the hardware serialises lanes that hit the same address).  The
// compiler's atomic optimiser would turn a divergent-value atomicAdd into a scalar loop over the active lanes (~8 SALU each).
__device__ __forceinline__ int lds_add_rtn(int* p, int v)
{
    int old;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)p;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(a), "v"(v) : "memory");
    return old;
}
__device__ __forceinline__ fs2 fpk(uint32_t hi_src, uint32_t lo_src, uint32_t sel) { return __builtin_bit_cast(fs2, __builtin_amdgcn_perm(hi_src, lo_src, sel)); }

// One workgroup per 128x32 tile of one pyramid level of one frame.
//   phase 1  the reference detector's opposite-pair pretest (a 9-arc contains ring pixel k or k+8, so both pairs (0,8) and (4,12)
//            must hold a darker -- or both a brighter -- pixel) on FOUR adjacent pixels per lane: the five operands (centre, up, down,
//            left, right) come from aligned LDS words through v_perm into packed 16-bit lanes, v_pk_min/max/add/sub_i16 do two pixels
//            per instruction, the sign bits are the verdict.  All 256 lanes; survivors (a few %) go to an LDS list.
//   phase 2  full 16-pixel ring on the survivors: score = largest threshold at which the pixel is still a corner, by a packed
//            doubling min/max tree over the 16 arcs; corners write their score into a sparse LDS score tile.
//   phase 3  the same list again: strict 3x3 NMS against the score tile, border and mask filter; kept corners are appended to the
//            level's corner list in HBM (one reservation per tile).  No dense score map exists; the score histogram retainBest needs is
//            built by k_select from the list (round 2 added ~60 global atomics per tile for it here).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fast(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ mpyr,
                                              uint2* __restrict__ clist, int* __restrict__ ccnt,
                                              uint8_t* __restrict__ blur,
                                              bh_orb_geom g, const uint32_t* __restrict__ tile_tab, uint32_t* __restrict__ status,
                                              const uint32_t* __restrict__ mbits, const uint32_t* __restrict__ gtab)
{
    // 17.3 KB of LDS: eight workgroups per CU.  (26.2 KB allowed six, and the kernel is bound by the latency of its barrier-separated
    // phases as much as by its instruction count: 2.18 -> 1.98 ms per 1000 frames.)  The candidate list holds HALF the score tile -
    // a tile where more than half of the pixels pass the pretest (noise) takes the exact list-free path below - and the kept-corner
    // staging array reuses the pixel tile, which is dead once phase 2 and the blur are through.
    __shared__ __attribute__((aligned(16))) uint32_t s_px[FPX_H * FPW + 8];      // +8: the last k-block of the blur reads 15 bytes past the last row (zero taps)
    __shared__ uint32_t s_mask[TILE_H * (TILE_W / 32)];         // the tile of the bit-packed mask (no HBM round trip in phase 3)
    __shared__ __attribute__((aligned(16))) uint32_t s_scw[FITEMS];
    __shared__ __attribute__((aligned(16))) uint32_t s_k1[4];   // 16 constant bytes: the blur's "+128" rides through spare k slots of its second product
    __shared__ __attribute__((aligned(16))) uint16_t s_list[FLIST_CAP];      // (its space stages the kept corners once phase 2 is through)
    __shared__ uint32_t s_vm[FPW];
    __shared__ uint16_t s_cl[FCL_CAP];                          // the candidates phase 2 found to be corners: phase 3 walks these, not the whole candidate list
    __shared__ int s_n, s_k, s_base, s_c;
#ifdef FAST_LDS_PAD                                              // diagnostic builds: fewer workgroups per CU (does a co-running kernel of another stream fill the gap?)
    __shared__ uint32_t s_pad[FAST_LDS_PAD];
    if (status[0] == 0xfffffff1u) { s_pad[threadIdx.x] = clist[threadIdx.x].x; __syncthreads(); if (s_pad[(threadIdx.x * 7) % FAST_LDS_PAD] == 7u) status[1] = 1; }
#endif

    const int tid = threadIdx.x;
    // Which (tile, frame): workgroups b and b + 8 of the launch order share an XCD (observed placement; speed only).  The tiles of ONE frame go to
    // workgroups of equal b mod 8, in consecutive launch positions: x-adjacent tiles read the two halves of the same 128-B lines (a tile row is 134
    // bytes at any alignment) and y-adjacent ones share six halo rows - dealt tile by tile over the XCDs every such line was fetched into two or three L2s.
    int ti = blockIdx.x, frame = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, grp = lin / (8 * (int)gridDim.x), r = lin - grp * 8 * (int)gridDim.x;
        frame = grp * 8 + (r & 7); ti = r >> 3;
    }
    const uint32_t tt = tile_tab[ti];                           // level | tile column << 4 | tile row << 16 (host-built: no per-wave level search / division)
    const int level = tt & 15;
    const bh_level_geom L = g.lv[level];
    const int x0 = (int)((tt >> 4) & 0xfffu) * TILE_W, y0 = (int)(tt >> 16) * TILE_H;
    const size_t fbase = (size_t)frame * g.pyr_stride + L.off;
    const uint8_t* img = pyr + fbase;

    // The tile's halo outside the level image is filled by reflection (BORDER_REFLECT_101, what the 7x7 blur below needs; FAST never
    // scores a pixel whose ring leaves the image).  Words inside the image are one aligned load; the few that straddle or lie
    // just beyond the left / right border gather their 4 bytes; anything further than 8 px beyond the border is never read.
    // Lane -> (row r0 + 7 j, word c) with r0 = tid / 34, c = tid % 34: one division per lane instead of one per load (238 of the 256
    // lanes load; six rounds cover the 40 rows).
    constexpr int NLD = (FPX_H + 6) / 7;
    uint32_t pw[NLD];
    const int lr0 = tid / FRW, lc = tid - lr0 * FRW;
    const bool ldl = tid < 7 * FRW;
    const bool yin = y0 >= 4 && y0 + TILE_H + 4 <= L.h;          // rows y0-4 .. y0+35 all inside the level: no reflection per row
    const bool interior = yin && x0 >= 4 && x0 + TILE_W + 4 <= L.w;      // 37 % of the tiles at 640 x 480
    // Every lane loads from an address clamped into its row (level base + 32-bit offset: scalar base + vector offset form, no 64-bit
    // multiply per load, no predicate): a word further than 8 px beyond the border is never read with a non-zero weight, the words that
    // straddle the border are overwritten by the fix lanes below.
    // Buffer loads (round 4): the lane's row x pitch + column is ONE 24-bit multiply-add, the step of seven rows per load rides in the scalar
    // offset - as flat loads the compiler formed every address with a 64-bit multiply-add, a move and a 64-bit add (18 of them, the 64-bit
    // multiply at a quarter of the rate, in a kernel that is bound by vector issue).
    const uint32_t xwc = (uint32_t)min(max(x0 - 4 + 4 * lc, 0), L.pitch - 4);
    const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, 0x7fffffff, 0x00020000);
    if (yin) {
        const int vo = (int)(__umul24((uint32_t)(y0 - 4 + lr0), (uint32_t)L.pitch) + xwc);
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            if (7 * (NLD - 1) + 6 < FPX_H || j < NLD - 1) pw[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_img, vo, 7 * j * L.pitch, 0);
            else pw[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_img, (int)(__umul24((uint32_t)(y0 - 4 + min(lr0 + 7 * j, FPX_H - 1)), (uint32_t)L.pitch) + xwc), 0, 0);
        }
    } else
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int ry = reflect101(y0 - 4 + lr0 + 7 * j, L.h);
        pw[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_img, (int)(__umul24((uint32_t)ry, (uint32_t)L.pitch) + xwc), 0, 0);
    }
    // The words that straddle or lie just beyond the left / right border (one on the left, at most three on the right, per row) are
    // filled with reflection by 160 dedicated lanes, once: as a branch of the loop above, every wave of a border tile ran a 4-byte gather
    // for each of its six loads because some lane always needed it (~300 of the tile's ~1 000 instructions per wave).
    // Round 5: the reflected pixels of a row are pixels of the SAME row, up to 8 columns inside the border - bytes the loop above has just
    // put into the LDS row.  The fix lanes pick them up there after the barrier (four byte reads at tile positions p or A - p) instead of
    // reflecting four addresses and gathering four bytes from global memory: ~25 instead of ~80 instructions in three of the four waves
    // of 63 % of the tiles.  Only a tile whose last columns are narrower than the reflection reaches (fewer than 12 image columns in it, or a level
    // under 16 pixels wide) still gathers from global memory.
    const bool xedge = !interior && (x0 == 0 || x0 + TILE_W + 4 > L.w);
    const bool fix_lds = xedge && L.w >= 16 && (x0 + TILE_W + 4 <= L.w || L.w - x0 >= 12);
    uint32_t fixv = 0; int fixi = -1, fixc = 0;
    if (xedge && tid < 4 * FPX_H) {
        const int r = tid >> 2, q = tid & 3, y = y0 - 4 + r;
        const int c = q == 0 ? 0 : ((L.w - x0 + 4) >> 2) + q - 1, x = x0 - 4 + 4 * c;
        if (c < FRW && y < L.h + 8 && x < L.w + 8 && (x < 0 || x + 3 >= L.w)) {
            fixi = r * FPW + 1 + c; fixc = c;
            if (!fix_lds) {
                const uint8_t* row = img + (size_t)reflect101(y, L.h) * L.pitch;
#pragma unroll
                for (int k = 0; k < 4; k++) fixv |= (uint32_t)row[reflect101(x + k, L.w)] << (8 * k);
            }
        }
    }
    uint32_t mw = 0;
    if (mbits && tid < TILE_H * (TILE_W / 32)) {
        const int my = y0 + tid / (TILE_W / 32), mwi = (x0 >> 5) + (tid & (TILE_W / 32 - 1));
        if (my < L.h && mwi < (L.pitch >> 5)) mw = (mbits + (size_t)frame * g.mstride + L.moff)[__umul24((uint32_t)my, (uint32_t)L.mpitch) + (uint32_t)mwi];      // (uniform base, 24-bit row multiply)
    }
    const int thr = g.fast_thr, et = g.edge_thr;
    // Only pixels that can survive the border filter, or suppress a survivor in the 3x3 NMS, need a score:
    // [et-1, dim-et] intersected with the detector's own scan range [3, dim-4], and the tile's score columns x0-1 .. x0+128.
    const int xlo = max(max(3, et - 1), x0 - 1), xhi = min(min(L.w - 4, L.w - et), x0 + TILE_W);
    const int ylo = max(3, et - 1), yhi = min(L.h - 4, L.h - et);
    if (tid < TILE_H * (TILE_W / 32)) s_mask[tid] = mw;
    if (tid == 0) { s_n = 0; s_k = 0; s_c = 0; s_k1[0] = 0x80808181u; s_k1[1] = 0x80808080u; s_k1[2] = 0x80808080u; s_k1[3] = 0x80808080u; }
    if (tid < FPW) {                                             // per tile word: 0x80 in the bytes whose pixel is in [xlo, xhi]; pads 0
        const int xb = x0 - 8 + 4 * tid;
        const int lo = min(max(xlo - xb, 0), 4), hi = min(max(xhi - xb + 1, 0), 4);
        s_vm[tid] = (tid >= 1 && tid <= FRW) ? (uint32_t)((0x80808080ull << (8 * lo)) & (0x80808080ull >> (8 * (4 - hi)))) : 0u;
    }
    {
        static_assert(FITEMS % 4 == 0 && FITEMS / 4 <= 512, "score tile is zeroed by two 16-byte stores per lane");
        uint4* z = reinterpret_cast<uint4*>(s_scw);
        z[tid] = make_uint4(0, 0, 0, 0);
        if (tid < FITEMS / 4 - 256) z[256 + tid] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int r = lr0 + 7 * j;
        if (ldl && r < FPX_H) s_px[r * FPW + 1 + lc] = pw[j];
    }
    __syncthreads();
    if (xedge) {
        if (fixi >= 0) {
            if (fix_lds) {
                // tile position p = 4 c + k of the lane's word (0 = column x0 - 4): left of the image (only c = 0 of a tile with x0 = 0) the
                // reflection of p is 8 - p; right of it, with e = w - x0 + 4 the first position outside, 2 (e - 1) - p; positions inside stay.
                // The source bytes are pixels the aligned loads above wrote (a concurrent fix of a straddling word rewrites its inside bytes
                // with the same values).
                const int c = fixc, e = L.w - x0 + 4;
                const int A = c == 0 && x0 == 0 ? 8 : 2 * (e - 1), th = c == 0 && x0 == 0 ? 0 : e;
                const uint8_t* rowb = reinterpret_cast<const uint8_t*>(s_px) + (fixi - 1 - c) * 4 + 4;
#pragma unroll
                for (int k = 0; k < 4; k++) { const int p = 4 * c + k; fixv |= (uint32_t)rowb[p < th ? p : A - p] << (8 * k); }
            }
            s_px[fixi] = fixv;
        }
        __syncthreads();
    }

    const uint8_t* px = reinterpret_cast<const uint8_t*>(s_px);
    uint8_t* s_sc = reinterpret_cast<uint8_t*>(s_scw);
#if defined(FAST_STOP) && FAST_STOP == 1      // diagnostic builds (tools/fast_phases.sh): where does the kernel's time go
    if (s_px[tid] == 0x12345678u) status[1] = 1;
    return;
#endif
    // ---- phase 1
    // Lane -> (score row sy0 + 7 it, word w) with sy0 = tid / 36, w = tid % 36 (252 lanes, five rounds): the word's column mask is
    // loaded once, no division per item.  A lane collects the verdicts of its <= 20 pixels in one register (bit 8 k + it = pixel k of
    // round it) and reserves its list entries with ONE LDS atomic; the entries are written by a loop over the set bits (about one per
    // lane: ~5 % of the pixels pass) instead of four predicated blocks per item.
    // Round 5 (the kernel is bound by vector issue, so what a round issues besides its 30 packed operations counts): the round's rows are
    // a compile-time offset from ONE lane address (the row clamp is needed in the last round only); the loop exists twice, for tiles whose
    // score rows are all valid (every tile but those of a level's first and last tile row: no row test, two compares and two selects per
    // round less) and for the others - as one loop the compiler turned the uniform test into selects; the byte selectors of the v_perm
    // unpacking live in scalar registers (as literals they were re-materialised into vector registers in every round).
    {
        const int sy0 = min(tid / FPW, 6), w = tid - (tid / FPW) * FPW;      // (lanes 252 .. 255 hold no item: their column mask is 0)
        const uint32_t vmw = tid < 7 * FPW ? s_vm[w] : 0u;
        // rows: the tile's score rows y0-1 .. y0+32 all inside [ylo, yhi] (every tile but those of the level's first and last tile rows):
        // no row test per round; a band of 7 rows with no row inside (edgeThreshold = 31 leaves the first 29 rows of a level unscored)
        // is skipped by the whole workgroup.
        const bool allrows = y0 - 1 >= ylo && y0 + TILE_H <= yhi;
        const bool anycol = xlo <= xhi;
        uint32_t SEL01, SEL23, SEL12, SEL34, SELSG;
        asm volatile("s_mov_b32 %0, 0x0c010c00\n\ts_mov_b32 %1, 0x0c030c02\n\ts_mov_b32 %2, 0x0c020c01\n\ts_mov_b32 %3, 0x0c040c03\n\ts_mov_b32 %4, 0x07050301"
                     : "=s"(SEL01), "=s"(SEL23), "=s"(SEL12), "=s"(SEL34), "=s"(SELSG));
        const uint32_t* const cw0 = s_px + (sy0 + 3) * FPW + w;
        const uint32_t* const cw4 = s_px + (min(sy0 + 28, FSC_H - 1) + 3) * FPW + w;
        const fs2 T = fs2{ (short)thr, (short)thr };
        auto pretest = [&](auto ar) -> uint32_t {
            constexpr bool AR = decltype(ar)::value;
            uint32_t M = 0;
#pragma unroll
            for (int it = 0; it < (FSC_H + 6) / 7; it++) {
                const int yb0 = y0 - 1 + 7 * it, yb1 = y0 - 1 + min(7 * it + 6, FSC_H - 1);      // the band's rows (uniform)
                if (!AR && !(yb1 >= ylo && yb0 <= yhi)) continue;
                const int sy = sy0 + 7 * it;
                uint32_t vm = (7 * it + 6 < FSC_H || sy < FSC_H) ? vmw : 0u;
                if (!AR) { const int y = y0 - 1 + sy; vm = (y >= ylo && y <= yhi) ? vm : 0u; }
                const uint32_t* cw = 7 * it + 6 < FSC_H ? cw0 + 7 * it * FPW : cw4;
                const uint32_t W0 = cw[0], Wm = cw[-1], Wp = cw[1], U = cw[-3 * FPW], D = cw[3 * FPW];
                // packed 16-bit operand pairs: pixels (0,1) and (2,3) of the word; left = pixel - 3, right = pixel + 3
                const fs2 C0 = fpk(0, W0, SEL01), C1 = fpk(0, W0, SEL23);
                const fs2 U0 = fpk(0, U, SEL01), U1 = fpk(0, U, SEL23);
                const fs2 D0 = fpk(0, D, SEL01), D1 = fpk(0, D, SEL23);
                const fs2 L0 = fpk(0, Wm, SEL12), L1 = fpk(W0, Wm, SEL34);
                const fs2 R0 = fpk(Wp, W0, SEL34), R1 = fpk(0, Wp, SEL12);
                const fs2 dk0 = fmx(fmn(D0, U0), fmn(R0, L0)), br0 = fmn(fmx(D0, U0), fmx(R0, L0));
                const fs2 dk1 = fmx(fmn(D1, U1), fmn(R1, L1)), br1 = fmn(fmx(D1, U1), fmx(R1, L1));
                const fs2 m0 = T - fmx(C0 - dk0, br0 - C0);             // negative <=> darker pair below v - thr, or brighter pair above v + thr
                const fs2 m1 = T - fmx(C1 - dk1, br1 - C1);
                const uint32_t P = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, m1), __builtin_bit_cast(uint32_t, m0), SELSG) & vm;
                M |= P >> (7 - it);                                      // vm leaves only the sign bits 7, 15, 23, 31
            }
            return M;
        };
        uint32_t M = 0;
        if (anycol) M = allrows ? pretest(std::true_type{}) : pretest(std::false_type{});
        if (M) {
            const int cnt = __popc(M);
            int pos = lds_add_rtn(&s_n, cnt);
            if (pos + cnt <= FLIST_CAP) {                            // (a list that overflows is dropped as a whole below: s_n > FLIST_CAP)
                const int b0 = 4 * (sy0 * FPW + w);
                do {
                    const int q = __ffs(M) - 1;
                    M &= M - 1;
                    s_list[pos++] = (uint16_t)(b0 + (q & 7) * (4 * 7 * FPW) + (q >> 3));      // byte index into the score tile
                } while (M);
            }
        }
    }
    __syncthreads();
#if defined(FAST_STOP) && FAST_STOP == 2
    if (s_n == 0x12345678) status[1] = 1;
    return;
#endif
    // ---- phase 2
    // More survivors than the list holds (more than half of the tile: noise): the list is dropped and EVERY pixel of the score tile
    // takes the full ring test, filtered by the same validity mask as phase 1 - the pretest is only a necessary condition of that
    // test, so the corners are the same.  Slow (19 rounds instead of 1) and exact.
    const bool ovf = s_n > FLIST_CAP;
    const int n = ovf ? 4 * FITEMS : s_n;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + tid;
        if (j >= n) continue;
        int b;
        if (ovf) {
            b = j;
            const int sy = frow_div(b), xb = b - sy * FROW, y = y0 - 1 + sy;
            if (!(y >= ylo && y <= yhi && ((s_vm[xb >> 2] >> (8 * (xb & 3))) & 0x80u))) continue;
        } else b = s_list[j];
        // The ring as 8 registers (ring[k] | ring[k + 8] << 16): one v_lshl_or per pair, then the 16 differences are 8 packed subtractions.
        // (ds_read_u8_d16 / _d16_hi would fill the halves for free, but with SRAM ECC on - gfx950 - a d16 load clears the other half.)
        const uint8_t* c = px + b + 3 * FROW;
        const uint32_t vc = c[0];
        const fs2 VV = __builtin_bit_cast(fs2, (vc | (vc << 16)) + 0x01000100u);      // + 256 in both halves (see below)
        fs2 R8[8];
        R8[0] = __builtin_bit_cast(fs2, (uint32_t)c[3 * FROW]      | ((uint32_t)c[-3 * FROW] << 16));
        R8[1] = __builtin_bit_cast(fs2, (uint32_t)c[3 * FROW + 1]  | ((uint32_t)c[-3 * FROW - 1] << 16));
        R8[2] = __builtin_bit_cast(fs2, (uint32_t)c[2 * FROW + 2]  | ((uint32_t)c[-2 * FROW - 2] << 16));
        R8[3] = __builtin_bit_cast(fs2, (uint32_t)c[FROW + 3]      | ((uint32_t)c[-FROW - 3] << 16));
        R8[4] = __builtin_bit_cast(fs2, (uint32_t)c[3]             | ((uint32_t)c[-3] << 16));
        R8[5] = __builtin_bit_cast(fs2, (uint32_t)c[-FROW + 3]     | ((uint32_t)c[FROW - 3] << 16));
        R8[6] = __builtin_bit_cast(fs2, (uint32_t)c[-2 * FROW + 2] | ((uint32_t)c[2 * FROW - 2] << 16));
        R8[7] = __builtin_bit_cast(fs2, (uint32_t)c[-3 * FROW + 1] | ((uint32_t)c[3 * FROW - 1] << 16));
        // Packed 16-bit lanes: P[k] = (d[k], d[k+8]) + 256; the ring rotated by j is P[k+j], with P[8+i] = swap(P[i]).
        // A = max over the 16 arcs of min(d over 9 ring pixels), Bn = min over arcs of max(d): the pixel is a corner iff
        // A > thr or -Bn > thr, and its score (cornerScore<16>) is max(thr, A, -Bn) - 1.
        // The biased differences 1..511 are positive f16 DENORMALS as bit patterns, which order like the integers: gfx950's three-input
        // v_pk_minimum3_f16 / v_pk_maximum3_f16 build a 9-arc as min3 of three min3 of three (16 + 8 packed ops instead of the 40 of the
        // two-input doubling tree; f16 denormals are never flushed in this build's float mode).
        fs2 P[8], t3[8], u3[8];
#pragma unroll
        for (int k = 0; k < 8; k++) P[k] = VV - R8[k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const fs2 p1 = k < 7 ? P[k + 1] : fswap(P[0]), p2 = k < 6 ? P[k + 2] : fswap(P[k - 6]);
            t3[k] = fmn3(P[k], p1, p2); u3[k] = fmx3(P[k], p1, p2);
        }
        fs2 a9[8], b9[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const fs2 q1 = k < 5 ? t3[k + 3] : fswap(t3[k - 5]), q2 = k < 2 ? t3[k + 6] : fswap(t3[k - 2]);
            const fs2 r1 = k < 5 ? u3[k + 3] : fswap(u3[k - 5]), r2 = k < 2 ? u3[k + 6] : fswap(u3[k - 2]);
            a9[k] = fmn3(t3[k], q1, q2); b9[k] = fmx3(u3[k], r1, r2);
        }
        const fs2 am = fmx3(fmx3(a9[0], a9[1], a9[2]), fmx3(a9[3], a9[4], a9[5]), fmx(a9[6], a9[7]));
        const fs2 bm2 = fmn3(fmn3(b9[0], b9[1], b9[2]), fmn3(b9[3], b9[4], b9[5]), fmn(b9[6], b9[7]));
        const int A = max((int)am.x, (int)am.y) - 256, Bn = min((int)bm2.x, (int)bm2.y) - 256;
        if (!(A > thr || -Bn > thr)) continue;
        const int a0 = max(thr, A);
        const int b0 = min(Bn, -a0);
        s_sc[b] = (uint8_t)(-b0 - 1);
        const int cp = atomicAdd(&s_c, 1);                        // (one LDS atomic per wave: the compiler aggregates a uniform increment)
        if (cp < FCL_CAP) s_cl[cp] = (uint16_t)b;
    }
    __syncthreads();
#if defined(FAST_STOP) && FAST_STOP == 3
    if (s_scw[tid] == 0x12345678u) status[1] = 1;
    return;
#endif
    // ---- 7x7 sigma-2 Gaussian of the tile from the same LDS pixels (integer kernel, (sum + 32768) >> 16; the descriptor's input),
    // ON THE MATRIX CORES.  A separable convolution is two products with banded Toeplitz matrices, and v_mfma_i32_32x32x32_i8 does a
    // 32 x 32 x 32 block of one in 32 cycles - 95 % of the taps it multiplies are zeros and it still frees the vector ALU, which is
    // what this kernel is bound by (the packed dot4 / dot2 version was 321 of its 1 080 vector instructions per wave).  Wave w owns the
    // tile's 32-column block w:
    //   pass 1 (rows)     T''[r][c] = sum_k (px[r][k] - 128) g[k - c - 5] + 128   for tile rows 0..31 and 32..63 (only 40 exist; the rest
    //                     meets zero taps): A = 16 consecutive LDS bytes per lane (xor 0x80 makes them signed), B = taps from a table;
    //                     two k-blocks of 32 bytes cover the 38 input columns.  T = sum g px = T'' + 32768 fits 16 bits (sum g = 257).
    //   pass 2 (columns)  the accumulator tile IS the next product's A operand when that product sums over its ROW index
    //                     (out' = T' G'): hi = T'' >> 8 (a signed byte as it stands), lo = (T'' & 255) - 128, packed with v_perm;
    //                     out = 256 sum g hi + sum g lo + 257 (128 + 32768) + 32768, two accumulators, exact in int32.
    //   The result tile has the output ROW on the lane and 4 consecutive columns in 4 registers: one dword each, transposed through
    //   LDS (the pixel tile's own space, dead by then) so that the stores are whole 128-B row segments.
    // The k order of an accumulator-turned-operand is permuted (element j of lane half h is row (j & 3) + 8 (j >> 2) + 4 h): the tap
    // table of pass 2 is built in that order on the host.  Register budget: 64 per lane (eight waves per SIMD) - the passes are
    // ordered so that at most two accumulator tiles, the packed operands and one pair of tap fragments are live at a time.
    {
        typedef int fv4 __attribute__((ext_vector_type(4)));
        typedef int fv16 __attribute__((ext_vector_type(16)));
        const int lane = tid & 63, wv = tid >> 6, rn = lane & 31, h = lane >> 5;
        const fv4* gt = reinterpret_cast<const fv4*>(gtab) + lane;
        const uint8_t* pxb = reinterpret_cast<const uint8_t*>(s_px);
        fv4 alo[2], ahi[2];
        {
            const fv4 P1B0 = gt[0], P1B1 = gt[64];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const uint8_t* rp = pxb + min(32 * rb + rn, FPX_H - 1) * FROW + 32 * wv + 16 * h;
                // k slots 48 .. 63 of the second product (lane half 1) only meet zero taps: they carry the constant bytes (1, 1, 0, ..)
                // against taps (64, 64, 0, ..) instead - the "+ 128" without sixteen accumulator moves
                fv4 a0 = *reinterpret_cast<const fv4*>(rp), a1 = *reinterpret_cast<const fv4*>(h ? reinterpret_cast<const uint8_t*>(s_k1) : rp + 32);
                a0 ^= (int)0x80808080u; a1 ^= (int)0x80808080u;
                fv16 acc = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, P1B0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, P1B1, acc, 0, 0, 0);
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t q01 = __builtin_amdgcn_perm((uint32_t)acc[4 * d + 1], (uint32_t)acc[4 * d], 0x05010400u);      // (r0.b0, r1.b0, r0.b1, r1.b1)
                    const uint32_t q23 = __builtin_amdgcn_perm((uint32_t)acc[4 * d + 3], (uint32_t)acc[4 * d + 2], 0x05010400u);
                    alo[rb][d] = (int)(__builtin_amdgcn_perm(q23, q01, 0x05040100u) ^ 0x80808080u);
                    ahi[rb][d] = (int)__builtin_amdgcn_perm(q23, q01, 0x07060302u);
                }
            }
        }
        asm volatile("" ::: "memory");                             // keep the pass-2 tap loads below pass 1 (hoisted, they cost a spilled register quad)
        const fv4 P2B0 = gt[128], P2B1 = gt[192];
        fv16 accH = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        accH = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[0], P2B0, accH, 0, 0, 0);
        accH = __builtin_amdgcn_mfma_i32_32x32x32_i8(ahi[1], P2B1, accH, 0, 0, 0);
        const int CI = 257 * (128 + 32768) + 32768;
        fv16 accL;                                                 // 256 sum g hi + the constant is the C operand of the low product: no constant fill, no combine afterwards
#pragma unroll
        for (int i = 0; i < 16; i++) accL[i] = (accH[i] << 8) + CI;
        accL = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[0], P2B0, accL, 0, 0, 0);
        accL = __builtin_amdgcn_mfma_i32_32x32x32_i8(alo[1], P2B1, accL, 0, 0, 0);
        // lane = output row rn; registers 4 g .. 4 g + 3 = columns 32 w + 8 g + 4 h + (0..3): bits 16 .. 24 of the sums, clamped to 255
        uint32_t* s_o = s_px;                                      // 32 rows x 33 words
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const us2 lim = us2{ 255, 255 };
        uint32_t ob[4];
#pragma unroll
        for (int g2 = 0; g2 < 4; g2++) {
            const us2 p01 = __builtin_elementwise_min(__builtin_bit_cast(us2, __builtin_amdgcn_perm((uint32_t)accL[4 * g2 + 1], (uint32_t)accL[4 * g2], 0x07060302u)), lim);
            const us2 p23 = __builtin_elementwise_min(__builtin_bit_cast(us2, __builtin_amdgcn_perm((uint32_t)accL[4 * g2 + 3], (uint32_t)accL[4 * g2 + 2], 0x07060302u)), lim);
            ob[g2] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p23), __builtin_bit_cast(uint32_t, p01), 0x06040200u);
        }
        // ---- phase 3: kept corners are staged in LDS as (score << 24) | (score-tile byte index), then the tile reserves its
        // range of the level's corner list with ONE global atomic (a returning atomic per wave was measured 35 % slower: every wave
        // then waits out an L2 round trip on a counter all tiles of the level contend for).
        // Round 5: in the common case (the corners fit s_cl) phase 3 needs neither the pixel tile nor the candidate list, so it runs HERE, on
        // the heels of the blur's products, staging into the dead candidate list - and the tail of the kernel is two barriers (every wave is
        // through with the pixels -> the blurred rows are transposed through the pixel tile's space while lane 0 reserves the list range ->
        // row stores and list entries go out together) where the blur and phase 3 one after the other took five.
        const int nc = s_c;
        const bool cl = nc <= FCL_CAP;                              // corners listed: ~60 % of the candidates; otherwise walk the candidates as before
        const int n3 = cl ? nc : n;
        uint32_t* const s_out = cl ? reinterpret_cast<uint32_t*>(s_list) : s_px;      // kept corners: strict 3x3 maxima, at most one per 2x2 block = 1024 entries
        const size_t lv = (size_t)frame * g.nlevels + level;
        uint8_t* bt = blur + fbase + (uint32_t)(y0 * L.pitch + x0);
        const __amdgpu_buffer_rsrc_t rs_bt = __builtin_amdgcn_make_buffer_rsrc(bt, 0, 0x7fffffff, 0x00020000);
        auto blur_rows_out = [&]() {   // whole 128-byte row segments; tile base in scalar registers, one 32-bit offset per lane
            const int wd = tid & 31, r0 = tid >> 5;
            const bool xin = x0 + 4 * wd < L.w;
            const int o0 = (int)(__umul24((uint32_t)r0, (uint32_t)L.pitch) + 4u * (uint32_t)wd);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int row = r0 + 8 * k;
                if (xin && y0 + row < L.h) __builtin_amdgcn_raw_buffer_store_b32(s_o[row * 33 + wd], rs_bt, o0, 8 * k * L.pitch, 0);
            }
        };
        auto nms = [&]() {
            for (int j = tid; j < n3; j += 256) {
                const int b = cl ? s_cl[j] : (ovf ? j : s_list[j]);
                const int sc = s_sc[b];
                if (sc == 0) continue;
                const int sy = frow_div(b), xb = b - sy * FROW;
                if (sy < 1 || sy > TILE_H || xb < 8 || xb >= 8 + TILE_W) continue;            // halo positions belong to the neighbour tiles
                const int x = x0 - 8 + xb, y = y0 - 1 + sy;
                if (!(x >= et && x < L.w - et && y >= et && y < L.h - et)) continue;
                const uint8_t* q = s_sc + b;
                const bool keep = sc > q[-1] && sc > q[1] && sc > q[-FROW - 1] && sc > q[-FROW] && sc > q[-FROW + 1] &&
                                  sc > q[FROW - 1] && sc > q[FROW] && sc > q[FROW + 1];
                if (!keep) continue;
                if (mbits) { if (!((s_mask[(sy - 1) * (TILE_W / 32) + ((xb - 8) >> 5)] >> ((xb - 8) & 31)) & 1u)) continue; }
                else if (mpyr && mpyr[fbase + (size_t)y * L.pitch + x] == 0) continue;
                const int k = atomicAdd(&s_k, 1);
                s_out[k] = ((uint32_t)sc << 24) | (uint32_t)b;
            }
        };
        auto reserve = [&]() {
#if defined(FAST_STOP) && FAST_STOP == 5      // diagnostic: no returning global atomic at the tile's end (wrong lists)
            if (tid == 0 && s_k) s_base = 0;
#else
            if (tid == 0 && s_k) s_base = atomicAdd(&ccnt[lv], s_k);
#endif
        };
        auto list_out = [&]() {
            const int kk = s_k;
            if (kk) {
                const int base = s_base;
                uint2* dst = clist + (size_t)frame * g.list_stride + L.list_off;
                for (int i = tid; i < kk; i += 256) {
                    const uint32_t e = s_out[i];
                    const int b = e & 0xffffffu, sy = frow_div(b), xb = b - sy * FROW;
                    const uint32_t x = (uint32_t)(x0 - 8 + xb), y = (uint32_t)(y0 - 1 + sy);
                    if (base + i < L.list_cap) dst[base + i] = make_uint2((y << 16) | x, e >> 24);
                    else atomicOr(status, BH_ST_CAND_OVERFLOW);
                }
            }
        };
#if defined(FAST_STOP) && FAST_STOP == 4
        __syncthreads();
#pragma unroll
        for (int g2 = 0; g2 < 4; g2++) s_o[rn * 33 + 8 * wv + 2 * g2 + h] = ob[g2];
        __syncthreads();
        blur_rows_out();
        return;
#endif
        if (cl) {
            nms();
            __syncthreads();                                       // every wave has read its pixels; the kept corners are staged
#pragma unroll
            for (int g2 = 0; g2 < 4; g2++) s_o[rn * 33 + 8 * wv + 2 * g2 + h] = ob[g2];
            reserve();
            __syncthreads();
            blur_rows_out();
            list_out();
        } else {                                                   // more corners than s_cl holds: phase 3 walks the candidate list (or the tile) and stages in the pixel tile's space
            __syncthreads();                                       // every wave has read its pixels
#pragma unroll
            for (int g2 = 0; g2 < 4; g2++) s_o[rn * 33 + 8 * wv + 2 * g2 + h] = ob[g2];
            __syncthreads();
            blur_rows_out();
            __syncthreads();                                       // the pixel tile is dead from here on: s_out takes its place
            nms();
            __syncthreads();
            reserve();
            __syncthreads();
            list_out();
        }
    }
}

// ------------------------------------------------------------------------------------------------ select (per frame)
#define SEL_T    512
#define SEL_W    8             // waves
#define CAND_CAP 4096
#define VL_CAP   768           // recorded non-empty score-map vectors per wave

__device__ __forceinline__ int wave_excl_scan(int v, int lane)
{
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(s, d, 64); if (lane >= d) s += o; }
    return s - v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Wave-wide integer sum on the DPP network (row shifts inside the four 16-lane rows, then row_bcast:15 / row_bcast:31 into the
// following rows): no LDS crossbar round trips.  The total arrives in lane 63 and is handed to every lane.
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int count_ge(uint32_t w, uint32_t thr)
{
    return (int)(((w & 255u) >= thr) + (((w >> 8) & 255u) >= thr) + (((w >> 16) & 255u) >= thr) + ((w >> 24) >= thr));
}


__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = __fdiv_rn(ax, ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// One workgroup per (level, frame).  Input: the level's corner list written by k_fast (unordered) and its score histogram.
// Output: the level's kept keypoints in raster order in a staging slot sel_xy/sel_resp[frame][level][sel_cap] + sel_cnt[frame][level];
// k_describe turns slots into final keypoints.
__global__ __launch_bounds__(SEL_T) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_select(const uint8_t* __restrict__ pyr, const uint2* __restrict__ clist,
                                                  const int* __restrict__ ccnt, bh_orb_geom g,
                                                  int sel_cap, uint32_t* __restrict__ sel_xy, float* __restrict__ sel_resp,
                                                  int* __restrict__ sel_cnt, uint32_t* __restrict__ status)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_xy[CAND_CAP];
    __shared__ __attribute__((aligned(16))) float    s_resp[CAND_CAP];
    __shared__ uint32_t s_hs[256];
    __shared__ int s_thr, s_m, s_kept;

    const int tid = threadIdx.x;
    // frames along x (the fast launch index): with levels along x and eight levels, workgroup b lands on XCD b mod 8 = its LEVEL - one XCD had every
    // level-0 list (a third of all candidates, the largest images) and another only the smallest
    const int level = blockIdx.y;
    const size_t f = blockIdx.x;
    const size_t fb = f * g.pyr_stride;
    const bh_level_geom L = g.lv[level];
    const int n2 = 2 * L.nfeat;
    // (a) retainBest(2 n_l) threshold on the u8 FAST score: largest s with #(score >= s) >= 2 n_l
    const uint2* lst = clist + f * (size_t)g.list_stride + L.list_off;
    const int nc = min(ccnt[f * g.nlevels + level], L.list_cap);
    if (tid < 256) s_hs[tid] = 0;
    if (tid == 0) { s_thr = 1; s_m = 0; s_kept = 0; }
    __syncthreads();
    // The workgroup is a chain of dependent phases and the launch lasts as long as its slowest workgroup (a level-0 list): what it waits for is
    // memory round trips in series.  Round 5: the first SEL_LR x 512 list entries are loaded ONCE, all loads in flight together (the histogram pass
    // and the candidate pass each walked the list 512 entries per round trip), and the nine rows of a Harris window are requested before the
    // first is used (the compiler's schedule had five in flight and fetched the other four one behind the other).
    constexpr int SEL_LR = 4;
    uint2 lr[SEL_LR];
#pragma unroll
    for (int k = 0; k < SEL_LR; k++) { const int i = tid + k * SEL_T; lr[k] = i < nc ? lst[i] : make_uint2(0u, 0u); }
#pragma unroll
    for (int k = 0; k < SEL_LR; k++) if (tid + k * SEL_T < nc) atomicAdd(&s_hs[lr[k].y & 255u], 1u);      // score histogram of the level's corners
    for (int i = tid + SEL_LR * SEL_T; i < nc; i += SEL_T) atomicAdd(&s_hs[lst[i].y & 255u], 1u);
    __syncthreads();
    if (tid >= 1 && tid < 256) {
        uint32_t suf = 0;
        for (int b = tid; b < 256; b++) suf += s_hs[b];
        const uint32_t suf1 = suf - s_hs[tid];                 // #(score >= tid+1)
        if (suf >= (uint32_t)n2 && suf1 < (uint32_t)n2) s_thr = tid;
    }
    __syncthreads();
    const uint32_t thr = n2 > 0 ? (uint32_t)s_thr : 256u;
    // (b) the corners at or above the threshold (ties at the cut all kept), in whatever order the list has them
#pragma unroll
    for (int k = 0; k < SEL_LR; k++)
        if (tid + k * SEL_T < nc && lr[k].y >= thr) { const int pos = atomicAdd(&s_m, 1); if (pos < CAND_CAP) s_xy[pos] = lr[k].x; }
    for (int i = tid + SEL_LR * SEL_T; i < nc; i += SEL_T) {
        const uint2 r = lst[i];
        if (r.y >= thr) { const int pos = atomicAdd(&s_m, 1); if (pos < CAND_CAP) s_xy[pos] = r.x; }
    }
    __syncthreads();
    int m = s_m;
    if (m > CAND_CAP) { if (tid == 0) atomicOr(status, BH_ST_CAND_OVERFLOW); m = CAND_CAP; }
    // (c) Harris response, 7x7 block of 3x3 Sobel products
    const uint8_t* img = pyr + fb + L.off;
    for (int i = tid; i < m; i += SEL_T) {
        const uint32_t xy = s_xy[i];
        const int cx = xy & 0xffffu, cy = xy >> 16;
        int a = 0, b = 0, c = 0;
        // The 9 bytes of a window row are ONE aligned 12-byte load (rows are 64-B aligned, so the window's byte offset is cx - 4 mod 4)
        // plus funnel shifts: 81 single-byte loads per corner, every lane on its own cache line, were the longest phase of this kernel.
        // The Sobel products run on 16-bit pairs: with the column sums cs = r0 + 2 r1 + r2 and row differences rd = r2 - r0 as
        // (even, odd) halves, Ix[k] = cs[k+1] - cs[k-1] is one v_pk_sub of neighbouring pairs, Iy[k] = rd[k-1] + 2 rd[k] + rd[k+1]
        // two packed adds, and sum Ix^2, Iy^2, Ix Iy take two pixels per v_dot2_i32_i16 (|Ix|, |Iy| <= 1020; exact integers).
        const uint8_t* p = img + (size_t)(cy - 4) * L.pitch + ((cx - 4) & ~3);
        const uint32_t sh = ((uint32_t)(cx - 4) & 3u) << 3;
        fs2 X0[5], X1[5], X2[5];
        uint32_t RW[9][3];                                      // all nine rows in flight together
#pragma unroll
        for (int row = 0; row < 9; row++) {
            const uint32_t* q_ = reinterpret_cast<const uint32_t*>(p + (size_t)row * L.pitch);
            RW[row][0] = q_[0]; RW[row][1] = q_[1]; RW[row][2] = q_[2];
        }
#pragma unroll
        for (int row = 0; row < 9; row++) asm volatile("" : "+v"(RW[row][0]), "+v"(RW[row][1]), "+v"(RW[row][2]));      // (keeps the loads above the first use)
#define SEL_ROW(X, row) { const uint32_t w0_ = RW[row][0], w1_ = RW[row][1], w2_ = RW[row][2]; \
            const uint32_t lo_ = __builtin_amdgcn_alignbit(w1_, w0_, sh), hi_ = __builtin_amdgcn_alignbit(w2_, w1_, sh); \
            X[0] = fpk(0, lo_, 0x0c010c00u); X[1] = fpk(0, lo_, 0x0c030c02u); X[2] = fpk(0, hi_, 0x0c010c00u); X[3] = fpk(0, hi_, 0x0c030c02u); \
            X[4] = __builtin_bit_cast(fs2, (w2_ >> sh) & 255u); }
        SEL_ROW(X0, 0); SEL_ROW(X1, 1);
#pragma unroll
        for (int row = 2; row < 9; row++) {
            SEL_ROW(X2, row);
            fs2 cs[5], rd[5];
#pragma unroll
            for (int j = 0; j < 5; j++) { cs[j] = X0[j] + X2[j] + X1[j] + X1[j]; rd[j] = X2[j] - X0[j]; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                fs2 ix = cs[j + 1] - cs[j];                          // (Ix[2j+1], Ix[2j+2])
                const fs2 md = __builtin_bit_cast(fs2, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, rd[j + 1]), __builtin_bit_cast(uint32_t, rd[j]), 16u));
                fs2 iy = rd[j] + md + md + rd[j + 1];                // (Iy[2j+1], Iy[2j+2])
                if (j == 3) { ix.y = 0; iy.y = 0; }                  // k = 8 is outside the 7x7 block
                a = __builtin_amdgcn_sdot2(ix, ix, a, false); b = __builtin_amdgcn_sdot2(iy, iy, b, false); c = __builtin_amdgcn_sdot2(ix, iy, c, false);
            }
#pragma unroll
            for (int j = 0; j < 5; j++) { X0[j] = X1[j]; X1[j] = X2[j]; }
        }
#undef SEL_ROW
        const float scale = 1.f / ((1 << 2) * 7 * 255.f);
        const float scale_sq_sq = scale * scale * scale * scale;
        const float fa = (float)a, fbb = (float)b, fc = (float)c;
        const float t1 = fa * fbb, t2 = fc * fc, sm = fa + fbb;
        const float t3 = (0.04f * sm) * sm;
        s_resp[i] = ((t1 - t2) - t3) * scale_sq_sq;
    }
    __syncthreads();
    // (d) retainBest(n_l): keep i iff fewer than n_l responses are strictly greater (ties at the cut all kept).  The O(m^2) scans
    // read the shared arrays four entries per ds_read_b128; the tail up to a multiple of 4 is padded so that it never counts.
    const int m4 = (m + 3) & ~3;
    for (int i = m + tid; i < m4; i += SEL_T) { s_resp[i] = -INFINITY; s_xy[i] = 0xffffffffu; }
    __syncthreads();
    for (int i = tid; i < m; i += SEL_T) {
        bool keep = true;
        if (m > L.nfeat) {
            const float r = s_resp[i];
            int rank = 0;
            for (int j = 0; j < m4; j += 4) {
                const float4 q = *reinterpret_cast<const float4*>(s_resp + j);
                rank += (q.x > r) + (q.y > r) + (q.z > r) + (q.w > r);
            }
            keep = rank < L.nfeat;
        }
        if (!keep) s_xy[i] = 0xffffffffu;                         // dropped entries sort behind every kept one (nobody else reads s_xy here)
        if (keep) atomicAdd(&s_kept, 1);
    }
    __syncthreads();
    // (e) canonical order: a kept keypoint's slot is the number of kept keypoints before it in raster order ((y << 16) | x is the key)
    const size_t slot = (f * g.nlevels + level) * (size_t)sel_cap;
    for (int i = tid; i < m; i += SEL_T) {
        const uint32_t key = s_xy[i];
        if (key == 0xffffffffu) continue;
        int pos = 0;
        for (int j = 0; j < m4; j += 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(s_xy + j);
            pos += (q.x < key) + (q.y < key) + (q.z < key) + (q.w < key);
        }
        if (pos < sel_cap) { sel_xy[slot + pos] = key; sel_resp[slot + pos] = s_resp[i]; }
    }
    if (tid == 0) {
        int kt = s_kept;
        if (kt > sel_cap) { atomicOr(status, BH_ST_KP_OVERFLOW); kt = sel_cap; }
        sel_cnt[f * g.nlevels + level] = kt;
    }
}

// ------------------------------------------------------------------------------------------------ angle + steered BRIEF
// sin/cos in f64 for |x| <= ~8 (the keypoint angle in radians), accurate to < 1 ulp: Cody-Waite reduction by pi/2 (two
// constants: k <= 6 so k*PIO2_HI is exact) and the classic degree-13/14 minimax kernels on [-pi/4, pi/4].  The reference
// needs (float)cos((double)angle) for a float angle in [0, 2 pi]: tools/check_sincos_small.c runs this exact arithmetic
// (IEEE +,-,*, rint; no FMA) over ALL 1 086 918 650 floats in [0, 6.2832] and finds zero differences from glibc's
// sin/cos after the rounding to float — and it costs a quarter of the generic library routine (no Payne-Hanek path).
__device__ __forceinline__ void sincos_small(double x, double* sn, double* cs)
{
    const double INV_PIO2 = 6.36619772367581382433e-01, PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double kd = rint(x * INV_PIO2);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double z = r * r;
    const double ps = z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double s0 = r + r * ps;
    const double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double c0 = (1.0 - 0.5 * z) + z * pc;
    const int q = k & 3;
    const double ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

__constant__ __attribute__((aligned(4))) signed char c_pattern[1024] = {
#include "brief_pattern.h"
};

// One 256-lane workgroup per 64 final keypoints of a frame, 16 per wave (final index = prefix of the per-level counts + slot:
// level-major, raster inside a level).  What a keypoint costs is split by who can share it:
//   phase A  wave per keypoint: intensity-centroid sums over the 749-pixel disc (lanes cover 2 rows x 32 columns per step, integer DPP
//            wave sums) -> m01, m10 into LDS;
//   phase B  LANE per keypoint: fastAtan2, the f64 sin / cos of the angle, the scaled coordinates - the ~160 instructions that are
//            the same in all 64 lanes when a wave owns one keypoint run once for 64 keypoints here; the keypoint record is stored
//            (64 contiguous 24-B structs);
//   phase C  wave per keypoint: steered BRIEF, lane l evaluates tests l, 64+l, 128+l, 192+l (its 16 pattern coordinates are
//            converted to float once per wave and reused for all 16 keypoints); four ballots give the 32 descriptor bytes.
#ifndef DS_KPW
#define DS_KPW 16
#endif
#define DS_PR 37               // BRIEF window: the pattern's largest radius is 18.385, so a rotated, rounded coordinate is within +-18
#define DS_PH 18
#ifndef DS_PW
#define DS_PW 12               // words per staged window row: 37 bytes + up to 3 of alignment = 40 B are needed; 48 B (a multiple of 16) keep every lane QUAD
#endif                         // inside one row - with 10 words a quad straddled two rows in half of the cases, which costs the address path 4 x (tools/vmem_rate.hip)
#define DS_NL ((DS_PR * DS_PW + 63) / 64)   // staging loads per lane (7)
__global__ __launch_bounds__(256) void k_describe(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                  bh_orb_geom g, const uint32_t* __restrict__ ic_mask, float patch_size, int sel_cap, int max_kp, int nframes,
                                                  const uint32_t* __restrict__ sel_xy, const float* __restrict__ sel_resp,
                                                  const int* __restrict__ sel_cnt, bh_kp* __restrict__ kps,
                                                  uint8_t* __restrict__ desc, int* __restrict__ n_kp,
                                                  uint32_t* __restrict__ status)
{
    __shared__ int s_m01[64], s_m10[64];
    __shared__ uint32_t s_loc[64], s_bxy[64];                   // level << 28 | cy << 14 | cx (all ones: no keypoint);  by << 16 | bx
    __shared__ float s_a[64], s_b[64], s_resp[64], s_scale[BH_MAX_LEVELS];
    __shared__ uint32_t s_patch[4][2][DS_PR * DS_PW];            // per wave: the blurred 37 x 37 windows of two keypoints (rows of 10 words)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the level lookups stay scalar
    // Which keypoints a workgroup takes.  Its 64 items are 4 CONSECUTIVE keypoints (one per wave) of each of 16 FRAMES (a wave's 16 steps),
    // and the frames of one workgroup - and of every workgroup whose launch index is equal mod 8 - are f = x + 8 n: workgroups b and b + 8
    // share an XCD and its 4 MiB L2 (observed placement - speed only, any placement is correct).  So at any moment the waves of an XCD work on
    // the same few frames, whose pyramid / blurred levels (0.6 MB each) stay in that L2: a keypoint's window is 32 + 37 rows of one or two
    // 128-B lines, neighbouring keypoints share most of them, and with 16 frames per XCD in flight at once (a wave per 16 keypoints of ONE
    // frame, the layout until round 4) two thirds of the lines came from HBM a second and a third time (tools/vmem_rate.hip: a line from HBM
    // costs a CU 13-16 cycles, from L2 3.4).
    const int lin = blockIdx.x, xcd = lin & 7, nb4 = (max_kp + 3) >> 2, jb = lin >> 3, fc = jb / nb4, kq = (jb - fc * nb4) * 4;
    const int fbase = xcd + 8 * DS_KPW * fc;                     // item i of a wave: frame fbase + 8 i, keypoint kq + wv
    if (tid < g.nlevels) s_scale[tid] = g.lv[tid].scale;
    // ---- phase A: intensity-centroid sums.  Lane = (row parity, column u): step `it` visits row v = 2 it + (lane >> 5) - 15.  Which steps
    // fall inside the disc is a per-lane constant (ic_mask, built on the host from the reference's umax table), and with S = sum I,
    // T = sum it*I over a lane's in-disc pixels: m10 = u S, m01 = 2 T + ((lane >> 5) - 15) S (integers, exact).
    const int u = (lane & 31) - 15, vb = (lane >> 5) - 15;
    const uint32_t msk = ic_mask[lane];
    uint32_t kc[16];                                             // per step: in the disc ? 1 | it << 16 : 0
#pragma unroll
    for (int it = 0; it < 16; it++) { kc[it] = (msk >> it) & 1u ? (1u | ((uint32_t)it << 16)) : 0u; asm volatile("" : "+v"(kc[it])); }      // (opaque: the compiler would split the multiply into and + shifts again)
    // lane i < 16 looks up keypoint i of this wave (level, position, response) - one round trip for all sixteen; the patch loads of FOUR
    // keypoints are in flight together (a wave that walks its keypoints one dependent load chain at a time leaves the memory system
    // idle: the kernel was 4 % slower than one wave per keypoint with 40 % fewer instructions)
    uint32_t my_loc = 0xffffffffu; float my_resp = 0.f;
    if (lane < DS_KPW && fbase + 8 * lane < nframes) {
        const size_t f = fbase + 8 * lane;
        const int* cnt = sel_cnt + f * g.nlevels;
        const int k = kq + wv;
        int level = -1, slot = 0, t2 = 0;
        for (int l = 0; l < g.nlevels; l++) { const int c = cnt[l]; if (k >= t2 && k < t2 + c) { level = l; slot = k - t2; } t2 += c; }
        if (kq == 0 && wv == 0) { n_kp[f] = min(t2, max_kp); if (t2 > max_kp) atomicOr(status, BH_ST_KP_OVERFLOW); }
        if (level >= 0 && k < max_kp) {
            const size_t si = (f * g.nlevels + level) * (size_t)sel_cap + slot;
            const uint32_t xy = sel_xy[si];
            my_loc = ((uint32_t)level << 28) | ((xy >> 16) << 14) | (xy & 0xffffu);
            my_resp = sel_resp[si];
        }
    }
    if (lane < DS_KPW) { s_loc[wv * DS_KPW + lane] = my_loc; s_resp[wv * DS_KPW + lane] = my_resp; }
    if (!__syncthreads_or(my_loc != 0xffffffffu)) return;       // no keypoint 4 q .. 4 q + 3 in any of the 16 frames
#pragma unroll
    for (int i0 = 0; i0 < DS_KPW; i0 += 4) {
        int I[4][16];
        uint32_t loc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            loc[q] = (uint32_t)__builtin_amdgcn_readlane((int)my_loc, i0 + q);
            if (loc[q] == 0xffffffffu) continue;                 // wave-uniform
            const bh_level_geom L = g.lv[loc[q] >> 28];
            const uint8_t* __restrict__ lvl = pyr + (size_t)(fbase + 8 * (i0 + q)) * g.pyr_stride + L.off;           // wave-uniform base, 32-bit lane offsets
            const uint32_t o = __umul24((uint32_t)((int)((loc[q] >> 14) & 0x3fffu) + vb), (uint32_t)L.pitch) + (uint32_t)((int)(loc[q] & 0x3fffu) + u);      // (24-bit multiply: a full 32-bit one runs at a quarter of the rate)
            // buffer loads: the lane offset stays one VGPR, the row step rides in the scalar offset (16 address adds per keypoint saved)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(lvl), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int it = 0; it < 16; it++) I[q][it] = __builtin_amdgcn_raw_buffer_load_b8(rs, (int)o, it * 2 * L.pitch, 0);      // unconditional (v <= 16, |u| <= 16: inside the level's 31-px border)
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (loc[q] == 0xffffffffu) continue;
            uint32_t X = 0;                                      // S (<= 16 x 255) | T (<= 120 x 255) << 16: one 24-bit multiply-add per pixel
#pragma unroll
            for (int it = 0; it < 16; it++) X = __umul24((uint32_t)I[q][it], kc[it]) + X;
            const int S = (int)(X & 0xffffu), T = (int)(X >> 16);
            const int m10 = wave_sum_dpp(u * S), m01 = wave_sum_dpp(2 * T + vb * S);
            if (lane == 0) { s_m10[wv * DS_KPW + i0 + q] = m10; s_m01[wv * DS_KPW + i0 + q] = m01; }
        }
    }
    __syncthreads();
    // ---- phase B: one lane per keypoint
    if (tid < 4 * DS_KPW && s_loc[tid] != 0xffffffffu) {
        const uint32_t loc = s_loc[tid];
        const int level = loc >> 28, cx = loc & 0x3fffu, cy = (loc >> 14) & 0x3fffu;
        const float scale = s_scale[level];
        const float angle_deg = fast_atan2_deg((float)s_m01[tid], (float)s_m10[tid]);
        bh_kp kp;
        kp.x = (float)cx * scale; kp.y = (float)cy * scale;
        kp.size = patch_size * scale; kp.angle = angle_deg; kp.response = s_resp[tid]; kp.octave = level;
        kps[(size_t)(fbase + 8 * (tid % DS_KPW)) * max_kp + kq + tid / DS_KPW] = kp;
        // the BRIEF centre is recomputed from the scaled float coordinates, as the reference does
        const float inv = 1.f / scale;
        const int bx = __float2int_rn(kp.x * inv), by = __float2int_rn(kp.y * inv);
        float angle = angle_deg;
        angle *= (float)(3.1415926535897932384626433832795 / 180.f);
        double sd, cd;
        sincos_small((double)angle, &sd, &cd);
        s_a[tid] = (float)cd; s_b[tid] = (float)sd; s_bxy[tid] = ((uint32_t)by << 16) | (uint32_t)bx;
    }
    __syncthreads();
    // ---- phase C: steered BRIEF on the blurred level
    float qx0[4], qy0[4], qx1[4], qy1[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int q4 = *reinterpret_cast<const int*>(c_pattern + (t * 64 + lane) * 4);       // the four signed bytes of test 64 t + lane
        qx0[t] = (float)(signed char)q4; qy0[t] = (float)(signed char)(q4 >> 8); qx1[t] = (float)(signed char)(q4 >> 16); qy1[t] = (float)(q4 >> 24);
    }
    // which (row, word) of the window a lane stages in load m does not depend on the keypoint: row | byte offset << 6 | LDS word << 12
    uint32_t stg[DS_NL];
#pragma unroll
    for (int m = 0; m < DS_NL; m++) {
        const int idx = min(lane + 64 * m, DS_PR * DS_PW - 1), row = idx / DS_PW, dw = idx - row * DS_PW;
        stg[m] = (uint32_t)row | ((uint32_t)(4 * dw) << 6) | ((uint32_t)(row * DS_PW + dw) << 12);
    }
    // The 512 samples of a keypoint are 8 loads of 64 scattered bytes: one cache line per lane, which is what bounded this kernel
    // (the address path, not the ALU: 40 % fewer instructions had changed nothing).  The 37 x 37 window they all fall into is staged
    // in LDS with 6 coalesced word loads per wave instead, and the samples are ds_read_u8.
#pragma unroll
    for (int i0 = 0; i0 < DS_KPW; i0 += 2) {                     // two keypoints' windows in flight together
        bool have[2];
        float pa[2], pb[2];
        int sh[2];
        uint32_t wq[2][DS_NL];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int j = wv * DS_KPW + i0 + q;
            const uint32_t loc = s_loc[j], bxy = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_bxy[j]);
            have[q] = loc != 0xffffffffu;                       // wave-uniform
            if (!have[q]) continue;
            const int level = __builtin_amdgcn_readfirstlane((int)(loc >> 28));
            const bh_level_geom L = g.lv[level];
            pa[q] = s_a[j]; pb[q] = s_b[j];
            const int bx = (int)(bxy & 0xffffu) - DS_PH, by = (int)(bxy >> 16) - DS_PH;
            sh[q] = bx & 3;
            // buffer loads from a wave-uniform window origin: the lane's (row x pitch + byte offset) is ONE 24-bit multiply-add per load - as pointer
            // arithmetic each of the 7 x 16 loads cost a 64-bit multiply-add at a quarter of the rate (a fifth of this kernel's vector cycles)
            const uint8_t* org = blur + (size_t)(fbase + 8 * (i0 + q)) * g.pyr_stride + L.off + (size_t)by * L.pitch + (bx & ~3);
            const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(org), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int m = 0; m < DS_NL; m++)
                wq[q][m] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsb, (int)(__umul24(stg[m] & 63u, (uint32_t)L.pitch) + ((stg[m] >> 6) & 63u)), 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (!have[q]) continue;
#pragma unroll
            for (int m = 0; m < DS_NL; m++) s_patch[wv][q][stg[m] >> 12] = wq[q][m];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (!have[q]) continue;
            const uint8_t* center = reinterpret_cast<const uint8_t*>(s_patch[wv][q]) + DS_PH * (4 * DS_PW) + DS_PH + sh[q];
            const float a = pa[q], b = pb[q];
            unsigned long long bits[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float x0 = qx0[t] * a - qy0[t] * b, y0 = qx0[t] * b + qy0[t] * a;
                const float x1 = qx1[t] * a - qy1[t] * b, y1 = qx1[t] * b + qy1[t] * a;
                // round-half-even of each coordinate as the reference's cvRound; row * 40 + column is exact in f32 (|.| < 800), one conversion
                const int t0 = center[(int)__fmaf_rn(rintf(y0), (float)(4 * DS_PW), rintf(x0))];
                const int t1 = center[(int)__fmaf_rn(rintf(y1), (float)(4 * DS_PW), rintf(x1))];
                bits[t] = __ballot(t0 < t1);
            }
            if (lane < 4) {
                unsigned long long v = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
                reinterpret_cast<unsigned long long*>(desc + ((size_t)(fbase + 8 * (i0 + q)) * max_kp + kq + wv) * 32)[lane] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------------ kp -> int coords + 3-D
__global__ __launch_bounds__(256) void k_kp_backproject(const bh_kp* __restrict__ kps, const int* __restrict__ n_kp,
                                                        const uint16_t* __restrict__ depth, int H, int W, int max_kp,
                                                        double fx, double fy, double ppx, double ppy,
                                                        long long* __restrict__ kp_xy, double* __restrict__ cloud)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const size_t f = blockIdx.y;
    if (k >= n_kp[f]) return;
    const bh_kp kp = kps[f * max_kp + k];
    const long long u = (long long)kp.x, v = (long long)kp.y;            // int() truncation, camera.py:142
    kp_xy[(f * max_kp + k) * 2] = u; kp_xy[(f * max_kp + k) * 2 + 1] = v;
    if (cloud && depth) {
        const double z = (double)depth[f * H * W + (size_t)v * W + u] / 1000;
        const double xn = ((double)u - ppx) / fx, yn = ((double)v - ppy) / fy;
        double* o = cloud + (f * max_kp + k) * 3;
        o[0] = xn * z; o[1] = yn * z; o[2] = z;
    }
}

// ------------------------------------------------------------------------------------------------ host side
static void umax_table(int* umax /*[17]*/)
{
    const int hp = 15;
    int v, v0, vmax = (int)floor(hp * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceil(hp * sqrtf(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = round_even_d(sqrt((double)hp * hp - v * v));
    for (v = hp, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

static void gauss7_coeffs(int* k)
{
    float cf[7]; double sum = 0, sigma = 2.0, s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(s2 * x * x); sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); k[i] = round_even_f(cf[i] * 256.f); }
}

// Tables of k_pyr_mfma for every level (see the kernel).  False: some window does not fit the kernel's tile (then k_pyr_down_bits runs).
static bool build_pm_tables(const bh_orb_geom& g, const uint32_t* tab, std::vector<uint32_t>& out, bh_pm_level* pm)
{
    out.clear();
    auto align4 = [&]() { while (out.size() & 3) out.push_back(0); };
    auto put_taps = [&](const int* w /*[64 lanes][16]*/) {          // one fragment pair (c >> 1, c - (c >> 1)), negated, [half][lane][16 bytes]
        const size_t base = out.size(); out.resize(base + 2 * 64 * 4, 0);
        int8_t* b = reinterpret_cast<int8_t*>(out.data() + base);
        for (int i = 0; i < 64 * 16; i++) { const int c = w[i], s0 = c >> 1, s1 = c - s0; b[i] = (int8_t)-s0; b[64 * 16 + i] = (int8_t)-s1; }
    };
    for (int l = 1; l < g.nlevels; l++) {
        const bh_level_geom& D = g.lv[l];
        const uint32_t* tx = tab + D.tabx_off; const uint32_t* ty = tab + D.taby_off;
        bh_pm_level& P = pm[l];
        P.tiles_x = (D.w + PM_TW - 1) / PM_TW; P.tiles_y = (D.h + PM_TH - 1) / PM_TH;
        const int ncb = P.tiles_x * 4;
        // ---- columns: window start + tap fragments per 32-column block
        std::vector<int> wsv(ncb, 0);
        align4(); P.xinfo = (int)out.size(); out.resize(out.size() + ncb, 0);
        align4(); P.xtap = (int)out.size();
        for (int cb = 0; cb < ncb; cb++) {
            const int c0 = 32 * cb;
            int W[32][64]; memset(W, 0, sizeof(W));
            if (c0 < D.w) {
                wsv[cb] = (int)(tx[c0] >> 16) & ~15;
                for (int c = 0; c < 32 && c0 + c < D.w; c++) {
                    const int ofs = (int)(tx[c0 + c] >> 16) - wsv[cb], c1 = (int)(tx[c0 + c] & 0xffffu);
                    if (ofs < 0 || ofs + (c1 ? 1 : 0) > 63 || c1 > 256) return false;
                    W[c][ofs] += 256 - c1; if (c1) W[c][ofs + 1] += c1;
                }
            }
            out[P.xinfo + cb] = (uint32_t)wsv[cb];
            for (int kb = 0; kb < 2; kb++) {
                int frag[64 * 16];
                for (int lane = 0; lane < 64; lane++) for (int j = 0; j < 16; j++) frag[lane * 16 + j] = W[lane & 31][32 * kb + 16 * (lane >> 5) + j];
                put_taps(frag);
            }
        }
        // ---- rows: first source row + tap fragments (k in accumulator order) per 24-row block
        align4(); P.yinfo = (int)out.size(); out.resize(out.size() + P.tiles_y, 0);
        align4(); P.ytap = (int)out.size();
        for (int rb = 0; rb < P.tiles_y; rb++) {
            const int yb = PM_TH * rb, r0 = (int)(ty[yb] >> 16);
            int W[32][32]; memset(W, 0, sizeof(W));
            for (int r = 0; r < PM_TH && yb + r < D.h; r++) {
                const int ofs = (int)(ty[yb + r] >> 16) - r0, c1 = (int)(ty[yb + r] & 0xffffu);
                if (ofs < 0 || ofs + (c1 ? 1 : 0) > 31 || c1 > 256) return false;
                W[r][ofs] += 256 - c1; if (c1) W[r][ofs + 1] += c1;
            }
            out[P.yinfo + rb] = (uint32_t)r0;
            int frag[64 * 16];
            for (int lane = 0; lane < 64; lane++) for (int j = 0; j < 16; j++) frag[lane * 16 + j] = W[lane & 31][(j & 3) + 8 * (j >> 2) + 4 * (lane >> 5)];
            put_taps(frag);
        }
        // ---- mask: per 32-pixel output word the two half windows' first source column and the run masks (source - destination distance 0 .. 5)
        align4(); P.mtab = (int)out.size();
        for (int wc = 0; wc < ncb; wc++) {
            uint32_t e[16]; memset(e, 0, sizeof(e));
            for (int hf = 0; hf < 2; hf++) {
                const int X = 32 * wc + 16 * hf;
                if (X >= D.w) { if (PM_TW * (wc / 4) < D.w) e[hf] = (tx[PM_TW * (wc / 4)] >> 16) & ~31u; continue; }      // empty half: window at the tile's first staged word, no run
                const int s = (int)(tx[X] >> 16);
                e[hf] = (uint32_t)s;
                for (int i = 0; i < 16 && X + i < D.w; i++) {
                    const int d = (int)(tx[X + i] >> 16) - s, k = d - i;
                    if (k < 0 || k > 5 || d + 1 > 31) return false;
                    e[hf ? 8 + k : 2 + k] |= 1u << i;
                }
            }
            out.insert(out.end(), e, e + 16);
        }
        // first source word of a tile's mask window (8 words per row are staged); every window of the tile must lie inside it
        P.mwb = (int)out.size();
        for (int t = 0; t < P.tiles_x; t++) {
            const int xa = PM_TW * t, xb = (xa + PM_TW < D.w ? xa + PM_TW : D.w) - 1;
            const int wb0 = (int)(tx[xa] >> 16) >> 5;
            if ((((int)(tx[xb] >> 16) + 1) >> 5) - wb0 + 1 > 7) return false;      // + the funnel shift's second word
            out.push_back((uint32_t)wb0);
        }
        // ---- mask: special positions (some tap weighs <= 128 of 65 536): truth table of the reference rule T >= 65 408 over the four taps
        const int mty_n = (D.h + PM_MH - 1) / PM_MH;
        for (int b = 0; b < mty_n; b++) {                          // the rows a mask block reads must fit its staging area
            const int ya = PM_MH * b, yb = (ya + PM_MH < D.h ? ya + PM_MH : D.h) - 1;
            if ((int)(ty[yb] >> 16) + 1 - (int)(ty[ya] >> 16) > PM_MSR - 1) return false;
        }
        std::vector<std::vector<uint32_t>> lists((size_t)P.tiles_x * mty_n);
        for (int y = 0; y < D.h; y++) {
            const int cy = (int)(ty[y] & 0xffffu);
            for (int x = 0; x < D.w; x++) {
                const int cx = (int)(tx[x] & 0xffffu);
                const int w4[4] = { (256 - cx) * (256 - cy), cx * (256 - cy), (256 - cx) * cy, cx * cy };      // (ra, ox), (ra, ox+1), (rb, ox), (rb, ox+1)
                uint32_t tt = 0;
                for (int pat = 0; pat < 16; pat++) {
                    int sum = 0;
                    for (int b = 0; b < 4; b++) if (pat >> b & 1) sum += w4[b];
                    if (sum >= 65408) tt |= 1u << pat;
                }
                if (tt != 0x8000u) {
                    std::vector<uint32_t>& v = lists[(size_t)(y / PM_MH) * P.tiles_x + x / PM_TW];
                    v.push_back((uint32_t)(x % PM_TW) | (uint32_t)(y % PM_MH) << 7 | tt << 13);
                    v.push_back((tx[x] >> 16) | (ty[y] >> 16) << 16);
                }
            }
        }
        P.fidx = (int)out.size();
        { const size_t nt = lists.size(); out.resize(out.size() + nt + 1, 0); uint32_t acc = 0; for (size_t t = 0; t < nt; t++) { out[P.fidx + t] = acc; acc += (uint32_t)(lists[t].size() / 2); } out[P.fidx + nt] = acc; }
        while (out.size() & 1) out.push_back(0);
        P.fix = (int)out.size();
        for (auto& v : lists) out.insert(out.end(), v.begin(), v.end());
    }
    return true;
}

static int ensure_geom(bh_ctx* ctx, const bh_orb_params* p, int H, int W)
{
    if (ctx->geom_valid && memcmp(&ctx->geom_params, p, sizeof(*p)) == 0 && ctx->geom.H == H && ctx->geom.W == W)
        return BH_OK;
    bh_orb_geom g;
    if (build_geom(p, H, W, &g) != BH_OK) return bh_set_err(ctx, BH_ERR_INVALID, "unsupported ORB parameters / image size");
    const int gt_off = align_up(g.tab_len + 4 + g.ntiles + 64, 4), tab_words = gt_off + 4 * 64 * 4;
    uint32_t* tab = (uint32_t*)calloc(tab_words, sizeof(uint32_t));
    {   // k_describe's per-lane disc masks (after the tile table): bit `it` set iff pixel (u, v) = ((lane & 31) - 15, 2 it + (lane >> 5) - 15)
        // lies in the intensity-centroid disc of the reference (|v| <= 15, |u| <= umax[|v|])
        int um[17]; umax_table(um);
        for (int lane = 0; lane < 64; lane++) {
            const int u = (lane & 31) - 15;
            uint32_t m = 0;
            for (int it = 0; it < 16; it++) {
                const int v = it * 2 + (lane >> 5) - 15, av = v < 0 ? -v : v;
                if (av <= 15 && u >= -um[av] && u <= um[av]) m |= 1u << it;
            }
            tab[g.tab_len + 4 + g.ntiles + lane] = m;
        }
    }
    {   // k_fast's blur on the matrix cores: int8 tap fragments, [pass 1 k-block 0, 1][pass 2 k-block 0, 1][lane][16 bytes]
        int gk[7]; gauss7_coeffs(gk);
        int8_t* gt = reinterpret_cast<int8_t*>(tab + gt_off);
        for (int kb = 0; kb < 2; kb++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 16; j++) {
                    const int n = lane & 31, hh = lane >> 5;
                    const int t1 = 32 * kb + 16 * hh + j - n - 5;                                    // pass 1: k = byte offset in the LDS row, n = output column
                    const int t2 = 32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hh - n - 1;                // pass 2: k = tile row (accumulator order), n = output row
                    gt[((0 + kb) * 64 + lane) * 16 + j] = (int8_t)((t1 >= 0 && t1 <= 6) ? gk[t1] : ((kb == 1 && hh == 1 && j < 2) ? 64 : 0));      // + the bias slots (k_fast, pass 1)
                    gt[((2 + kb) * 64 + lane) * 16 + j] = (int8_t)((t2 >= 0 && t2 <= 6) ? gk[t2] : 0);
                }
    }
    for (int l = 0; l < g.nlevels; l++)                          // tile table, after the coefficient table (offset tab_len + 4)
        for (int t = 0; t < g.lv[l].tiles_x * g.lv[l].tiles_y; t++)
            tab[g.tab_len + 4 + g.lv[l].tile_base + t] = (uint32_t)l | ((uint32_t)(t % g.lv[l].tiles_x) << 4) | ((uint32_t)(t / g.lv[l].tiles_x) << 16);
    for (int l = 1; l < g.nlevels; l++) {
        lin_coeffs(g.lv[l - 1].w, g.lv[l].w, tab + g.lv[l].tabx_off);
        for (int x = g.lv[l].w; x < align_up(g.lv[l].w, 4); x++) tab[g.lv[l].tabx_off + x] = tab[g.lv[l].tabx_off + g.lv[l].w - 1];
        lin_coeffs(g.lv[l - 1].h, g.lv[l].h, tab + g.lv[l].taby_off);
    }
    // k_gray_l1: level-1 rows per strip = the most (in whole lane items) whose level-0 rows fit the staged GL_R0, checked against the row table
    // itself (the resize ratio of a small level differs from scaleFactor by its rounding)
    ctx->gl_sr = 0;
    if (g.nlevels >= 2) {
        const uint32_t* ty = tab + g.lv[1].taby_off; const int Dh = g.lv[1].h, Sh = g.lv[0].h;
        for (int sr = (int)((GL_R0 - 2) / p->scaleFactor) / GL_PR * GL_PR; sr >= GL_PR && !ctx->gl_sr; sr -= GL_PR) {
            int worst = 0;
            for (int a = 0; a < Dh; a += sr) {
                const int b = std::min(Dh, a + sr), ra = (int)(ty[a] >> 16);
                const int rb = b == Dh ? Sh - 1 : std::min((int)(ty[b - 1] >> 16) + 1, Sh - 1);
                worst = std::max(worst, rb - ra + 1);
            }
            if (worst <= GL_R0) ctx->gl_sr = sr;
        }
    }
    // k_pyr_pair: the same for levels l, l + 1 from level l - 1; besides the staged rows the strips must cover every row of level l
    // (they do for any scale factor below 2: a strip ends one row below its last source row, the next starts at most there)
    for (int l = 0; l < BH_MAX_LEVELS; l++) ctx->pp_sr[l] = 0;
    for (int l = 1; l + 1 < g.nlevels; l++) {
        const uint32_t* ty = tab + g.lv[l + 1].taby_off; const int Dh = g.lv[l + 1].h, Sh = g.lv[l].h;
        const size_t lds = (size_t)GL_R0 * (g.lv[l].pitch + 16) + (size_t)GL_R0 * (g.lv[l].mpitch + 1) * 4;
        if (lds > 64 * 1024) continue;
        for (int sr = (int)((GL_R0 - 2) / p->scaleFactor) / GL_PR * GL_PR; sr >= GL_PR && !ctx->pp_sr[l]; sr -= GL_PR) {
            int worst = 0, next = 0; bool covered = true;
            for (int a = 0; a < Dh; a += sr) {
                const int b = std::min(Dh, a + sr), ra = (int)(ty[a] >> 16);
                const int rb = b == Dh ? Sh - 1 : std::min((int)(ty[b - 1] >> 16) + 1, Sh - 1);
                worst = std::max(worst, rb - ra + 1);
                covered = covered && ra <= next;
                next = rb + 1;
            }
            if (worst <= GL_R0 && covered && next == Sh) ctx->pp_sr[l] = sr;
        }
    }
    int rc = bh_reserve(ctx, &ctx->tab, sizeof(uint32_t) * tab_words);
    if (rc == BH_OK) {
        hipError_t e = hipMemcpyAsync(ctx->tab.p, tab, sizeof(uint32_t) * tab_words, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = bh_set_err(ctx, BH_ERR_HIP, "coefficient table upload failed: %s", hipGetErrorString(e));
    }
    if (rc == BH_OK) {
        std::vector<uint32_t> rt;
        // k_pyr_mfma is bit-identical and issues 40 % fewer vector instructions than k_pyr_down_bits, but its strided 16-byte row loads and
        // tap fragments keep the texture address path busier than the ALUs were: 0.98 - 1.01 against 0.92 ms per 1000 frames,
        // so it runs on request only
        ctx->pm_ok = getenv("BH_ORB_PYR_MFMA") && build_pm_tables(g, tab, rt, ctx->pm) ? 1 : 0;
        if (ctx->pm_ok && !rt.empty()) {
            rc = bh_reserve(ctx, &ctx->rtab, sizeof(uint32_t) * rt.size());
            if (rc == BH_OK) {
                hipError_t e = hipMemcpyAsync(ctx->rtab.p, rt.data(), sizeof(uint32_t) * rt.size(), hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) rc = bh_set_err(ctx, BH_ERR_HIP, "resize table upload failed: %s", hipGetErrorString(e));
            }
        }
    }
    free(tab);
    if (rc != BH_OK) return rc;
    ctx->geom = g; ctx->geom_params = *p; ctx->geom_valid = 1; ctx->ws_frames = 0;
    return BH_OK;
}

extern "C" int bh_orb_geometry(int H, int W, const bh_orb_params* params, int* w, int* h, int* pitch, int* off,
                               int* nfeat, int* total_bytes)
{
    bh_orb_geom g;
    if (build_geom(params, H, W, &g) != BH_OK) return BH_ERR_INVALID;
    for (int l = 0; l < g.nlevels; l++) {
        if (w) w[l] = g.lv[l].w;
        if (h) h[l] = g.lv[l].h;
        if (pitch) pitch[l] = g.lv[l].pitch;
        if (off) off[l] = g.lv[l].off;
        if (nfeat) nfeat[l] = g.lv[l].nfeat;
    }
    if (total_bytes) *total_bytes = g.pyr_stride;
    return BH_OK;
}

#define ORB_CHUNK 256            // host-mode chunk (single pipeline)
#define ORB_CHUNK_DEV 128        // device-mode chunk per pipeline

extern "C" int bh_orb_detect_describe(bh_ctx* ctx, const uint8_t* img, int channels, const uint8_t* mask,
                                      const uint16_t* depth, int B, int H, int W, const bh_orb_params* params,
                                      int max_kp, bh_kp* kps, uint8_t* desc, int* n_kp, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!img || !params || !kps || !desc || !n_kp || B < 0 || max_kp < 1 || (channels != 1 && channels != 3) ||
        (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_orb_detect_describe: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    BH_TRY(ensure_geom(ctx, params, H, W));
    if (B == 0) return BH_OK;
    const bh_orb_geom& g = ctx->geom;
    hipStream_t st = ctx->stream;
    int chunk_cfg = mem == BH_MEM_DEVICE ? ORB_CHUNK_DEV : ORB_CHUNK;
    if (const char* e = getenv("BH_ORB_CHUNK")) { const int v = atoi(e); if (v >= 1 && v <= 4096) chunk_cfg = v; }
    const int chunk = B < chunk_cfg ? B : chunk_cfg;
    const bool has_mask = mask != nullptr || depth != nullptr;

    // Device-resident batches run as two independent chunk pipelines on two streams (each with its own workspace), so
    // the latency-bound kernels of one chunk (k_select, k_describe) overlap the VALU-bound ones of the next.
    const int nchunks = (B + chunk - 1) / chunk;
    // two pipelines by default: measured at 1 000 frames x 640 x 480 (tools/chunk_sweep.sh, round 4) 1 / 2 / 3 / 4 pipelines of 128-frame chunks
    // give 267 / 291 / 289 / 285 k frames/s - a second stream fills the launch gaps and the tails of the latency-bound kernels, more of them
    // only compete for the L2 (250-frame chunks: 274 / 286 / 281)
    int pipe_cfg = 2;
    if (const char* e = getenv("BH_ORB_PIPELINES")) { const int v = atoi(e); if (v >= 1 && v <= BH_MAX_PIPES) pipe_cfg = v; }
    const int npipe = mem == BH_MEM_DEVICE ? (nchunks < pipe_cfg ? nchunks : pipe_cfg) : 1;
    const size_t ws = (size_t)g.pyr_stride * chunk;
    int max_nf = 0;
    for (int l = 0; l < g.nlevels; l++) max_nf = g.lv[l].nfeat > max_nf ? g.lv[l].nfeat : max_nf;
    const int sel_cap = (max_nf + 64 + 7) & ~7;                 // per-level staging slots: n_l + room for ties at the cut
    for (int p = 0; p < npipe; p++) {
        bh_orb_ws& Wk = ctx->ows[p];
        BH_TRY(bh_reserve(ctx, &Wk.pyr, ws + 256));             // +256: the 12-byte resize windows may read past the last row
        if (mask) BH_TRY(bh_reserve(ctx, &Wk.mpyr, ws + 256));
        else if (depth) BH_TRY(bh_reserve(ctx, &Wk.mpyr, sizeof(uint32_t) * (size_t)g.mstride * chunk));      // bit-packed: depth masks are binary
        BH_TRY(bh_reserve(ctx, &Wk.clist, sizeof(uint2) * (size_t)g.list_stride * chunk));
        BH_TRY(bh_reserve(ctx, &Wk.blur, ws));
        BH_TRY(bh_reserve(ctx, &Wk.hist, sizeof(uint32_t) * 257 * g.nlevels * chunk));      // histograms, then the corner counts
        BH_TRY(bh_reserve(ctx, &Wk.sel_xy, sizeof(uint32_t) * (size_t)sel_cap * g.nlevels * chunk));
        BH_TRY(bh_reserve(ctx, &Wk.sel_resp, sizeof(float) * (size_t)sel_cap * g.nlevels * chunk));
        BH_TRY(bh_reserve(ctx, &Wk.sel_cnt, sizeof(int) * (size_t)g.nlevels * chunk));
    }
    BH_TRY(bh_status_ready(ctx));
    ctx->ws_frames = chunk;

    const size_t fpx = (size_t)H * W;
    if (mem == BH_MEM_HOST) {
        BH_TRY(bh_reserve(ctx, &ctx->st_img, fpx * channels * chunk));
        if (mask) BH_TRY(bh_reserve(ctx, &ctx->st_mask, fpx * chunk));
        else if (depth) BH_TRY(bh_reserve(ctx, &ctx->st_depth, fpx * 2 * chunk));
        BH_TRY(bh_reserve(ctx, &ctx->st_kps, sizeof(bh_kp) * (size_t)max_kp * chunk));
        BH_TRY(bh_reserve(ctx, &ctx->st_desc, (size_t)32 * max_kp * chunk));
        BH_TRY(bh_reserve(ctx, &ctx->st_nkp, sizeof(int) * chunk));
    }
    if (npipe > 1) {                                             // fork: the aux streams start after everything queued so far
        BH_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        for (int p = 1; p < npipe; p++) BH_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream[p - 1], ctx->ev_fork, 0));
    }


    for (int f0 = 0, ci = 0; f0 < B; f0 += chunk, ci++) {
        const int nb = (B - f0) < chunk ? (B - f0) : chunk;
        const int pipe = ci % npipe;
        bh_orb_ws& Wk = ctx->ows[pipe];
        st = pipe ? ctx->aux_stream[pipe - 1] : ctx->stream;
        ctx->tstream = st;
        const uint8_t* d_img; const uint8_t* d_mask = nullptr; const uint16_t* d_depth = nullptr;
        bh_kp* d_kps; uint8_t* d_desc; int* d_nkp;
        if (mem == BH_MEM_HOST) {
            BH_HIP(ctx, hipMemcpyAsync(ctx->st_img.p, img + (size_t)f0 * fpx * channels, fpx * channels * nb, hipMemcpyHostToDevice, st));
            d_img = (const uint8_t*)ctx->st_img.p;
            if (mask) { BH_HIP(ctx, hipMemcpyAsync(ctx->st_mask.p, mask + (size_t)f0 * fpx, fpx * nb, hipMemcpyHostToDevice, st)); d_mask = (const uint8_t*)ctx->st_mask.p; }
            else if (depth) { BH_HIP(ctx, hipMemcpyAsync(ctx->st_depth.p, depth + (size_t)f0 * fpx, fpx * 2 * nb, hipMemcpyHostToDevice, st)); d_depth = (const uint16_t*)ctx->st_depth.p; }
            d_kps = (bh_kp*)ctx->st_kps.p; d_desc = (uint8_t*)ctx->st_desc.p; d_nkp = (int*)ctx->st_nkp.p;
        } else {
            d_img = img + (size_t)f0 * fpx * channels;
            if (mask) d_mask = mask + (size_t)f0 * fpx; else if (depth) d_depth = depth + (size_t)f0 * fpx;
            d_kps = kps + (size_t)f0 * max_kp; d_desc = desc + (size_t)f0 * max_kp * 32; d_nkp = n_kp + f0;
        }
        uint8_t* pyr = (uint8_t*)Wk.pyr.p; uint8_t* mpyr = mask ? (uint8_t*)Wk.mpyr.p : nullptr;
        uint32_t* mbits = (!mask && depth) ? (uint32_t*)Wk.mpyr.p : nullptr;
        uint2* clist = (uint2*)Wk.clist.p; uint8_t* blur = (uint8_t*)Wk.blur.p;
        uint32_t* hist = (uint32_t*)Wk.hist.p;
        int* ccnt = (int*)(hist + (size_t)256 * g.nlevels * chunk);

        BH_HIP(ctx, hipMemsetAsync(ccnt, 0, sizeof(int) * g.nlevels * chunk, st));
        bool gl_fused = false; int gl_sr = 0;
        {
            const int n = (g.lv[0].pitch >> 2) * g.lv[0].h;
            // level 0 + level 1 in one launch (k_gray_l1) when the reference call's case applies, a second level exists and the staged rows fit LDS
            gl_sr = ctx->gl_sr;                                     // (ensure_geom: whole lane items whose level-0 rows fit the staged strip)
            const size_t gl_lds = (size_t)GL_R0 * (g.lv[0].pitch + 16) + (size_t)GL_R0 * (g.lv[0].mpitch + 1) * 4;
            gl_fused = mbits && channels == 3 && (W & 3) == 0 && g.nlevels >= 2 && !ctx->pm_ok && gl_sr >= GL_PR && gl_lds <= 64 * 1024 && !getenv("BH_ORB_GRAY_GENERAL") &&
                       !(getenv("BH_ORB_GRAY_L1") && atoi(getenv("BH_ORB_GRAY_L1")) == 0);
            if (gl_fused) {
                bh_time_begin(ctx, BH_K_GRAY);
                hipLaunchKernelGGL(k_gray_l1, dim3((g.lv[1].h + gl_sr - 1) / gl_sr, nb), dim3(256), gl_lds, st, d_img, d_depth, pyr, g, gl_sr, (const uint32_t*)ctx->tab.p, mbits,
                                   (uint32_t*)ctx->status.p);
                bh_time_end(ctx);
            } else if (mbits && channels == 3 && (W & 3) == 0 && !getenv("BH_ORB_GRAY_GENERAL"))      // the reference call's case
                BH_LAUNCH(ctx, BH_K_GRAY, k_gray_bits, dim3((n + 255) / 256, nb), dim3(256), st, d_img, d_depth, pyr, W, H, g.lv[0].pitch, g.pyr_stride, mbits,
                          g.lv[0].mpitch, g.mstride);
            else
                BH_LAUNCH(ctx, BH_K_GRAY, k_gray_mask, dim3((n + 255) / 256, nb), dim3(256), st, d_img, channels, d_mask, d_depth,
                          pyr, mpyr, W, H, g.lv[0].pitch, g.pyr_stride, mbits, g.lv[0].mpitch, g.mstride);
        }
        // (on request: bit-identical, 0.674 -> 0.668 ms of kernel time and an unchanged step at configs[1] - the six short launches were not what the pyramid costs)
        const bool pairs_on = mbits && !ctx->pm_ok && getenv("BH_ORB_PYR_PAIR") && atoi(getenv("BH_ORB_PYR_PAIR")) == 1;
        for (int l = gl_fused ? 2 : 1; l < g.nlevels; l++) {
            if (pairs_on && l + 1 < g.nlevels && ctx->pp_sr[l] >= GL_PR) {      // levels l and l + 1 in one launch
                const int sr = ctx->pp_sr[l];
                const size_t lds = (size_t)GL_R0 * (g.lv[l].pitch + 16) + (size_t)GL_R0 * (g.lv[l].mpitch + 1) * 4;
                bh_time_begin(ctx, BH_K_PYR);
                hipLaunchKernelGGL(k_pyr_pair, dim3((g.lv[l + 1].h + sr - 1) / sr, nb), dim3(256), lds, st, pyr, g, l, sr, (const uint32_t*)ctx->tab.p, mbits, (uint32_t*)ctx->status.p);
                bh_time_end(ctx);
                l++;
            } else if (mbits && ctx->pm_ok) {
                BH_LAUNCH(ctx, BH_K_PYR, k_pyr_mfma, dim3(ctx->pm[l].tiles_x, ctx->pm[l].tiles_y + (g.lv[l].h + PM_MH - 1) / PM_MH, nb), dim3(256), st, pyr, g, l, (const uint32_t*)ctx->tab.p,
                          (const uint32_t*)ctx->rtab.p, ctx->pm[l], mbits);
            } else if (mbits) {
                const int n = (g.lv[l].pitch >> 2) * ((g.lv[l].h + PD_R - 1) / PD_R);
                BH_LAUNCH(ctx, BH_K_PYR, k_pyr_down_bits, dim3((n + 255) / 256, nb), dim3(256), st, pyr, g, l, (const uint32_t*)ctx->tab.p, mbits);
            } else {
                const int n = (g.lv[l].pitch >> 2) * g.lv[l].h;
                BH_LAUNCH(ctx, BH_K_PYR, k_pyr_down, dim3((n + 255) / 256, nb), dim3(256), st, pyr, mpyr, g, l, (const uint32_t*)ctx->tab.p);
            }
        }
        BH_LAUNCH(ctx, BH_K_FAST, k_fast, dim3(g.ntiles, nb), dim3(256), st, pyr, mpyr, clist, ccnt, blur, g,
                  (const uint32_t*)ctx->tab.p + g.tab_len + 4, (uint32_t*)ctx->status.p, (const uint32_t*)mbits,
                  (const uint32_t*)ctx->tab.p + align_up(g.tab_len + 4 + g.ntiles + 64, 4));
        BH_LAUNCH(ctx, BH_K_SELECT, k_select, dim3(nb, g.nlevels), dim3(SEL_T), st, pyr, clist, ccnt, g, sel_cap, (uint32_t*)Wk.sel_xy.p,
                  (float*)Wk.sel_resp.p, (int*)Wk.sel_cnt.p, (uint32_t*)ctx->status.p);
        BH_LAUNCH(ctx, BH_K_BRIEF, k_describe, dim3(8 * ((max_kp + 3) / 4) * ((nb + 8 * DS_KPW - 1) / (8 * DS_KPW))), dim3(256), st, pyr, blur, g,
                  (const uint32_t*)ctx->tab.p + g.tab_len + 4 + g.ntiles, (float)params->patchSize, sel_cap, max_kp, nb, (const uint32_t*)Wk.sel_xy.p, (const float*)Wk.sel_resp.p,
                  (const int*)Wk.sel_cnt.p, d_kps, d_desc, d_nkp, (uint32_t*)ctx->status.p);
        BH_HIP(ctx, hipGetLastError());
        if (mem == BH_MEM_HOST) {
            BH_HIP(ctx, hipMemcpyAsync(kps + (size_t)f0 * max_kp, d_kps, sizeof(bh_kp) * (size_t)max_kp * nb, hipMemcpyDeviceToHost, st));
            BH_HIP(ctx, hipMemcpyAsync(desc + (size_t)f0 * max_kp * 32, d_desc, (size_t)32 * max_kp * nb, hipMemcpyDeviceToHost, st));
            BH_HIP(ctx, hipMemcpyAsync(n_kp + f0, d_nkp, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
            BH_HIP(ctx, hipStreamSynchronize(st));
        }
        ctx->last_B = nb; ctx->last_has_mask = has_mask; ctx->last_mask_bits = mbits != nullptr; ctx->last_pipe = pipe;
    }
    ctx->tstream = nullptr;
    for (int p = 1; p < npipe; p++) {                            // join: later work on the context's stream sees every chunk
        BH_HIP(ctx, hipEventRecord(ctx->ev_join[p - 1], ctx->aux_stream[p - 1]));
        BH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[p - 1], 0));
    }
    if (mem == BH_MEM_HOST) return bh_ctx_status(ctx);
    return BH_OK;
}

extern "C" int bh_orb_debug_fetch(bh_ctx* ctx, int frame, uint8_t* pyr, uint8_t* mask_pyr, uint8_t* nms, uint8_t* blur)
{
    if (!ctx || !ctx->geom_valid || frame < 0 || frame >= ctx->last_B)
        return ctx ? bh_set_err(ctx, BH_ERR_INVALID, "bh_orb_debug_fetch: no such frame in the last chunk") : BH_ERR_INVALID;
    const size_t n = ctx->geom.pyr_stride, o = (size_t)frame * n;
    const bh_orb_ws& Wk = ctx->ows[ctx->last_pipe];
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (pyr)  BH_HIP(ctx, hipMemcpy(pyr, (uint8_t*)Wk.pyr.p + o, n, hipMemcpyDeviceToHost));
    if (mask_pyr && ctx->last_has_mask && ctx->last_mask_bits) {      // bit-packed on the device: expanded to the byte layout of the image pyramid
        const bh_orb_geom& g = ctx->geom;
        uint32_t* bits = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)g.mstride);
        hipError_t e = hipMemcpy(bits, (uint32_t*)Wk.mpyr.p + (size_t)frame * g.mstride, sizeof(uint32_t) * (size_t)g.mstride, hipMemcpyDeviceToHost);
        memset(mask_pyr, 0, n);
        if (e == hipSuccess)
            for (int l = 0; l < g.nlevels; l++) {
                const bh_level_geom& L = g.lv[l];
                for (int y = 0; y < L.h; y++)
                    for (int x = 0; x < L.w; x++)
                        mask_pyr[L.off + (size_t)y * L.pitch + x] = ((bits[L.moff + (size_t)y * L.mpitch + (x >> 5)] >> (x & 31)) & 1u) ? 255 : 0;
            }
        free(bits);
        if (e != hipSuccess) return bh_set_err(ctx, BH_ERR_HIP, "bh_orb_debug_fetch: %s", hipGetErrorString(e));
    } else if (mask_pyr && ctx->last_has_mask) BH_HIP(ctx, hipMemcpy(mask_pyr, (uint8_t*)Wk.mpyr.p + o, n, hipMemcpyDeviceToHost));
    if (nms) {                                                   // the dense NMS score map no longer exists on the device: rebuild it from the corner lists
        const bh_orb_geom& g = ctx->geom;
        memset(nms, 0, n);
        uint2* lst = (uint2*)malloc(sizeof(uint2) * (size_t)g.list_stride);
        int cnt[BH_MAX_LEVELS];
        const int* ccnt = (const int*)((const uint32_t*)Wk.hist.p + (size_t)256 * g.nlevels * ctx->ws_frames);
        hipError_t e = hipMemcpy(lst, (uint2*)Wk.clist.p + (size_t)frame * g.list_stride, sizeof(uint2) * (size_t)g.list_stride, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(cnt, ccnt + (size_t)frame * g.nlevels, sizeof(int) * g.nlevels, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (int l = 0; l < g.nlevels; l++) {
                const bh_level_geom& L = g.lv[l];
                for (int i = 0; i < cnt[l] && i < L.list_cap; i++) {
                    const uint2 r = lst[L.list_off + i];
                    nms[L.off + (size_t)(r.x >> 16) * L.pitch + (r.x & 0xffffu)] = (uint8_t)r.y;
                }
            }
        free(lst);
        if (e != hipSuccess) return bh_set_err(ctx, BH_ERR_HIP, "bh_orb_debug_fetch: %s", hipGetErrorString(e));
    }
    if (blur) BH_HIP(ctx, hipMemcpy(blur, (uint8_t*)Wk.blur.p + o, n, hipMemcpyDeviceToHost));
    return BH_OK;
}

extern "C" int bh_kp_backproject(bh_ctx* ctx, const bh_kp* kps, const int* n_kp, const uint16_t* depth, int B, int H,
                                 int W, int max_kp, double fx, double fy, double ppx, double ppy,
                                 int64_t* kp_xy, double* cloud_kp, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!kps || !n_kp || !kp_xy || B < 0 || max_kp < 1 || (cloud_kp && !depth))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_kp_backproject: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bh_kp* d_kps = kps; const int* d_n = n_kp; const uint16_t* d_depth = depth;
    long long* d_xy = (long long*)kp_xy; double* d_cl = cloud_kp;
    const size_t nk = (size_t)B * max_kp;
    if (mem == BH_MEM_HOST) {
        BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(bh_kp) * nk));
        BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(int) * B));
        BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(long long) * 2 * nk));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, kps, sizeof(bh_kp) * nk, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, n_kp, sizeof(int) * B, hipMemcpyHostToDevice, st));
        d_kps = (const bh_kp*)ctx->st_a.p; d_n = (const int*)ctx->st_b.p; d_xy = (long long*)ctx->st_c.p;
        if (cloud_kp) {
            BH_TRY(bh_reserve(ctx, &ctx->st_d, (size_t)B * H * W * 2));
            BH_TRY(bh_reserve(ctx, &ctx->st_e, sizeof(double) * 3 * nk));
            BH_HIP(ctx, hipMemcpyAsync(ctx->st_d.p, depth, (size_t)B * H * W * 2, hipMemcpyHostToDevice, st));
            d_depth = (const uint16_t*)ctx->st_d.p; d_cl = (double*)ctx->st_e.p;
        }
    }
    BH_LAUNCH(ctx, BH_K_BACKPROJ, k_kp_backproject, dim3((max_kp + 255) / 256, B), dim3(256), st, d_kps, d_n, d_depth, H, W, max_kp,
                       fx, fy, ppx, ppy, d_xy, d_cl);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(kp_xy, d_xy, sizeof(long long) * 2 * nk, hipMemcpyDeviceToHost, st));
        if (cloud_kp) BH_HIP(ctx, hipMemcpyAsync(cloud_kp, d_cl, sizeof(double) * 3 * nk, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}
