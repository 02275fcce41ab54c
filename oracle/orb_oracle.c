/*
 * orb_oracle.c — CPU restatement of the ORB detect+describe path the reference runs at
 *   camera.py:49-62  (cv2.ORB_create(nfeatures, scaleFactor=1.2, nlevels=8, edgeThreshold=31,
 *                     firstLevel=0, patchSize=31, fastThreshold=20); WTA_K=2, HARRIS_SCORE implicit)
 *   camera.py:141    (detectAndCompute(frame_bgr, mask=(depth>0)*255))
 *   camera.py:142    (kp_arr = int-truncated keypoint coordinates)
 *   camera.py:90-127 (depth -> 3-D back-projection gathered at the keypoints)
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * this.  The product path (boslam_amd/) never links, imports or executes anything in oracle/.
 *
 * PARITY UNPINNED: the arithmetic lives in OpenCV (pinned commit 976ee4468276d8f90403cc0885b8fd8576c58f4d,
 * docker/install_opencv.sh:14), which is absent from /root/reference and from this image, and the
 * reference ships no golden vectors for this path.  This file restates OpenCV's published
 * algorithm (modules/features2d/src/{orb,fast,fast_score}.cpp, imgproc resize/smooth/color) stage by stage:
 *
 *   gray      : Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14                       (COLOR_BGR2GRAY, 8U)
 *   pyramid   : size_l = (cvRound(W/s_l), cvRound(H/s_l)), s_l=(float)pow(1.2,l); level l>0 resized from
 *               level l-1 with INTER_LINEAR_EXACT (8.8 fixed-point horizontal pass, 16.16 vertical pass,
 *               coefficients = cvRound(frac*256)); mask pyramid: same resize, then keep only ==255
 *   FAST-9/16 : threshold t; score = max over 9-arcs of min|v-p| - 1 (cornerScore<16>); 3x3 strict NMS;
 *               scan rows/cols 3..dim-4; raster emission
 *   filters   : mask[y][x]==0 removed; 31-px image-border filter; retainBest(2*n_l) on the FAST score
 *   Harris    : 7x7 block of 3x3 Sobel products, k=0.04, float formula of HarrisResponses()
 *   retainBest(n_l) on Harris; ties with the n-th value are all kept
 *   IC angle  : m10,m01 over the 749-px disc (u_max table), fastAtan2 (float polynomial, degrees)
 *   blur      : GaussianBlur 7x7 sigma 2 on each level, 8-bit symmetric fixed-point path
 *               (float kernel -> cvRound(k*256) ints, (sum + 32768) >> 16), BORDER_REFLECT_101
 *   BRIEF     : 256 rotated pixel-pair tests (bit_pattern_31_), cvRound = round-half-even
 *
 * CANONICAL ORDER: OpenCV's emission order inside a level is whatever std::nth_element/std::partition
 * leave (libstdc++-specific).  The SET is what this oracle pins; it emits keypoints level-major and,
 * inside a level, in raster order (y, then x).  DESIGN.md states this as the parity contract.
 *
 * All float arithmetic here is IEEE single, one rounding per operation: build with -ffp-contract=off.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include "brief_pattern.h"

#define ORC_MAX_LEVELS 16

typedef struct {
    int nfeatures; float scaleFactor; int nlevels; int edgeThreshold;
    int firstLevel; int patchSize; int fastThreshold;
} orc_orb_params;

typedef struct { float x, y, size, angle, response; int octave; } orc_kp;

/* ---- rounding helpers ------------------------------------------------------------------------ */
static int orc_round_f(float v)  { return (int)lrintf(v); }   /* cvRound(float): round-half-even */
static int orc_round_d(double v) { return (int)lrint(v);  }   /* cvRound(double) */

/* ---- level geometry (orb.cpp detectAndCompute: layerScale / layerInfo) ------------------------- */
typedef struct { int w, h; size_t off; float scale; } orc_level;

static int orc_levels(int W, int H, const orc_orb_params* p, orc_level* L, size_t* total)
{
    if (p->nlevels < 1 || p->nlevels > ORC_MAX_LEVELS || p->firstLevel != 0 || p->patchSize != 31)
        return -1;
    size_t off = 0;
    for (int l = 0; l < p->nlevels; l++) {
        float s = (float)pow((double)p->scaleFactor, (double)l);
        L[l].scale = s;
        L[l].w = orc_round_f((float)W / s);
        L[l].h = orc_round_f((float)H / s);
        L[l].off = off;
        off += (size_t)L[l].w * L[l].h;
    }
    *total = off;
    return 0;
}

/* features per level (orb.cpp computeKeyPoints) */
static void orc_features_per_level(const orc_orb_params* p, int* n)
{
    float factor = (float)(1.0 / p->scaleFactor);
    float nd = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p->nlevels));
    int sum = 0;
    for (int l = 0; l < p->nlevels - 1; l++) {
        n[l] = orc_round_f(nd);
        sum += n[l];
        nd *= factor;
    }
    int last = p->nfeatures - sum;
    n[p->nlevels - 1] = last > 0 ? last : 0;
}

/* ---- BGR -> gray --------------------------------------------------------------------------------- */
void orc_bgr2gray(const uint8_t* bgr, int n, uint8_t* gray)
{
    for (int i = 0; i < n; i++)
        gray[i] = (uint8_t)((bgr[3*i] * 1868 + bgr[3*i+1] * 9617 + bgr[3*i+2] * 4899 + 8192) >> 14);
}

/* ---- INTER_LINEAR_EXACT resize, 8U single channel -------------------------------------------------
 * per axis: fval = scale*(d+0.5)-0.5 with scale = 1/(dst/src) in double; ival = floor(fval);
 * inside [0, src-2]: offsets ival, coeffs c1 = cvRound((fval-ival)*256), c0 = 256-c1;
 * ival >= src-1 : replicate the last sample; ival < 0 : replicate the first. */
static void orc_lin_coeffs(int src, int dst, int* ofs, int* c0, int* c1)
{
    double inv_scale = (double)dst / (double)src;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; d++) {
        double fval = scale * ((double)d + 0.5) - 0.5;
        int ival = (int)floor(fval);
        if (ival >= 0 && src > 1) {
            if (ival < src - 1) {
                ofs[d] = ival;
                c1[d] = orc_round_d((fval - (double)ival) * 256.0);
                c0[d] = 256 - c1[d];
            } else { ofs[d] = src - 1; c0[d] = 256; c1[d] = 0; }
        } else { ofs[d] = 0; c0[d] = 256; c1[d] = 0; }
    }
}

void orc_resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    int* xo = malloc(sizeof(int) * dw * 3); int* xc0 = xo + dw; int* xc1 = xc0 + dw;
    int* yo = malloc(sizeof(int) * dh * 3); int* yc0 = yo + dh; int* yc1 = yc0 + dh;
    orc_lin_coeffs(sw, dw, xo, xc0, xc1);
    orc_lin_coeffs(sh, dh, yo, yc0, yc1);
    for (int y = 0; y < dh; y++) {
        const uint8_t* r0 = src + (size_t)yo[y] * sw;
        const uint8_t* r1 = src + (size_t)(yo[y] + 1 < sh ? yo[y] + 1 : yo[y]) * sw;
        for (int x = 0; x < dw; x++) {
            int a = xo[x], b = (a + 1 < sw) ? a + 1 : a;
            uint32_t h0 = (uint32_t)xc0[x] * r0[a] + (uint32_t)xc1[x] * r0[b];   /* 8.8 fixed */
            uint32_t h1 = (uint32_t)xc0[x] * r1[a] + (uint32_t)xc1[x] * r1[b];
            uint32_t v = (uint32_t)yc0[y] * h0 + (uint32_t)yc1[y] * h1;            /* 16.16 fixed */
            uint32_t o = (v + 32768u) >> 16;
            dst[(size_t)y * dw + x] = (uint8_t)(o > 255 ? 255 : o);
        }
    }
    free(xo); free(yo);
}

/* ---- FAST-9/16 --------------------------------------------------------------------------------- */
static const int orc_circ[16][2] = {
    {0,3},{1,3},{2,2},{3,1},{3,0},{3,-1},{2,-2},{1,-3},{0,-3},{-1,-3},{-2,-2},{-3,-1},{-3,0},{-3,1},{-2,2},{-1,3}
};

/* raw (pre-NMS) score map: 0 = not a corner, else cornerScore<16> stored as uchar (fast.cpp) */
void orc_fast_score_map(const uint8_t* img, int w, int h, int thr, uint8_t* score)
{
    memset(score, 0, (size_t)w * h);
    if (thr < 0) thr = 0;
    if (thr > 255) thr = 255;
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            int v = img[(size_t)y * w + x];
            int d[25];
            for (int k = 0; k < 25; k++) {
                int kk = k & 15;
                d[k] = v - img[(size_t)(y + orc_circ[kk][1]) * w + x + orc_circ[kk][0]];
            }
            /* corner test: >K=8 contiguous strictly darker (p < v-thr  <=> d > thr) or brighter (d < -thr) */
            int is_corner = 0, cnt = 0;
            for (int k = 0; k < 25; k++) { if (d[k] > thr) { if (++cnt > 8) { is_corner = 1; break; } } else cnt = 0; }
            if (!is_corner) {
                cnt = 0;
                for (int k = 0; k < 25; k++) { if (d[k] < -thr) { if (++cnt > 8) { is_corner = 1; break; } } else cnt = 0; }
            }
            if (!is_corner) continue;
            /* cornerScore<16> (fast_score.cpp) */
            int a0 = thr;
            for (int k = 0; k < 16; k += 2) {
                int a = d[k+1] < d[k+2] ? d[k+1] : d[k+2];
                a = a < d[k+3] ? a : d[k+3];
                if (a <= a0) continue;
                for (int j = 4; j <= 8; j++) a = a < d[k+j] ? a : d[k+j];
                int m = a < d[k] ? a : d[k];      a0 = a0 > m ? a0 : m;
                m = a < d[k+9] ? a : d[k+9];      a0 = a0 > m ? a0 : m;
            }
            int b0 = -a0;
            for (int k = 0; k < 16; k += 2) {
                int b = d[k+1] > d[k+2] ? d[k+1] : d[k+2];
                b = b > d[k+3] ? b : d[k+3];
                b = b > d[k+4] ? b : d[k+4];
                b = b > d[k+5] ? b : d[k+5];
                if (b >= b0) continue;
                b = b > d[k+6] ? b : d[k+6];
                b = b > d[k+7] ? b : d[k+7];
                b = b > d[k+8] ? b : d[k+8];
                int m = b > d[k] ? b : d[k];      b0 = b0 < m ? b0 : m;
                m = b > d[k+9] ? b : d[k+9];      b0 = b0 < m ? b0 : m;
            }
            score[(size_t)y * w + x] = (uint8_t)(-b0 - 1);
        }
    }
}

/* 3x3 strict-greater NMS on the raw score map; output: nms[y][x] = score if kept else 0 */
void orc_fast_nms(const uint8_t* score, int w, int h, uint8_t* nms)
{
    memset(nms, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = score[(size_t)y * w + x];
            if (!s) continue;
            int ok = 1;
            for (int dy = -1; dy <= 1 && ok; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    if (s <= score[(size_t)(y + dy) * w + x + dx]) { ok = 0; break; }
                }
            if (ok) nms[(size_t)y * w + x] = (uint8_t)s;
        }
}

/* ---- Harris response (orb.cpp HarrisResponses, blockSize 7) ------------------------------------ */
static float orc_harris(const uint8_t* img, int w, int x0, int y0)
{
    const int bs = 7, r = bs / 2;
    float scale = 1.f / ((1 << 2) * bs * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < bs; i++)
        for (int j = 0; j < bs; j++) {
            const uint8_t* p = img + (size_t)(y0 - r + i) * w + (x0 - r + j);
            int Ix = (p[1] - p[-1]) * 2 + (p[-w+1] - p[-w-1]) + (p[w+1] - p[w-1]);
            int Iy = (p[w] - p[-w]) * 2 + (p[w-1] - p[-w-1]) + (p[w+1] - p[-w+1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    float fa = (float)a, fb = (float)b, fc = (float)c;
    float t1 = fa * fb;
    float t2 = fc * fc;
    float s  = fa + fb;
    float t3 = (0.04f * s) * s;
    return ((t1 - t2) - t3) * scale_sq_sq;
}

/* ---- intensity-centroid angle (orb.cpp ICAngles + fastAtan2) ------------------------------------ */
static void orc_umax(int* umax /* [17] */)
{
    const int hp = 15;
    int v, v0, vmax = (int)floor(hp * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceil(hp * sqrtf(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = orc_round_d(sqrt((double)hp * hp - v * v));
    for (v = hp, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

float orc_fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static float orc_ic_angle(const uint8_t* img, int w, int cx, int cy, const int* umax)
{
    const uint8_t* c = img + (size_t)cy * w + cx;
    int m01 = 0, m10 = 0;
    for (int u = -15; u <= 15; ++u) m10 += u * c[u];
    for (int v = 1; v <= 15; ++v) {
        int vsum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = c[u + v * w], vm = c[u - v * w];
            vsum += (vp - vm);
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return orc_fast_atan2((float)m01, (float)m10);
}

/* ---- 7x7 sigma-2 Gaussian, 8-bit fixed-point symmetric path ------------------------------------- */
void orc_gauss7_coeffs(int* k /* [7] */)
{
    float cf[7]; double sum = 0, sigma = 2.0, s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(s2 * x * x); sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); k[i] = orc_round_f(cf[i] * 256.f); }
}

static int orc_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}

void orc_gauss_blur7(const uint8_t* src, int w, int h, uint8_t* dst)
{
    int k[7]; orc_gauss7_coeffs(k);
    int32_t* tmp = malloc(sizeof(int32_t) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * src[(size_t)y * w + orc_reflect101(x + i - 3, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * tmp[(size_t)orc_reflect101(y + i - 3, h) * w + x];
            int32_t o = (s + 32768) >> 16;
            dst[(size_t)y * w + x] = (uint8_t)(o < 0 ? 0 : o > 255 ? 255 : o);
        }
    free(tmp);
}

/* ---- steered BRIEF (orb.cpp computeOrbDescriptors, WTA_K=2) -------------------------------------- */
static void orc_brief(const uint8_t* blur, int w, int cx, int cy, float angle_deg, uint8_t* desc)
{
    float angle = angle_deg;
    angle *= (float)(3.1415926535897932384626433832795 / 180.f);
    float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t* center = blur + (size_t)cy * w + cx;
    const signed char* pat = orc_bit_pattern_31;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            const signed char* q = pat + (size_t)(i * 8 + j) * 4;
            float x0 = q[0] * a - q[1] * b, y0 = q[0] * b + q[1] * a;
            float x1 = q[2] * a - q[3] * b, y1 = q[2] * b + q[3] * a;
            int t0 = center[orc_round_f(y0) * w + orc_round_f(x0)];
            int t1 = center[orc_round_f(y1) * w + orc_round_f(x1)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ---- retainBest threshold: n-th largest value; returns 1 if filtering applies ---------------------- */
static int orc_cmp_desc_f(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}
static int orc_nth_largest(const float* v, int m, int n, float* thr)
{
    if (n < 0 || m <= n) return 0;
    if (n == 0) { *thr = INFINITY; return 1; }          /* retainBest(0) clears the list */
    float* t = malloc(sizeof(float) * m);
    memcpy(t, v, sizeof(float) * m);
    qsort(t, m, sizeof(float), orc_cmp_desc_f);
    *thr = t[n - 1];
    free(t);
    return 1;
}

/* ---- pyramid builder shared by the full pipeline and the debug entry ----------------------------- */
static int orc_build(const uint8_t* img, int channels, const uint8_t* mask, int H, int W,
                     const orc_orb_params* p, orc_level* L, uint8_t** pyr, uint8_t** mpyr)
{
    size_t total;
    if (orc_levels(W, H, p, L, &total)) return -1;
    *pyr = malloc(total);
    *mpyr = mask ? malloc(total) : NULL;
    if (channels == 3) orc_bgr2gray(img, W * H, *pyr);
    else if (channels == 1) memcpy(*pyr, img, (size_t)W * H);
    else return -1;
    if (mask) memcpy(*mpyr, mask, (size_t)W * H);
    for (int l = 1; l < p->nlevels; l++) {
        orc_resize_linear_exact(*pyr + L[l-1].off, L[l-1].w, L[l-1].h, *pyr + L[l].off, L[l].w, L[l].h);
        if (mask) {
            uint8_t* m = *mpyr + L[l].off;
            orc_resize_linear_exact(*mpyr + L[l-1].off, L[l-1].w, L[l-1].h, m, L[l].w, L[l].h);
            for (size_t i = 0; i < (size_t)L[l].w * L[l].h; i++) m[i] = m[i] > 254 ? m[i] : 0; /* THRESH_TOZERO */
        }
    }
    return 0;
}

/* Query level geometry: w[l], h[l], off[l], n_per_level[l]; returns total pyramid bytes (or 0). */
size_t orc_orb_geometry(int H, int W, const orc_orb_params* p, int* w, int* h, size_t* off, int* nfeat)
{
    orc_level L[ORC_MAX_LEVELS]; size_t total;
    if (orc_levels(W, H, p, L, &total)) return 0;
    int n[ORC_MAX_LEVELS]; orc_features_per_level(p, n);
    for (int l = 0; l < p->nlevels; l++) { w[l] = L[l].w; h[l] = L[l].h; off[l] = L[l].off; nfeat[l] = n[l]; }
    return total;
}

/*
 * Full detectAndCompute for one frame.
 *   img: u8[H][W][channels] (channels 3 = BGR, 1 = gray); mask: u8[H][W] or NULL
 *   kps/desc: capacity max_kp; *n_kp receives the number produced (may exceed nfeatures on ties).
 *   dbg_*: optional (may be NULL) dumps of intermediates, tightly packed per level at orc_orb_geometry offsets:
 *     dbg_pyr (gray pyramid), dbg_mask (mask pyramid, only if mask), dbg_nms (post-NMS/mask/border score map),
 *     dbg_blur (blurred pyramid)
 * returns 0 ok, -1 bad params, -2 capacity exceeded (output truncated in canonical order).
 */
int orc_orb_detect_describe_dbg(const uint8_t* img, int channels, const uint8_t* mask, int H, int W,
                                const orc_orb_params* p, int max_kp, orc_kp* kps, uint8_t* desc, int* n_kp,
                                uint8_t* dbg_pyr, uint8_t* dbg_mask, uint8_t* dbg_nms, uint8_t* dbg_blur)
{
    orc_level L[ORC_MAX_LEVELS];
    uint8_t *pyr = NULL, *mpyr = NULL;
    if (orc_build(img, channels, mask, H, W, p, L, &pyr, &mpyr)) { free(pyr); free(mpyr); return -1; }
    size_t total = L[p->nlevels-1].off + (size_t)L[p->nlevels-1].w * L[p->nlevels-1].h;
    int nfl[ORC_MAX_LEVELS]; orc_features_per_level(p, nfl);
    int umax[17]; orc_umax(umax);
    uint8_t* blur = malloc(total);
    int N = 0, rc = 0;
    int et = p->edgeThreshold;

    if (dbg_pyr) memcpy(dbg_pyr, pyr, total);
    if (dbg_mask && mpyr) memcpy(dbg_mask, mpyr, total);

    for (int l = 0; l < p->nlevels; l++) {
        int w = L[l].w, h = L[l].h;
        const uint8_t* im = pyr + L[l].off;
        uint8_t* sc  = malloc((size_t)w * h);
        uint8_t* nms = malloc((size_t)w * h);
        orc_fast_score_map(im, w, h, p->fastThreshold, sc);
        orc_fast_nms(sc, w, h, nms);
        /* runByPixelsMask, runByImageBorder(edgeThreshold) */
        int m = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                size_t i = (size_t)y * w + x;
                if (!nms[i]) continue;
                int keep = 1;
                if (mpyr && mpyr[L[l].off + i] == 0) keep = 0;
                if (!(x >= et && x < w - et && y >= et && y < h - et)) keep = 0;
                if (!keep) nms[i] = 0; else m++;
            }
        if (dbg_nms) memcpy(dbg_nms + L[l].off, nms, (size_t)w * h);
        /* candidates in raster order */
        int* cx = malloc(sizeof(int) * (m + 1)); int* cy = malloc(sizeof(int) * (m + 1));
        float* resp = malloc(sizeof(float) * (m + 1));
        int c = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                if (nms[(size_t)y * w + x]) { cx[c] = x; cy[c] = y; resp[c] = (float)nms[(size_t)y * w + x]; c++; }
        /* retainBest(2*n_l) on FAST score */
        float thr;
        if (orc_nth_largest(resp, m, 2 * nfl[l], &thr)) {
            int k = 0;
            if (2 * nfl[l] > 0)
                for (int i = 0; i < m; i++) if (resp[i] >= thr) { cx[k] = cx[i]; cy[k] = cy[i]; k++; }
            m = k;
        }
        /* Harris, retainBest(n_l) */
        for (int i = 0; i < m; i++) resp[i] = orc_harris(im, w, cx[i], cy[i]);
        if (orc_nth_largest(resp, m, nfl[l], &thr)) {
            int k = 0;
            if (nfl[l] > 0)
                for (int i = 0; i < m; i++) if (resp[i] >= thr) { cx[k] = cx[i]; cy[k] = cy[i]; resp[k] = resp[i]; k++; }
            m = k;
        }
        /* angle, scale to level-0 coordinates */
        for (int i = 0; i < m; i++) {
            if (N >= max_kp) { rc = -2; break; }
            orc_kp* o = &kps[N];
            o->angle = orc_ic_angle(im, w, cx[i], cy[i], umax);
            o->x = (float)cx[i] * L[l].scale;
            o->y = (float)cy[i] * L[l].scale;
            o->size = (float)p->patchSize * L[l].scale;
            o->response = resp[i];
            o->octave = l;
            N++;
        }
        free(sc); free(nms); free(cx); free(cy); free(resp);
        if (rc) break;
    }
    /* blur each level, then descriptors */
    for (int l = 0; l < p->nlevels; l++)
        orc_gauss_blur7(pyr + L[l].off, L[l].w, L[l].h, blur + L[l].off);
    if (dbg_blur) memcpy(dbg_blur, blur, total);
    for (int i = 0; i < N; i++) {
        int l = kps[i].octave;
        float s = 1.f / L[l].scale;
        int cx = orc_round_f(kps[i].x * s), cy = orc_round_f(kps[i].y * s);
        orc_brief(blur + L[l].off, L[l].w, cx, cy, kps[i].angle, desc + (size_t)i * 32);
    }
    *n_kp = N;
    free(pyr); free(mpyr); free(blur);
    return rc;
}

int orc_orb_detect_describe(const uint8_t* img, int channels, const uint8_t* mask, int H, int W,
                            const orc_orb_params* p, int max_kp, orc_kp* kps, uint8_t* desc, int* n_kp)
{
    return orc_orb_detect_describe_dbg(img, channels, mask, H, W, p, max_kp, kps, desc, n_kp, 0, 0, 0, 0);
}

/* Batched convenience used by the cpu_baseline leg: frames processed independently (OpenMP when built -fopenmp).
 * depth (u16, may be NULL) supplies mask=(depth>0)*255 exactly as camera.py:141 builds it. */
int orc_orb_batch(const uint8_t* imgs, int channels, const uint16_t* depth, int B, int H, int W,
                  const orc_orb_params* p, int max_kp, orc_kp* kps, uint8_t* desc, int* n_kp)
{
    int rc_all = 0;
    #pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; b++) {
        uint8_t* mask = NULL;
        if (depth) {
            mask = malloc((size_t)H * W);
            for (size_t i = 0; i < (size_t)H * W; i++) mask[i] = depth[(size_t)b * H * W + i] > 0 ? 255 : 0;
        }
        int rc = orc_orb_detect_describe(imgs + (size_t)b * H * W * channels, channels, mask, H, W, p, max_kp,
                                         kps + (size_t)b * max_kp, desc + (size_t)b * max_kp * 32, &n_kp[b]);
        if (rc) {
            #pragma omp critical
            rc_all = rc;
        }
        free(mask);
    }
    return rc_all;
}

/* camera.py:142 — kp_arr[i] = (int(pt.x), int(pt.y)) truncation toward zero, int64 (x,y) */
void orc_kp_to_int(const orc_kp* kps, int n, int64_t* xy)
{
    for (int i = 0; i < n; i++) { xy[2*i] = (int64_t)kps[i].x; xy[2*i+1] = (int64_t)kps[i].y; }
}

/* camera.py:90-127 — z = depth/1000 ; x=(u-ppx)/fx*z ; y=(v-ppy)/fy*z ; gathered at kp_arr (u=x, v=y) */
void orc_backproject(const uint16_t* depth, int W, const int64_t* xy, int n,
                     double fx, double fy, double ppx, double ppy, double* xyz)
{
    for (int i = 0; i < n; i++) {
        int64_t u = xy[2*i], v = xy[2*i+1];
        double z = (double)depth[v * W + u] / 1000;
        double xn = ((double)(int16_t)u - ppx) / fx;   /* u,v meshgrid are int16 in camera.py:95-96 */
        double yn = ((double)(int16_t)v - ppy) / fy;
        xyz[3*i] = xn * z; xyz[3*i+1] = yn * z; xyz[3*i+2] = z;
    }
}

/* number of OpenMP threads the batched entry points use (1 when built without -fopenmp) */
#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
int orc_num_threads(void) { return 1; }
void orc_set_num_threads(int n) { (void)n; }
#endif

/* ---- single-stage entry points for the unit tests (tests/test_oracle_cpu.py) ---------------------------------- */
float orc_harris_at(const uint8_t* img, int w, int x, int y) { return orc_harris(img, w, x, y); }
float orc_ic_angle_at(const uint8_t* img, int w, int x, int y) { int um[17]; orc_umax(um); return orc_ic_angle(img, w, x, y, um); }
void  orc_umax_table(int* out16) { int um[17]; orc_umax(um); for (int i = 0; i < 16; i++) out16[i] = um[i]; }
void  orc_brief_at(const uint8_t* blur, int w, int x, int y, float angle_deg, uint8_t* desc32) { orc_brief(blur, w, x, y, angle_deg, desc32); }
const signed char* orc_pattern(void) { return orc_bit_pattern_31; }
