/* A plain C99 consumer of the C ABI (include/boslam_hip.h): no Python, no C++, no HIP headers - what a cgo / JNI / ctypes binding of the
 * reference's call sites sees.  Test infrastructure: tests/test_host_cpu.py compiles it (the header must be valid C), tests/test_orb_gpu.py
 * runs it on the GPU box and compares what it prints with the Python mirror on the same frames.
 *
 *   consumer <libboslam_hip.so> <frames.bin> H W
 * frames.bin = u8[2][H][W][3] BGR followed by u16[2][H][W] depth.  It runs, as the reference's frame loop would:
 *   detectAndCompute (camera.py:141) on both frames -> BFMatcher(crossCheck).match(des0, des1) (slam/tracking.py:56)
 * and prints n_kp of each frame, the number of matches and FNV-1a checksums of the keypoint records, descriptors and match lists. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "boslam_hip.h"

typedef int (*ctx_create_fn)(int, bh_ctx**);
typedef void (*ctx_destroy_fn)(bh_ctx*);
typedef const char* (*last_error_fn)(const bh_ctx*);
typedef int (*orb_fn)(bh_ctx*, const uint8_t*, int, const uint8_t*, const uint16_t*, int, int, int, const bh_orb_params*, int, bh_kp*, uint8_t*, int*, int);
typedef int (*match_fn)(bh_ctx*, const uint8_t*, const int*, int, const uint8_t*, const int*, int, int, int*, int*, int*, int*, int, int);

static unsigned long long fnv(const void* p, size_t n, unsigned long long h)
{
    const unsigned char* b = (const unsigned char*)p;
    size_t i;
    for (i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    void* lib;
    ctx_create_fn ctx_create; ctx_destroy_fn ctx_destroy; last_error_fn last_error; orb_fn orb; match_fn match;
    bh_ctx* ctx = NULL;
    bh_orb_params P;
    int H, W, rc, f, max_kp = 1200, n_kp[2], n_match = 0;
    size_t npx;
    uint8_t* img; uint16_t* depth; bh_kp* kps; uint8_t* desc; int *qi, *ti, *di;
    FILE* fp;
    unsigned long long hk = 1469598103934665603ull, hd = hk, hm = hk;
    if (argc != 5) { fprintf(stderr, "usage: consumer lib.so frames.bin H W\n"); return 2; }
    H = atoi(argv[3]); W = atoi(argv[4]); npx = (size_t)H * W;
    lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    *(void**)&ctx_create = dlsym(lib, "bh_ctx_create"); *(void**)&ctx_destroy = dlsym(lib, "bh_ctx_destroy");
    *(void**)&last_error = dlsym(lib, "bh_last_error"); *(void**)&orb = dlsym(lib, "bh_orb_detect_describe");
    *(void**)&match = dlsym(lib, "bh_match_hamming_crosscheck");
    if (!ctx_create || !ctx_destroy || !last_error || !orb || !match) { fprintf(stderr, "missing symbol\n"); return 4; }
    img = (uint8_t*)malloc(2 * npx * 3); depth = (uint16_t*)malloc(2 * npx * sizeof(uint16_t));
    kps = (bh_kp*)calloc((size_t)2 * max_kp, sizeof(bh_kp)); desc = (uint8_t*)calloc((size_t)2 * max_kp, 32);
    qi = (int*)malloc(sizeof(int) * max_kp); ti = (int*)malloc(sizeof(int) * max_kp); di = (int*)malloc(sizeof(int) * max_kp);
    fp = fopen(argv[2], "rb");
    if (!fp || fread(img, 1, 2 * npx * 3, fp) != 2 * npx * 3 || fread(depth, sizeof(uint16_t), 2 * npx, fp) != 2 * npx) { fprintf(stderr, "frames.bin: short read\n"); return 5; }
    fclose(fp);
    rc = ctx_create(0, &ctx);
    if (rc != BH_OK) { fprintf(stderr, "bh_ctx_create: %d\n", rc); return 6; }
    P.nfeatures = 1000; P.scaleFactor = 1.2f; P.nlevels = 8; P.edgeThreshold = 31; P.firstLevel = 0; P.patchSize = 31; P.fastThreshold = 20;
    rc = orb(ctx, img, 3, NULL, depth, 2, H, W, &P, max_kp, kps, desc, n_kp, BH_MEM_HOST);
    if (rc != BH_OK) { fprintf(stderr, "bh_orb_detect_describe: %d %s\n", rc, last_error(ctx)); return 7; }
    rc = match(ctx, desc, &n_kp[0], max_kp, desc + (size_t)max_kp * 32, &n_kp[1], max_kp, 1, qi, ti, di, &n_match, max_kp, BH_MEM_HOST);
    if (rc != BH_OK) { fprintf(stderr, "bh_match_hamming_crosscheck: %d %s\n", rc, last_error(ctx)); return 8; }
    for (f = 0; f < 2; f++) { hk = fnv(kps + (size_t)f * max_kp, sizeof(bh_kp) * (size_t)n_kp[f], hk); hd = fnv(desc + (size_t)f * max_kp * 32, 32 * (size_t)n_kp[f], hd); }
    hm = fnv(qi, sizeof(int) * (size_t)n_match, hm); hm = fnv(ti, sizeof(int) * (size_t)n_match, hm); hm = fnv(di, sizeof(int) * (size_t)n_match, hm);
    printf("n_kp %d %d n_match %d kps %016llx desc %016llx match %016llx\n", n_kp[0], n_kp[1], n_match, hk, hd, hm);
    ctx_destroy(ctx);
    free(img); free(depth); free(kps); free(desc); free(qi); free(ti); free(di);
    dlclose(lib);
    return 0;
}
