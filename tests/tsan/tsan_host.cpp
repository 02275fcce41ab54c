// Host-side code paths of libboslam_hip.so from two threads under ThreadSanitizer (tools/tsan_host.sh builds the library's HOST code
// with -fsanitize=thread; the device code cannot be instrumented and sanitizers do not run on the GPU pool).  Without a GPU, as on the
// build box, this exercises what the two reference threads (slam/main.py:37-47) share before any device work: the geometry /
// table builders (bh_orb_geometry), context creation and its failure path, the error string.  With a GPU it also runs one ORB call and
// one BoW handle per thread on its own context.
#include <thread>
#include <cstdio>
#include <cstring>
#include <vector>
#include "boslam_hip.h"

int main()
{
    int bad = 0, have_gpu = 0;
    auto work = [&](int id) {
        for (int i = 0; i < 100; i++) {
            bh_orb_params p; memset(&p, 0, sizeof(p));
            p.nfeatures = 500 + id; p.scaleFactor = 1.2f; p.nlevels = 8; p.edgeThreshold = 31; p.firstLevel = 0; p.patchSize = 31; p.fastThreshold = 20;
            int w[16], h[16], pitch[16], off[16], nf[16], tot = 0;
            if (bh_orb_geometry(480, 640, &p, w, h, pitch, off, nf, &tot) != BH_OK || w[1] != 533 || h[7] != 134) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
            bh_ctx* c = nullptr;
            const int rc = bh_ctx_create(0, &c);
            (void)bh_last_error(nullptr);
            if (rc != BH_OK) continue;
            __atomic_store_n(&have_gpu, 1, __ATOMIC_RELAXED);
            if (i < 3) {                                                  // a device is present: one front-end call and one BoW handle on this thread's context
                std::vector<uint8_t> img(480 * 640, (uint8_t)(40 + id)), desc(32 * 1024);
                for (size_t k = 0; k < img.size(); k++) img[k] = (uint8_t)((k * 2654435761u >> 13) + id);
                std::vector<bh_kp> kps(1024); int n = 0;
                if (bh_orb_detect_describe(c, img.data(), 1, nullptr, nullptr, 1, 480, 640, &p, 1024, kps.data(), desc.data(), &n, BH_MEM_HOST) != BH_OK)
                    __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
                bh_bow* bw = nullptr;
                if (bh_bow_create(c, &bw) == BH_OK) bh_bow_destroy(bw); else __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
            }
            bh_ctx_destroy(c);
        }
    };
    std::thread a(work, 0), b(work, 1);
    a.join(); b.join();
    printf("tsan host test: bad=%d gpu=%d\n", bad, have_gpu);
    return bad ? 1 : 0;
}
