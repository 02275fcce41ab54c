/*
 * match_oracle.c — CPU restatement of cv2.BFMatcher_create(cv2.NORM_HAMMING, crossCheck=True).match(q, t)
 * as called at slam/tracking.py:45,56,121, slam/local_mapping.py:21, slam/covisibility_graph.py:34 and
 * experiments/pnp_two_way_tracking.py:11,42.
 *
 * TEST INFRASTRUCTURE (see orb_oracle.c header).  PARITY UNPINNED vs OpenCV (absent; no golden vectors).
 *
 * Published algorithm (OpenCV core/src/batch_distance.cpp, crosscheck branch; features2d/src/matchers.cpp):
 *   1. for every TRAIN row j: its nearest QUERY row, scanning queries in order and updating on strict '<'
 *      (first minimum wins)  -> tq[j], td[j]
 *   2. dist[i]=INT_MAX, nidx[i]=-1 for every query i; for j ascending: i=tq[j]; if td[j] < dist[i]:
 *      dist[i]=td[j]; nidx[i]=j
 *   3. result = (i, nidx[i], (float)dist[i]) for i ascending where nidx[i] >= 0
 * This is NOT the symmetric mutual nearest neighbour: a query's own best train is never computed.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

static int orc_hamming32(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int k = 0; k < 32; k += 8) {
        uint64_t x, y;
        memcpy(&x, a + k, 8); memcpy(&y, b + k, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

int orc_hamming_distance(const uint8_t* a, const uint8_t* b) { return orc_hamming32(a, b); }

/* q: u8[nq][32], t: u8[nt][32]; outputs sized min(nq,nt); returns number of matches */
int orc_match_crosscheck(const uint8_t* q, int nq, const uint8_t* t, int nt,
                         int* qidx, int* tidx, int* dist)
{
    if (nq <= 0 || nt <= 0) return 0;
    int* best_d = malloc(sizeof(int) * nq);
    int* best_j = malloc(sizeof(int) * nq);
    for (int i = 0; i < nq; i++) { best_d[i] = INT_MAX; best_j[i] = -1; }
    for (int j = 0; j < nt; j++) {
        int td = INT_MAX, tq = -1;
        for (int i = 0; i < nq; i++) {
            int d = orc_hamming32(t + (size_t)j * 32, q + (size_t)i * 32);
            if (d < td) { td = d; tq = i; }
        }
        if (td < best_d[tq]) { best_d[tq] = td; best_j[tq] = j; }
    }
    int n = 0;
    for (int i = 0; i < nq; i++)
        if (best_j[i] >= 0) { qidx[n] = i; tidx[n] = best_j[i]; dist[n] = best_d[i]; n++; }
    free(best_d); free(best_j);
    return n;
}

/* batched: frame b's descriptors (n[b] rows at stride max_kp) matched against frame b-1's (query = frame b-1?
 * No: the bench pairs are (query = frame b, train = frame b-1) for b >= 1, mirroring tracking.py:56
 * matcher.match(frame.des, kf_ref.des)).  Pair b=0 is empty. Outputs at stride max_kp. */
void orc_match_consecutive(const uint8_t* desc, const int* n, int B, int max_kp,
                           int* qidx, int* tidx, int* dist, int* n_match)
{
    n_match[0] = 0;
    #pragma omp parallel for schedule(dynamic)
    for (int b = 1; b < B; b++)
        n_match[b] = orc_match_crosscheck(desc + (size_t)b * max_kp * 32, n[b],
                                          desc + (size_t)(b - 1) * max_kp * 32, n[b - 1],
                                          qidx + (size_t)b * max_kp, tidx + (size_t)b * max_kp,
                                          dist + (size_t)b * max_kp);
}

/*
 * Map-point medoid descriptor — slam/nodes.py:142-154 (MapPoint.add_observation): among k <= 10 observation descriptors,
 * dists[i][j] = uint8(Hamming(f_i, f_j)) (the reference stores the distance in a uint8 matrix: 256 wraps to 0), median of
 * every column (numpy: mean of the two middle values for even k), first argmin.  Returns the index of the medoid.
 */
int orc_medoid_index(const uint8_t* desc /* [k][32] */, int k)
{
    if (k <= 0) return -1;
    int best = 0, best_m2 = 1 << 30;
    for (int c = 0; c < k; c++) {
        int col[16];
        for (int r = 0; r < k; r++) col[r] = r == c ? 0 : (orc_hamming32(desc + 32 * r, desc + 32 * c) & 255);
        for (int i = 1; i < k; i++) { int v = col[i], j = i - 1; while (j >= 0 && col[j] > v) { col[j + 1] = col[j]; j--; } col[j + 1] = v; }
        int m2 = (k & 1) ? 2 * col[k / 2] : col[k / 2 - 1] + col[k / 2];          /* twice the median */
        if (m2 < best_m2) { best_m2 = m2; best = c; }
    }
    return best;
}
