/*
 * bow_oracle.c — CPU restatement of (1) the local-map visibility gate of slam/tracking.py:97-111 and (2) the DBoW3 L1
 * database query behind slam/bow_db.py:37-42 -> docker/pydbow3/src/dbow3.cpp:147-152 (Database::query(des, [], 1, -1)).
 *
 * TEST INFRASTRUCTURE (see orb_oracle.c).  PARITY UNPINNED: g2o (cam_map / SE3Quat) and DBoW3 (BOpermanis/DBow3 fork,
 * unpinned HEAD, docker/install_dbow3.sh:7) are absent from /root/reference and from this image; no fixtures exist.
 *
 * (1) gate, per map point p with normal n (tracking.py:101-104):
 *       q = R p + t ; pixel = (q.x/q.z * f + cx, q.y/q.z * f + cy)       g2o CameraParameters::cam_map(SE3Quat(R,t)*p)
 *       keep iff 0 <= u < width and 0 <= v < height and dot(see_vector, n) < cos(pi/3)
 *     kept indices are emitted in input order (the Python lists of tracking.py:106-110 are appended in loop order).
 * (2) queryL1 (DBoW3 Database.cpp): for each query word (ascending id) and each database entry holding that word:
 *       acc[entry] += |qv - dv| - |qv| - |dv| ;  entries without a common word do not appear;
 *     results sorted ascending by acc, truncated to max_results, Score = -acc/2.  Ties in Score: DBoW3 leaves them to
 *     std::sort (unspecified); this restatement breaks them by ascending entry id.
 *     Database layout here: CSR, entry e owns words[start[e]..start[e+1]) sorted ascending with f64 weights.
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

int orc_visibility_gate(const double* pts, const double* normals, int M, const double* R /*row-major*/, const double* t,
                        double f, double cx, double cy, int width, int height, const double* see, double cos_thr,
                        int* idx_out)
{
    int n = 0;
    for (int i = 0; i < M; i++) {
        const double* p = pts + 3 * i;
        double qx = R[0]*p[0] + R[1]*p[1] + R[2]*p[2] + t[0];
        double qy = R[3]*p[0] + R[4]*p[1] + R[5]*p[2] + t[1];
        double qz = R[6]*p[0] + R[7]*p[1] + R[8]*p[2] + t[2];
        double u = qx / qz * f + cx, v = qy / qz * f + cy;
        if (!(0 <= u && u < width && 0 <= v && v < height)) continue;
        const double* nn = normals + 3 * i;
        double d = see[0]*nn[0] + see[1]*nn[1] + see[2]*nn[2];
        if (d < cos_thr) idx_out[n++] = i;
    }
    return n;
}

typedef struct { double acc; int id; } orc_res;
static int orc_res_cmp(const void* a, const void* b)
{
    const orc_res* x = a; const orc_res* y = b;
    if (x->acc < y->acc) return -1;
    if (x->acc > y->acc) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* returns the number of results written (<= max_results); ids/scores sorted best first.  allowed (may be NULL): u8[n_entries], the
 * fork's target_inds restriction (slam/bow_db.py:37-39): only entries with allowed[e] != 0 are considered. */
int orc_bow_query_l1_subset(const uint32_t* q_words, const double* q_vals, int nq,
                            const int64_t* start, const uint32_t* words, const double* vals, int n_entries,
                            const uint8_t* allowed, int max_results, int* ids, double* scores);
int orc_bow_query_l1(const uint32_t* q_words, const double* q_vals, int nq,
                     const int64_t* start, const uint32_t* words, const double* vals, int n_entries,
                     int max_results, int* ids, double* scores)
{
    return orc_bow_query_l1_subset(q_words, q_vals, nq, start, words, vals, n_entries, 0, max_results, ids, scores);
}

/* the fork's compareBowsL1(i, j, cnt, score) behind Dbow.compareFrames (slam/bow_db.py:47-57 -> dbow3.cpp:154-162) [fork, unpinned]:
 * number of common words and -sum(|a - b| - |a| - |b|) / 2 over them, summed in ascending word order */
void orc_bow_compare_l1(const int64_t* start, const uint32_t* words, const double* vals, int ei, int ej, int* n_common, double* score)
{
    int64_t a = start[ei], a1 = start[ei + 1], b = start[ej], b1 = start[ej + 1];
    double acc = 0; int cnt = 0;
    while (a < a1 && b < b1) {
        if (words[a] < words[b]) a++;
        else if (words[a] > words[b]) b++;
        else { acc += fabs(vals[a] - vals[b]) - fabs(vals[a]) - fabs(vals[b]); cnt++; a++; b++; }
    }
    *n_common = cnt; *score = -acc / 2.0;
}

int orc_bow_query_l1_subset(const uint32_t* q_words, const double* q_vals, int nq,
                            const int64_t* start, const uint32_t* words, const double* vals, int n_entries,
                            const uint8_t* allowed, int max_results, int* ids, double* scores)
{
    orc_res* r = malloc(sizeof(orc_res) * (size_t)(n_entries > 0 ? n_entries : 1));
    int nr = 0;
    for (int e = 0; e < n_entries; e++) {
        if (allowed && !allowed[e]) continue;
        /* merge-join: both lists ascending; accumulation order = ascending word id, as in DBoW3's outer loop */
        int64_t a = start[e], b = start[e + 1];
        int i = 0; int hit = 0; double acc = 0;
        while (a < b && i < nq) {
            if (words[a] < q_words[i]) a++;
            else if (words[a] > q_words[i]) i++;
            else { double qv = q_vals[i], dv = vals[a]; acc += fabs(qv - dv) - fabs(qv) - fabs(dv); hit = 1; a++; i++; }
        }
        if (hit) { r[nr].acc = acc; r[nr].id = e; nr++; }
    }
    qsort(r, nr, sizeof(orc_res), orc_res_cmp);
    if (max_results > 0 && nr > max_results) nr = max_results;
    for (int k = 0; k < nr; k++) { ids[k] = r[k].id; scores[k] = -r[k].acc / 2.0; }
    free(r);
    return nr;
}

/*
 * (3) Vocabulary::transform (DBoW3 Vocabulary.cpp; reached from slam/bow_db.py:26 Database.add(frame.des) and :39 query):
 *     every descriptor walks the tree from the root, at each node taking the child with the smallest Hamming distance
 *     (children scanned in their stored order, strict '<': first minimum) until a leaf; the leaf's (word id, weight) is
 *     accumulated v[word] += weight in descriptor order (weights <= 0 are stop words); TF-IDF + L1: v /= sum(|v|) with the
 *     sum taken in ascending word order.  Tree layout here: children of node i are child_idx[child_start[i] ..
 *     child_start[i+1]); leaves have no children.
 */
#include <string.h>
static int orc_ham32(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int k = 0; k < 32; k += 8) { uint64_t x, y; memcpy(&x, a + k, 8); memcpy(&y, b + k, 8); d += __builtin_popcountll(x ^ y); }
    return d;
}

typedef struct { uint32_t w; double v; int first; } orc_wv;
static int orc_wv_cmp(const void* a, const void* b)
{
    const orc_wv* x = a; const orc_wv* y = b;
    if (x->w != y->w) return x->w < y->w ? -1 : 1;
    return (x->first > y->first) - (x->first < y->first);
}

/* desc u8[n][32] -> sparse BoW vector (words ascending); returns the number of distinct words */
int orc_voc_transform(const int* child_start, const int* child_idx, const uint8_t* node_desc, const double* node_weight,
                      const int* node_word, const uint8_t* desc, int n, uint32_t* words, double* vals)
{
    orc_wv* tmp = malloc(sizeof(orc_wv) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int r = 0; r < n; r++) {
        int node = 0;
        while (child_start[node + 1] > child_start[node]) {
            int best = -1, best_d = 1 << 30;
            for (int c = child_start[node]; c < child_start[node + 1]; c++) {
                int id = child_idx[c];
                int d = orc_ham32(desc + (size_t)r * 32, node_desc + (size_t)id * 32);
                if (d < best_d) { best_d = d; best = id; }
            }
            node = best;
        }
        if (node_weight[node] > 0) { tmp[m].w = (uint32_t)node_word[node]; tmp[m].v = node_weight[node]; tmp[m].first = r; m++; }
    }
    qsort(tmp, m, sizeof(orc_wv), orc_wv_cmp);              /* by word, then by descriptor order: the order addWeight sums in */
    int k = 0;
    for (int i = 0; i < m; ) {
        double acc = 0; int j = i;
        while (j < m && tmp[j].w == tmp[i].w) { acc += tmp[j].v; j++; }
        words[k] = tmp[i].w; vals[k] = acc; k++; i = j;
    }
    double norm = 0;
    for (int i = 0; i < k; i++) norm += fabs(vals[i]);
    if (norm > 0) for (int i = 0; i < k; i++) vals[i] /= norm;
    free(tmp);
    return k;
}
