/*
 * match_oracle.c — CPU restatement of eacham's descriptor-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (eacham_amd/, include/) may call this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * PARITY UNPINNED: the reference has no tests/golden vectors for this path and
 * its arithmetic lives in OpenCV 4.5.5 (conanfile.txt:3), which is absent here and cannot be
 * built. This file restates the published semantics:
 *
 *   FeatureMatcherFlann::Match      /root/reference/modules/base/features/FeatureMatcherFlann.cpp:14-30
 *     knnMatch(d1, d2, matches, 2)  (:17)   -> exact 2-NN under L2 (what FLANN approximates;
 *                                              cv::BFMatcher(NORM_L2) semantics, SURVEY.md App. B:
 *                                              distance = sqrtf(sum (a-b)^2) in fp32, candidates
 *                                              scanned in ascending train index, strict '<' =>
 *                                              ties keep the lower train index first)
 *     m[0].distance / m[1].distance < 0.8   (:23)  fp32 quotient promoted to double
 *     matchesPair.insert({queryIdx, trainIdx}) (:25)
 *   pair loop + mutual check        /root/reference/apps/sfm/main.cpp:84-147
 *     |m12| < 30 -> drop (:111); mutual: m21[m2] == m1 (:133-140); |mutual| > 30 -> edge (:142-146)
 *
 * Output lists are sorted by query index (the reference's unordered_map has no order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    float d0, d1;   /* sqrtf of the two smallest squared distances (d0 <= d1) */
    int32_t i0;     /* index of the nearest; -1 if none */
    int32_t have;   /* number of neighbours found (0, 1, 2) */
} top2_t;

static inline void top2_init(top2_t* s) {
    s->d0 = s->d1 = INFINITY;
    s->i0 = -1;
    s->have = 0;
}

/* K-best insertion with strict '<' while scanning candidates in ascending index. */
static inline void top2_push(top2_t* s, float d, int32_t idx) {
    if (d < s->d0) {
        s->d1 = s->d0;
        s->d0 = d;
        s->i0 = idx;
    } else if (d < s->d1) {
        s->d1 = d;
    }
    if (s->have < 2) s->have++;
}

/* FeatureMatcherFlann.cpp:23 — `m[0].distance / m[1].distance < 0.8` */
static inline int ratio_pass(const top2_t* s, double ratio) {
    if (s->have < 2) return 0; /* the reference would index m[1] out of bounds; we emit nothing */
    float q = s->d0 / s->d1;   /* fp32 division; 0/0 = NaN -> comparison false */
    return (double)q < ratio;
}

/* squared L2 in fp32, k ascending (plain restatement of normL2Sqr; exact for integer data) */
static inline float ssd_f32(const float* a, const float* b, int dim) {
    float s = 0.0f;
    for (int k = 0; k < dim; ++k) {
        float d = a[k] - b[k];
        s += d * d;
    }
    return s;
}

/* "L2 via dot product" in fp32, the form the device's float path evaluates on the f32 MFMA:
 *   dot = fma chain over k ascending, |x|^2 likewise, d2 = max(fma(-2, dot, |a|^2 + |b|^2), 0).
 * (v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain, MI355X guide §3.) This is NOT the
 * rounding of sum (a-b)^2 — near-duplicate descriptors cancel differently — so the float path's
 * parity is stated against THIS restatement, and its agreement with ssd_f32 is reported by the tests. */
static inline float sqnorm_f32(const float* a, int dim) {
    float s = 0.0f;
    for (int k = 0; k < dim; ++k) s = fmaf(a[k], a[k], s);
    return s;
}
static inline float dist2_dot_f32(const float* a, const float* b, int dim, float na, float nb) {
    float dot = 0.0f;
    for (int k = 0; k < dim; ++k) dot = fmaf(a[k], b[k], dot);
    float d2 = fmaf(-2.0f, dot, na + nb);
    return d2 > 0.0f ? d2 : 0.0f;
}

/* exact integer squared L2 for u8 data (vectorises; equals ssd_f32 bit-for-bit on such data
 * because every partial sum is an integer < 2^24) */
static inline int32_t ssd_u8(const uint8_t* a, const uint8_t* b, int dim) {
    int32_t s = 0;
    for (int k = 0; k < dim; ++k) {
        int32_t d = (int32_t)a[k] - (int32_t)b[k];
        s += d * d;
    }
    return s;
}

/* returns 1 and fills out (n*dim bytes) if all values are integers in [0,255] */
static int to_u8(const float* x, int64_t count, uint8_t* out) {
    for (int64_t i = 0; i < count; ++i) {
        float v = x[i];
        if (!(v >= 0.0f && v <= 255.0f) || v != floorf(v)) return 0;
        out[i] = (uint8_t)v;
    }
    return 1;
}

/*
 * Both directions of one unordered pair from a single pass over the distance matrix.
 * fwd[q] (n1 entries) / bwd[t] (n2 entries): top-2 state of each row of A against B / each row of
 * B against A. The q-outer / t-inner loop visits train candidates in ascending order for both.
 * force_f32 = 1 disables the (bit-identical) integer fast path (the tests prove the two agree);
 * force_f32 = 2 selects the fp32 dot-product form of the device's float path.
 */
static void top2_both(const float* A, int n1, const float* B, int n2, int dim, int force_f32,
                      top2_t* fwd, top2_t* bwd) {
    for (int q = 0; q < n1; ++q) top2_init(&fwd[q]);
    for (int t = 0; t < n2; ++t) top2_init(&bwd[t]);
    uint8_t *a8 = NULL, *b8 = NULL;
    int use_u8 = 0;
    if (!force_f32 && n1 > 0 && n2 > 0) {
        a8 = (uint8_t*)malloc((size_t)n1 * dim);
        b8 = (uint8_t*)malloc((size_t)n2 * dim);
        use_u8 = a8 && b8 && to_u8(A, (int64_t)n1 * dim, a8) && to_u8(B, (int64_t)n2 * dim, b8);
    }
    float *na = NULL, *nb = NULL;
    if (force_f32 == 2) {
        na = (float*)malloc(sizeof(float) * (size_t)(n1 > 0 ? n1 : 1));
        nb = (float*)malloc(sizeof(float) * (size_t)(n2 > 0 ? n2 : 1));
        for (int q = 0; q < n1; ++q) na[q] = sqnorm_f32(A + (size_t)q * dim, dim);
        for (int t = 0; t < n2; ++t) nb[t] = sqnorm_f32(B + (size_t)t * dim, dim);
    }
    for (int q = 0; q < n1; ++q) {
        for (int t = 0; t < n2; ++t) {
            float d2 = force_f32 == 2 ? dist2_dot_f32(A + (size_t)q * dim, B + (size_t)t * dim, dim, na[q], nb[t])
                       : use_u8     ? (float)ssd_u8(a8 + (size_t)q * dim, b8 + (size_t)t * dim, dim)
                                    : ssd_f32(A + (size_t)q * dim, B + (size_t)t * dim, dim);
            float d = sqrtf(d2);
            top2_push(&fwd[q], d, t);
            top2_push(&bwd[t], d, q);
        }
    }
    free(a8);
    free(b8);
    free(na);
    free(nb);
}

/* Directed Match(A, B): returns the number of (q, t) written, sorted by q. */
int oracle_match_directed(const float* A, int n1, const float* B, int n2, int dim, double ratio,
                          int force_f32, uint32_t* out_q, uint32_t* out_t) {
    top2_t* fwd = (top2_t*)malloc(sizeof(top2_t) * (size_t)(n1 > 0 ? n1 : 1));
    top2_t* bwd = (top2_t*)malloc(sizeof(top2_t) * (size_t)(n2 > 0 ? n2 : 1));
    top2_both(A, n1, B, n2, dim, force_f32, fwd, bwd);
    int cnt = 0;
    for (int q = 0; q < n1; ++q) {
        if (ratio_pass(&fwd[q], ratio)) {
            out_q[cnt] = (uint32_t)q;
            out_t[cnt] = (uint32_t)fwd[q].i0;
            ++cnt;
        }
    }
    free(fwd);
    free(bwd);
    return cnt;
}

/* Raw 2-NN of every row of A against B (for tests that inspect distances):
 * idx0[q], d0[q], d1[q] (sqrtf distances; INFINITY / -1 when absent). */
void oracle_knn2(const float* A, int n1, const float* B, int n2, int dim, int force_f32,
                 int32_t* idx0, float* d0, float* d1) {
    top2_t* fwd = (top2_t*)malloc(sizeof(top2_t) * (size_t)(n1 > 0 ? n1 : 1));
    top2_t* bwd = (top2_t*)malloc(sizeof(top2_t) * (size_t)(n2 > 0 ? n2 : 1));
    top2_both(A, n1, B, n2, dim, force_f32, fwd, bwd);
    for (int q = 0; q < n1; ++q) {
        idx0[q] = fwd[q].i0;
        d0[q] = fwd[q].d0;
        d1[q] = fwd[q].d1;
    }
    free(fwd);
    free(bwd);
}

/*
 * One unordered pair, apps/sfm/main.cpp:98-147 semantics. Writes the mutual matches sorted by q
 * into out_q/out_t (capacity n1) and returns |mutual| if the pair becomes an edge, else 0.
 * stats4 (optional): {|m12|, |m21|, |mutual|, edge}.
 */
int oracle_match_mutual(const float* A, int n1, const float* B, int n2, int dim, double ratio,
                        int min_dir, int min_mutual, int force_f32,
                        uint32_t* out_q, uint32_t* out_t, int32_t* stats4) {
    top2_t* fwd = (top2_t*)malloc(sizeof(top2_t) * (size_t)(n1 > 0 ? n1 : 1));
    top2_t* bwd = (top2_t*)malloc(sizeof(top2_t) * (size_t)(n2 > 0 ? n2 : 1));
    int32_t* m12 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1));
    int32_t* m21 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));
    top2_both(A, n1, B, n2, dim, force_f32, fwd, bwd);
    int c12 = 0, c21 = 0;
    for (int q = 0; q < n1; ++q) {
        m12[q] = ratio_pass(&fwd[q], ratio) ? fwd[q].i0 : -1;
        c12 += m12[q] >= 0;
    }
    for (int t = 0; t < n2; ++t) {
        m21[t] = ratio_pass(&bwd[t], ratio) ? bwd[t].i0 : -1;
        c21 += m21[t] >= 0;
    }
    int cm = 0;
    for (int q = 0; q < n1; ++q) { /* main.cpp:133-140 */
        int32_t t = m12[q];
        if (t >= 0 && m21[t] == q) {
            out_q[cm] = (uint32_t)q;
            out_t[cm] = (uint32_t)t;
            ++cm;
        }
    }
    /* main.cpp:111 (each direction `< 30` drops the pair) and :142 (`> 30` mutual connects) */
    int edge = (c12 >= min_dir) && (c21 >= min_dir) && (cm > min_mutual);
    if (stats4) {
        stats4[0] = c12;
        stats4[1] = c21;
        stats4[2] = cm;
        stats4[3] = edge;
    }
    free(fwd);
    free(bwd);
    free(m12);
    free(m21);
    return edge ? cm : 0;
}

/*
 * All pairs, threaded over pairs like the reference's for_each(par_unseq, pairs)
 * (apps/sfm/main.cpp:98). desc[f] -> n[f] x dim row-major fp32.
 * matches: npairs x stride x {q, t} (uint32 pairs); counts[p] as oracle_match_mutual.
 * Returns the number of threads used.
 */
int oracle_match_all_pairs(const float* const* desc, const int32_t* n, int dim,
                           const int32_t* pairs, int npairs, double ratio, int min_dir,
                           int min_mutual, int nthreads, int32_t* counts, uint32_t* matches,
                           int stride, int32_t* stats, int force_f32) {
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int p = 0; p < npairs; ++p) {
        int f1 = pairs[2 * p], f2 = pairs[2 * p + 1];
        int cap = n[f1] > 0 ? n[f1] : 1;
        uint32_t* q = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap);
        uint32_t* t = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap);
        int c = oracle_match_mutual(desc[f1], n[f1], desc[f2], n[f2], dim, ratio, min_dir,
                                    min_mutual, force_f32, q, t, stats ? stats + 4 * p : NULL);
        counts[p] = c;
        for (int k = 0; k < c && k < stride; ++k) {
            matches[((size_t)p * stride + k) * 2 + 0] = q[k];
            matches[((size_t)p * stride + k) * 2 + 1] = t[k];
        }
        free(q);
        free(t);
    }
    return used;
}
