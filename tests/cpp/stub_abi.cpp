// A CPU stand-in for libeacham_hip.so, TEST INFRASTRUCTURE ONLY: it lets the header-only C++ adapters
// (include/eacham/*.hpp: request combining, upload cache, graph walks, write-back) run under ASAN / UBSAN / TSAN on a
// box without a GPU (tests/test_sanitizers.py). It computes nothing of the hot path: matches come from a fake,
// content-dependent rule (so a stale cache slot or a result delivered to the wrong caller shows), bundle adjustment
// and triangulation copy their inputs through. Never linked into anything shipped.
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "eacham_hip.h"

struct eacham_ctx {
    std::mutex mu;
    std::string err;
    struct Frame { std::vector<float> v; int n = -1, dim = 0; bool f32 = false; };
    std::map<int, Frame> frames;
    long uploads = 0, launches = 0;
};

static int fail(eacham_ctx* c, int code, const char* msg) { c->err = msg; return code; }

extern "C" {

int eacham_ctx_create(int, eacham_ctx** out) { *out = new eacham_ctx(); return EACHAM_OK; }
void eacham_ctx_destroy(eacham_ctx* c) { delete c; }
const char* eacham_last_error(const eacham_ctx* c) { return c ? c->err.c_str() : "null context"; }
int eacham_ctx_sync(eacham_ctx*) { return EACHAM_OK; }
void* eacham_ctx_stream(eacham_ctx*) { return nullptr; }
const char* eacham_version(void) { return "eacham_hip stub (CPU, tests only)"; }

static int upload(eacham_ctx* c, int id, const float* p, int n, int dim, bool f32) {
    std::lock_guard<std::mutex> lk(c->mu);
    if (n < 0 || dim <= 0 || (n > 0 && !p)) return fail(c, EACHAM_ERR_INVALID, "bad descriptor shape");
    if (!f32 && dim % 16) return fail(c, EACHAM_ERR_UNSUPPORTED, "descriptor dim: need a multiple of 16");
    for (auto& kv : c->frames)
        if (kv.first != id && kv.second.n >= 0 && kv.second.f32 != f32)
            return fail(c, EACHAM_ERR_UNSUPPORTED, "all resident frames must share one descriptor kind");
    if (!f32)
        for (size_t i = 0; i < (size_t)n * dim; ++i)
            if (!(p[i] >= 0.0f && p[i] <= 255.0f) || p[i] != std::floor(p[i])) return fail(c, EACHAM_ERR_NOT_INTEGER, "not integer");
    eacham_ctx::Frame& f = c->frames[id];
    f.v.assign(p, p + (size_t)n * dim);  // a COPY: the adapter may not rely on the caller's buffer after the upload
    f.n = n; f.dim = dim; f.f32 = f32;
    ++c->uploads;
    return EACHAM_OK;
}
int eacham_upload_descriptors(eacham_ctx* c, int id, const float* p, int n, int dim) { return upload(c, id, p, n, dim, false); }
int eacham_upload_descriptors_dev(eacham_ctx* c, int id, const float* p, int n, int dim) { return upload(c, id, p, n, dim, false); }
int eacham_upload_descriptors_f32(eacham_ctx* c, int id, const float* p, int n, int dim) { return upload(c, id, p, n, dim, true); }
int eacham_frame_rows(eacham_ctx* c, int id) {
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->frames.find(id);
    return it == c->frames.end() ? EACHAM_ERR_INVALID : it->second.n;
}
int eacham_clear_descriptors(eacham_ctx* c) { std::lock_guard<std::mutex> lk(c->mu); c->frames.clear(); return EACHAM_OK; }

// the fake rule: q -> t = (|16 A[q][dim-1]| + q) mod n2, kept iff (|16 A[q][0]| + |16 B[0][0]|) mod 3 != 0
static void fake_match(const eacham_ctx::Frame& A, const eacham_ctx::Frame& B, std::vector<uint32_t>& q, std::vector<uint32_t>& t) {
    if (A.n <= 0 || B.n < 2) return;
    const long b0 = std::lround(std::fabs(16.0f * B.v[0]));
    for (int i = 0; i < A.n; ++i) {
        const long a0 = std::lround(std::fabs(16.0f * A.v[(size_t)i * A.dim])), a1 = std::lround(std::fabs(16.0f * A.v[(size_t)i * A.dim + A.dim - 1]));
        if ((a0 + b0) % 3 == 0) continue;
        q.push_back((uint32_t)i);
        t.push_back((uint32_t)((a1 + i) % B.n));
    }
}

static int pairs_csr(eacham_ctx* c, const int32_t* pairs, int npairs, int mutual, int32_t* counts, int64_t* offsets,
                     uint32_t* oq, uint32_t* ot, int64_t cap, int64_t* total) {
    std::lock_guard<std::mutex> lk(c->mu);
    ++c->launches;
    int64_t n = 0;
    for (int p = 0; p < npairs; ++p) {
        auto a = c->frames.find(pairs[2 * p]), b = c->frames.find(pairs[2 * p + 1]);
        if (a == c->frames.end() || b == c->frames.end()) return fail(c, EACHAM_ERR_INVALID, "pair references a frame which is not resident");
        std::vector<uint32_t> q, t;
        fake_match(a->second, b->second, q, t);
        if (mutual && q.size() < 3) q.clear(), t.clear();
        offsets[p] = n;
        counts[p] = (int32_t)q.size();
        for (size_t k = 0; k < q.size(); ++k, ++n)
            if (n < cap) { oq[n] = q[k]; ot[n] = t[k]; }
    }
    offsets[npairs] = n;
    *total = n;
    return n > cap ? fail(c, EACHAM_ERR_CAPACITY, "capacity") : EACHAM_OK;
}
int eacham_match_pairs_directed(eacham_ctx* c, const int32_t* pairs, int npairs, double, int32_t* counts, int64_t* offsets,
                                uint32_t* q, uint32_t* t, int64_t cap, int64_t* total) {
    return pairs_csr(c, pairs, npairs, 0, counts, offsets, q, t, cap, total);
}
int eacham_match_all_pairs(eacham_ctx* c, const int32_t* pairs, int npairs, double, int, int, int32_t* counts, int64_t* offsets,
                           uint32_t* q, uint32_t* t, int64_t cap, int64_t* total, int32_t*) {
    return pairs_csr(c, pairs, npairs, 1, counts, offsets, q, t, cap, total);
}
int eacham_match_pair(eacham_ctx* c, int f1, int f2, double, uint32_t* q, uint32_t* t, int cap, int* count) {
    int32_t pr[2] = {f1, f2}, cnt = 0;
    int64_t off[2], total = 0;
    const int rc = pairs_csr(c, pr, 1, 0, &cnt, off, q, t, cap, &total);
    *count = (int)total;
    return rc;
}

int eacham_graph_best_pair(eacham_ctx*, int, const int32_t* pairs, int npairs, const int32_t* counts, const int64_t*, const uint32_t*,
                           const uint32_t*, const uint8_t* valid, const uint8_t* excluded, const int64_t*, const uint8_t*, uint32_t* edge_counts,
                           uint32_t* best) {
    // a fake of the query that keeps its CONTRACT: an edge from a valid node to a not-yet-valid, not-excluded one (either way round)
    best[0] = best[1] = 0xffffffffu; best[2] = 0;
    for (int p = 0; p < npairs; ++p) {
        if (edge_counts) edge_counts[2 * p] = edge_counts[2 * p + 1] = 0;
        if (counts[p] <= 0) continue;
        for (int dir = 0; dir < 2; ++dir) {
            const int a = pairs[2 * p + dir], b = pairs[2 * p + 1 - dir];
            if (valid[a] && !valid[b] && !(excluded && excluded[b]) && counts[p] >= (int32_t)best[2]) {
                best[0] = (uint32_t)a; best[1] = (uint32_t)b; best[2] = (uint32_t)counts[p];
            }
        }
    }
    return EACHAM_OK;
}

// the resident form of the query: the stub keeps the arrays and answers with the same fake rule
struct eacham_graph {
    eacham_ctx* ctx;
    int n_frames;
    std::vector<int32_t> pairs, counts;
    std::vector<int64_t> kpo;
    std::vector<uint8_t> valid;
};
int eacham_graph_create(eacham_ctx* c, int n_frames, const int32_t* pairs, int npairs, const int32_t* counts, const int64_t* offsets,
                        const uint32_t* q, const uint32_t* t, const int64_t* kp_offsets, eacham_graph** out) {
    if (!out || n_frames <= 0 || !kp_offsets) return fail(c, EACHAM_ERR_INVALID, "null");
    long long acc = 0;
    for (int p = 0; p < npairs; ++p)
        for (int64_t k = offsets[p]; k < offsets[p] + counts[p]; ++k) acc += q[k] + t[k];   // every match is read
    (void)acc;
    eacham_graph* g = new eacham_graph{c, n_frames, {pairs, pairs + 2 * (size_t)npairs}, {counts, counts + npairs},
                                       {kp_offsets, kp_offsets + n_frames + 1}, std::vector<uint8_t>(n_frames, 0)};
    *out = g;
    return EACHAM_OK;
}
void eacham_graph_destroy(eacham_graph* g) { delete g; }
int eacham_graph_set_frame(eacham_graph* g, int frame, int valid, const uint8_t* has3d, int n_keypoints) {
    if (!g || frame < 0 || frame >= g->n_frames) return EACHAM_ERR_INVALID;
    if (has3d) {
        if (n_keypoints != g->kpo[frame + 1] - g->kpo[frame]) return fail(g->ctx, EACHAM_ERR_INVALID, "keypoint count");
        int acc = 0;
        for (int k = 0; k < n_keypoints; ++k) acc += has3d[k];
        (void)acc;
    }
    g->valid[frame] = valid != 0;
    return EACHAM_OK;
}
int eacham_graph_set_frames(eacham_graph* g, int n, const int32_t* frames, const uint8_t* valid, const uint8_t* has3d, const int64_t* has3d_offsets) {
    if (!g || n < 0) return EACHAM_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        const int rc = eacham_graph_set_frame(g, frames[i], valid[i], has3d ? has3d + has3d_offsets[i] : nullptr, (int)(has3d_offsets[i + 1] - has3d_offsets[i]));
        if (rc != EACHAM_OK) return rc;
    }
    return EACHAM_OK;
}
int eacham_graph_query(eacham_graph* g, const int32_t* excluded_frames, int n_excluded, uint32_t* best) {
    if (!g || !best) return EACHAM_ERR_INVALID;
    std::vector<uint8_t> ex(g->n_frames, 0);
    for (int k = 0; k < n_excluded; ++k) ex[excluded_frames[k]] = 1;
    return eacham_graph_best_pair(g->ctx, g->n_frames, g->pairs.data(), (int)g->counts.size(), g->counts.data(), nullptr, nullptr, nullptr,
                                  g->valid.data(), ex.data(), nullptr, nullptr, nullptr, best);
}

// bundle adjustment: every array is read completely (ASAN sees a short buffer), values pass through, K moves by +1
int eacham_ba_solve(eacham_ctx* c, const eacham_ba_problem* P, const eacham_ba_options* O, eacham_ba_result* R) {
    if (!P || !O || !R) return fail(c, EACHAM_ERR_INVALID, "null");
    double acc = 0;
    for (int o = 0; o < P->n_obs; ++o) {
        if (P->obs_cam[o] >= (uint32_t)P->n_cams || P->obs_point[o] >= (uint32_t)P->n_points) return fail(c, EACHAM_ERR_INVALID, "observation out of range");
        acc += P->obs_uv[2 * o] + P->obs_uv[2 * o + 1];
    }
    int used = 0;
    std::vector<char> seen(P->n_points > 0 ? P->n_points : 1, 0);
    for (int o = 0; o < P->n_obs; ++o) if (!seen[P->obs_point[o]]) seen[P->obs_point[o]] = 1, ++used;
    for (int j = 0; j < P->n_points; ++j) acc += P->point_observers[j];
    for (int i = 0; i < P->n_cams; ++i) acc += P->cam_fixed[i];
    if (P->n_cams) std::memcpy(R->cam_T_wc, P->cam_T_wc, sizeof(double) * 16 * (size_t)P->n_cams);
    if (P->n_points) std::memcpy(R->points, P->points, sizeof(double) * 3 * (size_t)P->n_points);
    R->status = used < O->min_landmarks ? EACHAM_BA_SKIPPED : EACHAM_BA_DONE;
    for (int k = 0; k < 4; ++k) R->K[k] = P->K[k] + (R->status == EACHAM_BA_DONE ? 1.0 : 0.0);
    R->initial_error = acc; R->final_error = 0.5 * acc; R->final_lambda = 1e-4;
    R->outer_iterations = R->inner_iterations = R->status == EACHAM_BA_DONE ? 1 : 0;
    R->trace_len = 0;
    return EACHAM_OK;
}

int eacham_triangulate_tracks(eacham_ctx* c, const double* T, int n_frames, int n_tracks, const int32_t* ptr, const uint32_t* of,
                              const double* uv, const double* K, float, float, double* points, int32_t* status, uint8_t* masks) {
    double acc = K[0] + K[1] + K[2] + K[3];
    for (int i = 0; i < 16 * n_frames; ++i) acc += T[i];
    for (int t = 0; t < n_tracks; ++t) {
        const int m = ptr[t + 1] - ptr[t];
        for (int k = ptr[t]; k < ptr[t + 1]; ++k) {
            if (of[k] >= (uint32_t)n_frames) return fail(c, EACHAM_ERR_INVALID, "frame out of range");
            acc += uv[2 * k] + uv[2 * k + 1];
            masks[k] = 1;
        }
        status[t] = m >= 2 ? 3 : 0;
        points[3 * t] = points[3 * t + 1] = 0.0;
        points[3 * t + 2] = m >= 2 ? 1.0 + 1e-300 * acc : 0.0;
    }
    return EACHAM_OK;
}
int eacham_reprojection_errors(eacham_ctx* c, const double* T, int n_frames, int n, const uint32_t* frame, const double* points,
                               const double* uv, const double* K, float* err) {
    (void)T; (void)K;
    for (int i = 0; i < n; ++i) {
        if (frame[i] >= (uint32_t)n_frames) return fail(c, EACHAM_ERR_INVALID, "frame out of range");
        err[i] = (float)(1e-300 * (points[3 * i] + points[3 * i + 1] + points[3 * i + 2] + uv[2 * i] + uv[2 * i + 1]));
    }
    return EACHAM_OK;
}
int eacham_two_view_points(eacham_ctx*, int n, const double* uv1, const double* uv2, const double* K, int nt, const double* T, float, float,
                           int, double* points, uint8_t* keep, int32_t* counts) {
    double acc = K[0];
    for (int i = 0; i < 2 * n; ++i) acc += uv1[i] + uv2[i];
    for (int k = 0; k < nt; ++k) {
        for (int i = 0; i < 16; ++i) acc += T[16 * k + i];
        counts[k] = n;
        for (int i = 0; i < n; ++i) { keep[(size_t)k * n + i] = 1; points[3 * ((size_t)k * n + i)] = points[3 * ((size_t)k * n + i) + 1] = 0; points[3 * ((size_t)k * n + i) + 2] = 1 + 1e-300 * acc; }
    }
    return EACHAM_OK;
}

int eacham_score_hypotheses(eacham_ctx*, int kind, int n, const double* a, const double* b, int nm, const double* models, const double* K,
                            float, float* errors, int32_t* counts, float* medians) {
    const int sa = kind == EACHAM_SCORE_PNP ? 3 : 2, sm = kind == EACHAM_SCORE_PNP ? 12 : 9;
    double acc = K ? K[0] + K[3] : 0.0;
    for (int i = 0; i < n * sa; ++i) acc += a[i];
    for (int i = 0; i < n * 2; ++i) acc += b[i];
    for (int m = 0; m < nm; ++m) {
        for (int k = 0; k < sm; ++k) acc += models[(size_t)sm * m + k];
        if (counts) counts[m] = n;
        if (medians) medians[m] = (float)(1e-300 * acc);
        if (errors) for (int i = 0; i < n; ++i) errors[(size_t)m * n + i] = 0.f;
    }
    return EACHAM_OK;
}


int eacham_solve_minimal(eacham_ctx* c, int kind, int n_points, const double* a, const double* b, const double*, int n_samples,
                         const int32_t* idx, double* models, int32_t* n_models) {
    // one fake model per sample, built from the sample's first correspondence (content-dependent, deterministic)
    const int m = kind == 0 ? 4 : 5, maxm = kind == 0 ? 1 : 10;
    for (int s = 0; s < n_samples; ++s) {
        for (int k = 0; k < m; ++k)
            if (idx[s * m + k] < 0 || idx[s * m + k] >= n_points) return fail(c, EACHAM_ERR_INVALID, "sample index out of range");
        double* out = models + (size_t)s * maxm * 9;
        for (int k = 0; k < maxm * 9; ++k) out[k] = 0.0;
        const int i = idx[s * m];
        const double v[9] = {1, 0, a[2 * i] - b[2 * i], 0, 1, a[2 * i + 1] - b[2 * i + 1], 0, 0, 1};
        for (int k = 0; k < 9; ++k) out[k] = v[k];
        n_models[s] = 1;
    }
    return EACHAM_OK;
}

int eacham_solve_pnp(eacham_ctx* c, int n_points, const double* obj, const double*, const double*, int sample_size, int n_samples,
                     const int32_t* idx, double* models, int32_t* n_models) {
    if (sample_size < 5) return fail(c, EACHAM_ERR_INVALID, "EPnP needs at least 5 points per sample");
    for (int s = 0; s < n_samples; ++s) {   // a fake pose per sample: identity rotation, t = -first object point
        for (int k = 0; k < sample_size; ++k)
            if (idx[(size_t)s * sample_size + k] < 0 || idx[(size_t)s * sample_size + k] >= n_points) return fail(c, EACHAM_ERR_INVALID, "sample index out of range");
        const int i = idx[(size_t)s * sample_size];
        const double v[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, -obj[3 * i], -obj[3 * i + 1], -obj[3 * i + 2]};
        for (int k = 0; k < 12; ++k) models[12 * (size_t)s + k] = v[k];
        n_models[s] = 1;
    }
    return EACHAM_OK;
}
}  // extern "C"
