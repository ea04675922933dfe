// triangulate.hip — per-track two-view RANSAC triangulation on gfx950.
//
// Replaces the per-point loop of TriangulateFrame -> TriangulatePointRansac
// (/root/reference/modules/sfm/reconstruction/Triangulator.cpp:96-186, :212-283) for a batch of
// tracks: every (track, observation pair) is one thread of K1, which triangulates the pair by DLT
// (the null vector of the 4x4 of :49-63, a one-sided Jacobi SVD held entirely in registers),
// applies the triangulation-angle gate (:21-47) and scores all observations of the track
// (reprojection error rounded to float as CalcReprojectionError does, ProjectionHelper.cpp:32-38,
// and IsPositiveDepth, :90-94). K2 then replays the reference's sequential selection per track:
// first pair with the strictly largest inlier count keeps its mask, the returned point is the one
// of the LAST pair tried, the ransac verdict is "that point has world z > 0 and more than two inliers
// were found", and TriangulateFrame adds the point iff additionally every observation is an inlier
// (:270-275). Both quirks are deliberate.
//
// fp64 vector-ALU work (~3.5 kFLOP per pair), no reuse worth staging in LDS: transforms of the
// window (<= a few hundred 128-byte matrices) live in L2/L1. The launch is flat over pairs so short
// tracks do not idle lanes; a binary search over the pair CSR finds the owning track.
#include "context.hpp"

namespace eacham {

namespace {

constexpr int TRI_SWEEPS = 12;
constexpr int TRI_BLOCK = 256;

__device__ __forceinline__ void null_vector_4x4(double (&A)[4][4], double (&x)[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < TRI_SWEEPS; ++sweep) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double al = 0, be = 0, ga = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    al += A[i][p] * A[i][p];
                    be += A[i][q] * A[i][q];
                    ga += A[i][p] * A[i][q];
                }
                const bool rot = !(fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be));
                const double gs = rot ? ga : 1.0;
                const double zeta = (be - al) / (2.0 * gs);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                if (!rot) { c = 1.0; s = 0.0; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
    }
    double nn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        nn[j] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) nn[j] += A[i][j] * A[i][j];
    }
    int best = 0;
    double bn = nn[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (nn[j] < bn) { bn = nn[j]; best = j; }
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = best == 0 ? V[i][0] : best == 1 ? V[i][1] : best == 2 ? V[i][2] : V[i][3];
}

__device__ __forceinline__ void cam_center(const double* __restrict__ T, double (&c)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) c[i] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
}

__device__ __forceinline__ double tri_angle(const double* __restrict__ T1, const double* __restrict__ T2, const double (&X)[3]) {
    double c1[3], c2[3], r1[3], r2[3];
    cam_center(T1, c1);
    cam_center(T2, c2);
#pragma unroll
    for (int i = 0; i < 3; ++i) { r1[i] = X[i] - c1[i]; r2[i] = X[i] - c2[i]; }
    const double n1 = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
    const double n2 = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
    if (fabs(n1) < (double)0.0000001f || fabs(n2) < (double)0.0000001f) return 0.0;
    const double dot = r1[0] * r2[0] + r1[1] * r2[1] + r1[2] * r2[2];
    const double ang = acos(dot / (n1 * n2));
    const double PI = 3.14159265358979323846;
    return ang < PI - ang ? ang : PI - ang;
}

__device__ __forceinline__ bool is_inlier(const double* __restrict__ T, double u0, double v0, const double (&K)[4],
                                          const double (&X)[3], float max_err) {
    const double px = T[0] * X[0] + T[1] * X[1] + T[2] * X[2] + T[3];
    const double py = T[4] * X[0] + T[5] * X[1] + T[6] * X[2] + T[7];
    const double pz = T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11];
    const double u = (K[0] * px) / pz + K[2], v = (K[1] * py) / pz + K[3];
    const float err = (float)sqrt((u0 - u) * (u0 - u) + (v0 - v) * (v0 - v));
    return err < max_err && pz >= 2.220446049250313e-16;
}

// the triangulation of observations r1 < r2 of a track (TriangulatePoint, Triangulator.cpp:49-63) and its angle gate
__device__ __forceinline__ bool tri_pair_point(const double* __restrict__ transforms, const unsigned* __restrict__ obs_frame,
                                               const double2* __restrict__ obs_uv, int o0, int r1, int r2, const double (&K)[4],
                                               float min_angle, double (&X)[3]) {
    const double* T1 = transforms + 16 * (size_t)obs_frame[o0 + r1];
    const double* T2 = transforms + 16 * (size_t)obs_frame[o0 + r2];
    const double2 p1 = obs_uv[o0 + r1], p2 = obs_uv[o0 + r2];
    const double x1 = (p1.x - K[2]) / K[0], y1 = (p1.y - K[3]) / K[1];
    const double x2 = (p2.x - K[2]) / K[0], y2 = (p2.y - K[3]) / K[1];
    double A[4][4], x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[1][j] = x1 * T1[8 + j] - T1[j];
        A[0][j] = y1 * T1[8 + j] - T1[4 + j];
        A[3][j] = x2 * T2[8 + j] - T2[j];
        A[2][j] = y2 * T2[8 + j] - T2[4 + j];
    }
    null_vector_4x4(A, x);
    X[0] = x[0] / x[3], X[1] = x[1] / x[3], X[2] = x[2] / x[3];
    return tri_angle(T1, T2, X) >= (double)min_angle;
}

// pair index of a track with m observations -> (r1 < r2), pairs in the reference's loop order
__device__ __forceinline__ void tri_pair_rows(int m, long long idx, int& r1, int& r2) {
    r1 = 0;
    while (idx >= m - 1 - r1) { idx -= m - 1 - r1; ++r1; }
    r2 = r1 + 1 + (int)idx;
}

// K1: one thread per (track, pair): the pair's triangulation, its inlier count over ALL observations of the track
// (-1: the angle gate failed). No per-pair mask is kept — a 64-bit word per pair used to cap a track at 64
// observations; the reference has no such limit (a 500-frame sequence can exceed it) — the selection kernel
// re-derives the mask of the ONE winning pair.
__global__ __launch_bounds__(TRI_BLOCK) void tri_pairs_kernel(
    const double* __restrict__ transforms, int n_tracks, const int* __restrict__ track_ptr,
    const long long* __restrict__ pair_ptr, const unsigned* __restrict__ obs_frame, const double2* __restrict__ obs_uv,
    const double* __restrict__ Kdev, float max_err, float min_angle, double* __restrict__ points,
    int* __restrict__ pair_inl) {
    const long long pid = (long long)blockIdx.x * TRI_BLOCK + threadIdx.x;
    const long long total = pair_ptr[n_tracks];
    if (pid >= total) return;
    int lo = 0, hi = n_tracks;  // largest t with pair_ptr[t] <= pid
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pair_ptr[mid] <= pid) lo = mid; else hi = mid;
    }
    const int t = lo;
    const int o0 = track_ptr[t], m = track_ptr[t + 1] - o0;
    int r1, r2;
    tri_pair_rows(m, pid - pair_ptr[t], r1, r2);
    const bool last = pid + 1 == pair_ptr[t + 1];
    const double K[4] = {Kdev[0], Kdev[1], Kdev[2], Kdev[3]};
    double X[3];
    const bool wide = tri_pair_point(transforms, obs_frame, obs_uv, o0, r1, r2, K, min_angle, X);
    if (last) {
        points[3 * (size_t)t + 0] = X[0];
        points[3 * (size_t)t + 1] = X[1];
        points[3 * (size_t)t + 2] = X[2];
    }
    int inl = -1;
    if (wide) {
        inl = 0;
        for (int i = 0; i < m; ++i) {
            const double2 p = obs_uv[o0 + i];
            if (is_inlier(transforms + 16 * (size_t)obs_frame[o0 + i], p.x, p.y, K, X, max_err)) ++inl;
        }
    }
    pair_inl[pid] = inl;
}

// K2: one thread per track: sequential selection exactly as the reference's loop leaves it; the inlier mask of the
// winning pair is evaluated again from that pair's triangulation (the same instructions on the same inputs as K1).
__global__ __launch_bounds__(TRI_BLOCK) void tri_select_kernel(
    const double* __restrict__ transforms, int n_tracks, const int* __restrict__ track_ptr, const long long* __restrict__ pair_ptr,
    const unsigned* __restrict__ obs_frame, const double2* __restrict__ obs_uv, const double* __restrict__ Kdev, float max_err,
    float min_angle, const int* __restrict__ pair_inl, double* __restrict__ points, int* __restrict__ accept,
    unsigned char* __restrict__ masks) {
    const int t = blockIdx.x * TRI_BLOCK + threadIdx.x;
    if (t >= n_tracks) return;
    const int o0 = track_ptr[t], m = track_ptr[t + 1] - o0;
    const long long p0 = pair_ptr[t], p1 = pair_ptr[t + 1];
    long long winner = -1;          // the pair whose mask the reference keeps
    int best = 0;
    bool ok = false, full = false;  // ok = TriangulatePointRansac's return value
    if (m < 2) {
        points[3 * (size_t)t] = points[3 * (size_t)t + 1] = points[3 * (size_t)t + 2] = 0.0;
    } else if (m == 2) {
        best = pair_inl[p0];
        if (best >= 0) {
            winner = p0;
            ok = points[3 * (size_t)t + 2] > 0.0;
            full = best == 2;
        }
    } else {
        for (long long p = p0; p < p1; ++p) {
            const int inl = pair_inl[p];
            if (inl > best) { best = inl; winner = p; }
        }
        ok = points[3 * (size_t)t + 2] > 0.0 && best > 2;
        full = best == m;
    }
    accept[t] = (ok ? 1 : 0) | (full ? 2 : 0);
    if (winner < 0) {
        for (int i = 0; i < m; ++i) masks[o0 + i] = 0;
        return;
    }
    const double K[4] = {Kdev[0], Kdev[1], Kdev[2], Kdev[3]};
    int r1, r2;
    tri_pair_rows(m, winner - p0, r1, r2);
    double X[3];
    (void)tri_pair_point(transforms, obs_frame, obs_uv, o0, r1, r2, K, min_angle, X);
    for (int i = 0; i < m; ++i) {
        const double2 p = obs_uv[o0 + i];
        masks[o0 + i] = is_inlier(transforms + 16 * (size_t)obs_frame[o0 + i], p.x, p.y, K, X, max_err) ? 1 : 0;
    }
}

// CalcReprojectionError for existing map points seen again (Triangulator.cpp:222-236).
__global__ __launch_bounds__(TRI_BLOCK) void reproject_kernel(const double* __restrict__ transforms, int n,
                                                              const unsigned* __restrict__ frame,
                                                              const double* __restrict__ points,
                                                              const double2* __restrict__ uv,
                                                              const double* __restrict__ Kdev, float* __restrict__ err) {
    const int i = blockIdx.x * TRI_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double* T = transforms + 16 * (size_t)frame[i];
    const double X0 = points[3 * (size_t)i], X1 = points[3 * (size_t)i + 1], X2 = points[3 * (size_t)i + 2];
    const double px = T[0] * X0 + T[1] * X1 + T[2] * X2 + T[3];
    const double py = T[4] * X0 + T[5] * X1 + T[6] * X2 + T[7];
    const double pz = T[8] * X0 + T[9] * X1 + T[10] * X2 + T[11];
    const double u = (Kdev[0] * px) / pz + Kdev[2], v = (Kdev[1] * py) / pz + Kdev[3];
    const double2 p = uv[i];
    err[i] = (float)sqrt((p.x - u) * (p.x - u) + (p.y - v) * (p.y - v));
}

// Two-view structure for candidate relative poses (ReconstructionManager.cpp:118-143, :162-186):
// thread = (transform, match); camera 1 is the identity.
__global__ __launch_bounds__(TRI_BLOCK) void two_view_kernel(int n, const double2* __restrict__ uv1,
                                                             const double2* __restrict__ uv2, const double* __restrict__ Kdev,
                                                             int nt, const double* __restrict__ transforms, float max_err,
                                                             float min_angle, int angle_strict, double* __restrict__ points,
                                                             unsigned char* __restrict__ keep) {
    const long long id = (long long)blockIdx.x * TRI_BLOCK + threadIdx.x;
    if (id >= (long long)n * nt) return;
    const int k = (int)(id / n), i = (int)(id % n);
    const double K[4] = {Kdev[0], Kdev[1], Kdev[2], Kdev[3]};
    const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const double* T = transforms + 16 * (size_t)k;
    const double2 p1 = uv1[i], p2 = uv2[i];
    const double x1 = (p1.x - K[2]) / K[0], y1 = (p1.y - K[3]) / K[1];
    const double x2 = (p2.x - K[2]) / K[0], y2 = (p2.y - K[3]) / K[1];
    double A[4][4], x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[1][j] = x1 * I4[8 + j] - I4[j];
        A[0][j] = y1 * I4[8 + j] - I4[4 + j];
        A[3][j] = x2 * T[8 + j] - T[j];
        A[2][j] = y2 * T[8 + j] - T[4 + j];
    }
    null_vector_4x4(A, x);
    const double X[3] = {x[0] / x[3], x[1] / x[3], x[2] / x[3]};
    double* out = points + 3 * (size_t)id;
    out[0] = X[0];
    out[1] = X[1];
    out[2] = X[2];
    bool ok = false;
    if (!(X[2] <= 0.0)) {
        const double u = (K[0] * X[0]) / X[2] + K[2], v = (K[1] * X[1]) / X[2] + K[3];
        const float err = (float)sqrt((p1.x - u) * (p1.x - u) + (p1.y - v) * (p1.y - v));
        const double ang = tri_angle(I4, T, X);
        ok = err < max_err && (angle_strict ? ang > (double)min_angle : !(ang < (double)min_angle));
    }
    keep[id] = ok ? 1 : 0;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

}  // namespace eacham

using namespace eacham;

extern "C" int eacham_triangulate_tracks(eacham_ctx* ctx, const double* transforms, int n_frames, int n_tracks,
                                         const int32_t* track_ptr, const uint32_t* obs_frame, const double* obs_uv,
                                         const double* K, float max_repr_error, float min_tri_angle, double* points,
                                         int32_t* status, uint8_t* masks) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_tracks < 0 || n_frames < 0 || !track_ptr || !K || !points || !status)
        return ctx->fail(EACHAM_ERR_INVALID, "triangulate: null argument or negative size");
    if (n_tracks == 0) return EACHAM_OK;
    if (track_ptr[0] != 0) return ctx->fail(EACHAM_ERR_INVALID, "triangulate: track_ptr[0] must be 0");
    std::vector<long long> pair_ptr((size_t)n_tracks + 1);
    pair_ptr[0] = 0;
    for (int t = 0; t < n_tracks; ++t) {
        const long long m = (long long)track_ptr[t + 1] - track_ptr[t];
        if (m < 0) return ctx->fail(EACHAM_ERR_INVALID, "triangulate: track_ptr not monotone at track %d", t);
        pair_ptr[t + 1] = pair_ptr[t] + (m < 2 ? 0 : m == 2 ? 1 : m * (m - 1) / 2);
    }
    const long long n_obs = track_ptr[n_tracks], n_pairs = pair_ptr[n_tracks];
    if (n_obs > 0 && (!transforms || !obs_frame || !obs_uv || !masks))
        return ctx->fail(EACHAM_ERR_INVALID, "triangulate: null observation arrays");
    for (long long i = 0; i < n_obs; ++i)
        if (obs_frame[i] >= (uint32_t)n_frames)
            return ctx->fail(EACHAM_ERR_INVALID, "triangulate: observation %lld names frame %u of %d", i, obs_frame[i], n_frames);
    if (n_pairs > (1ll << 31) * TRI_BLOCK) return ctx->fail(EACHAM_ERR_CAPACITY, "triangulate: too many pairs");

    EACHAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t o_T = take(sizeof(double) * 16 * (size_t)n_frames), o_K = take(sizeof(double) * 4);
    const size_t o_tp = take(sizeof(int) * ((size_t)n_tracks + 1)), o_pp = take(sizeof(long long) * ((size_t)n_tracks + 1));
    const size_t o_of = take(sizeof(unsigned) * (size_t)n_obs), o_uv = take(sizeof(double) * 2 * (size_t)n_obs);
    const size_t o_pt = take(sizeof(double) * 3 * (size_t)n_tracks), o_ac = take(sizeof(int) * (size_t)n_tracks);
    const size_t o_mk = take((size_t)n_obs), o_pi = take(sizeof(int) * (size_t)n_pairs);
    if (int rc = ensure_io(ctx, off)) return rc;
    if (int rc = ensure_io_host(ctx, o_pi)) return rc;   // everything but the per-pair scratch
    char* base = (char*)ctx->io;
    hipStream_t st = ctx->stream;
    IoPack io(ctx, st);
    if (n_obs > 0)
        if (int rc = io.in(o_T, transforms, sizeof(double) * 16 * (size_t)n_frames)) return rc;
    if (int rc = io.in(o_K, K, sizeof(double) * 4)) return rc;
    if (int rc = io.in(o_tp, track_ptr, sizeof(int) * ((size_t)n_tracks + 1))) return rc;
    if (int rc = io.in(o_pp, pair_ptr.data(), sizeof(long long) * ((size_t)n_tracks + 1))) return rc;
    if (n_obs > 0) {
        if (int rc = io.in(o_of, obs_frame, sizeof(unsigned) * (size_t)n_obs)) return rc;
        if (int rc = io.in(o_uv, obs_uv, sizeof(double) * 2 * (size_t)n_obs)) return rc;
    }
    if (int rc = io.flush_in()) return rc;
    {
        ProfileScope scope(ctx, EACHAM_KERNEL_TRIANGULATE);
        if (n_pairs > 0)
            tri_pairs_kernel<<<(unsigned)((n_pairs + TRI_BLOCK - 1) / TRI_BLOCK), TRI_BLOCK, 0, st>>>(
                (const double*)(base + o_T), n_tracks, (const int*)(base + o_tp), (const long long*)(base + o_pp),
                (const unsigned*)(base + o_of), (const double2*)(base + o_uv), (const double*)(base + o_K), max_repr_error,
                min_tri_angle, (double*)(base + o_pt), (int*)(base + o_pi));
        tri_select_kernel<<<(unsigned)((n_tracks + TRI_BLOCK - 1) / TRI_BLOCK), TRI_BLOCK, 0, st>>>(
            (const double*)(base + o_T), n_tracks, (const int*)(base + o_tp), (const long long*)(base + o_pp),
            (const unsigned*)(base + o_of), (const double2*)(base + o_uv), (const double*)(base + o_K), max_repr_error,
            min_tri_angle, (const int*)(base + o_pi), (double*)(base + o_pt), (int*)(base + o_ac), (unsigned char*)(base + o_mk));
    }
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    if (int rc = io.out(points, o_pt, sizeof(double) * 3 * (size_t)n_tracks)) return rc;
    if (int rc = io.out(status, o_ac, sizeof(int) * (size_t)n_tracks)) return rc;
    if (n_obs > 0)
        if (int rc = io.out(masks, o_mk, (size_t)n_obs)) return rc;
    if (int rc = io.finish()) return rc;
    return EACHAM_OK;
}

extern "C" int eacham_reprojection_errors(eacham_ctx* ctx, const double* transforms, int n_frames, int n,
                                          const uint32_t* frame, const double* points, const double* uv, const double* K,
                                          float* err) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n < 0 || n_frames < 0 || !K) return ctx->fail(EACHAM_ERR_INVALID, "reprojection: null argument or negative size");
    if (n == 0) return EACHAM_OK;
    if (!transforms || !frame || !points || !uv || !err) return ctx->fail(EACHAM_ERR_INVALID, "reprojection: null array");
    for (int i = 0; i < n; ++i)
        if (frame[i] >= (uint32_t)n_frames)
            return ctx->fail(EACHAM_ERR_INVALID, "reprojection: item %d names frame %u of %d", i, frame[i], n_frames);
    EACHAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t o_T = take(sizeof(double) * 16 * (size_t)n_frames), o_K = take(sizeof(double) * 4);
    const size_t o_f = take(sizeof(unsigned) * (size_t)n), o_p = take(sizeof(double) * 3 * (size_t)n);
    const size_t o_uv = take(sizeof(double) * 2 * (size_t)n), o_e = take(sizeof(float) * (size_t)n);
    if (int rc = ensure_io(ctx, off)) return rc;
    if (int rc = ensure_io_host(ctx, off)) return rc;
    char* base = (char*)ctx->io;
    hipStream_t st = ctx->stream;
    IoPack io(ctx, st);
    if (int rc = io.in(o_T, transforms, sizeof(double) * 16 * (size_t)n_frames)) return rc;
    if (int rc = io.in(o_K, K, sizeof(double) * 4)) return rc;
    if (int rc = io.in(o_f, frame, sizeof(unsigned) * (size_t)n)) return rc;
    if (int rc = io.in(o_p, points, sizeof(double) * 3 * (size_t)n)) return rc;
    if (int rc = io.in(o_uv, uv, sizeof(double) * 2 * (size_t)n)) return rc;
    if (int rc = io.flush_in()) return rc;
    reproject_kernel<<<(unsigned)((n + TRI_BLOCK - 1) / TRI_BLOCK), TRI_BLOCK, 0, st>>>(
        (const double*)(base + o_T), n, (const unsigned*)(base + o_f), (const double*)(base + o_p),
        (const double2*)(base + o_uv), (const double*)(base + o_K), (float*)(base + o_e));
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    if (int rc = io.out(err, o_e, sizeof(float) * (size_t)n)) return rc;
    if (int rc = io.finish()) return rc;
    return EACHAM_OK;
}

extern "C" int eacham_two_view_points(eacham_ctx* ctx, int n_matches, const double* uv1, const double* uv2, const double* K,
                                      int n_transforms, const double* transforms, float max_repr_error, float min_tri_angle,
                                      int angle_strict, double* points, uint8_t* keep, int32_t* counts) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_matches < 0 || n_transforms < 0 || !K) return ctx->fail(EACHAM_ERR_INVALID, "two_view: null argument or negative size");
    if (counts)
        for (int k = 0; k < n_transforms; ++k) counts[k] = 0;
    const long long total = (long long)n_matches * n_transforms;
    if (total == 0) return EACHAM_OK;
    if (!uv1 || !uv2 || !transforms || !points || !keep || !counts) return ctx->fail(EACHAM_ERR_INVALID, "two_view: null array");
    if (total > (1ll << 31) - 1) return ctx->fail(EACHAM_ERR_CAPACITY, "two_view: too many (transform, match) items");
    EACHAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t o_u1 = take(sizeof(double) * 2 * (size_t)n_matches), o_u2 = take(sizeof(double) * 2 * (size_t)n_matches);
    const size_t o_K = take(sizeof(double) * 4), o_T = take(sizeof(double) * 16 * (size_t)n_transforms);
    const size_t o_p = take(sizeof(double) * 3 * (size_t)total), o_k = take((size_t)total);
    if (int rc = ensure_io(ctx, off)) return rc;
    char* base = (char*)ctx->io;
    hipStream_t st = ctx->stream;
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(base + o_u1, uv1, sizeof(double) * 2 * (size_t)n_matches, hipMemcpyHostToDevice, st));
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(base + o_u2, uv2, sizeof(double) * 2 * (size_t)n_matches, hipMemcpyHostToDevice, st));
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(base + o_K, K, sizeof(double) * 4, hipMemcpyHostToDevice, st));
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(base + o_T, transforms, sizeof(double) * 16 * (size_t)n_transforms, hipMemcpyHostToDevice, st));
    {
        ProfileScope scope(ctx, EACHAM_KERNEL_TRIANGULATE);
        two_view_kernel<<<(unsigned)((total + TRI_BLOCK - 1) / TRI_BLOCK), TRI_BLOCK, 0, st>>>(
            n_matches, (const double2*)(base + o_u1), (const double2*)(base + o_u2), (const double*)(base + o_K), n_transforms,
            (const double*)(base + o_T), max_repr_error, min_tri_angle, angle_strict, (double*)(base + o_p),
            (unsigned char*)(base + o_k));
    }
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(points, base + o_p, sizeof(double) * 3 * (size_t)total, hipMemcpyDeviceToHost, st));
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(keep, base + o_k, (size_t)total, hipMemcpyDeviceToHost, st));
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int k = 0; k < n_transforms; ++k) {  // counts on the host: the masks are here anyway
        int32_t c = 0;
        for (int i = 0; i < n_matches; ++i) c += keep[(size_t)k * n_matches + i];
        counts[k] = c;
    }
    return EACHAM_OK;
}
