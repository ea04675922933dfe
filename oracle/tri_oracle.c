/*
 * tri_oracle.c — CPU restatement of eacham's per-track triangulation.
 *
 * TEST INFRASTRUCTURE ONLY (see the other oracle files). PARITY UNPINNED: the reference has no tests
 * for this path and Eigen 3.4.0 (JacobiSVD) is absent here. Restated call sites:
 *   TriangulatePoint (DLT, null vector of a 4x4)   /root/reference/modules/sfm/reconstruction/Triangulator.cpp:49-88
 *   TriangulationAngle                              :21-47   (returns 0 when a ray has ~zero length)
 *   IsPositiveDepth                                 :90-94
 *   TriangulatePointRansac                          :96-186  (exhaustive pairs; see quirks below)
 *   CalcReprojectionError                           /root/reference/modules/sfm/reconstruction/ProjectionHelper.cpp:32-38
 *   transformPoint3d / Project3dPoint               /root/reference/modules/base/tools/Tools3d.h:103-119
 * The null vector is computed with a one-sided (Hestenes) Jacobi SVD — the right singular vector of
 * the smallest singular value, which is what `svd.matrixV().col(3)` is, up to sign; hnormalized()
 * removes the sign.
 * Quirks kept on purpose (SURVEY Appendix C style): the returned point is the triangulation of the
 * LAST pair tried, not of the pair with the most inliers; `point3d.z() > 0` tests the WORLD z; the
 * reprojection error is rounded to float before the comparison.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TRI_SWEEPS 12

static void null_vector_4x4(double A[4][4], double* x) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < TRI_SWEEPS; ++sweep)
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 4; ++i) {
                    al += A[i][p] * A[i][p];
                    be += A[i][q] * A[i][q];
                    ga += A[i][p] * A[i][q];
                }
                if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 4; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
    int best = 0;
    double bn = INFINITY;
    for (int j = 0; j < 4; ++j) {
        double nn = 0;
        for (int i = 0; i < 4; ++i) nn += A[i][j] * A[i][j];
        if (nn < bn) { bn = nn; best = j; }
    }
    for (int i = 0; i < 4; ++i) x[i] = V[i][best];
}

/* TriangulatePoint(p1, p2, K, T1, T2): T row-major 4x4 world->camera, K = fx fy cx cy */
void oracle_triangulate_point(const double* T1, const double* T2, const double* uv1, const double* uv2,
                              const double* K, double* point) {
    const double x1 = (uv1[0] - K[2]) / K[0], y1 = (uv1[1] - K[3]) / K[1];
    const double x2 = (uv2[0] - K[2]) / K[0], y2 = (uv2[1] - K[3]) / K[1];
    double A[4][4];
    for (int j = 0; j < 4; ++j) {
        A[1][j] = x1 * T1[8 + j] - T1[j];
        A[0][j] = y1 * T1[8 + j] - T1[4 + j];
        A[3][j] = x2 * T2[8 + j] - T2[j];
        A[2][j] = y2 * T2[8 + j] - T2[4 + j];
    }
    double x[4];
    null_vector_4x4(A, x);
    point[0] = x[0] / x[3];
    point[1] = x[1] / x[3];
    point[2] = x[2] / x[3];
}

static void cam_center(const double* T, double* c) { /* translation of T^-1 (rigid) */
    for (int i = 0; i < 3; ++i) c[i] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
}

double oracle_triangulation_angle(const double* T1, const double* T2, const double* X) {
    double c1[3], c2[3], r1[3], r2[3];
    cam_center(T1, c1);
    cam_center(T2, c2);
    for (int i = 0; i < 3; ++i) { r1[i] = X[i] - c1[i]; r2[i] = X[i] - c2[i]; }
    const double n1 = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
    if (fabs(n1) < 0.0000001f) return 0.0;
    const double n2 = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
    if (fabs(n2) < 0.0000001f) return 0.0;
    const double dot = r1[0] * r2[0] + r1[1] * r2[1] + r1[2] * r2[2];
    const double ang = acos(dot / (n1 * n2));
    const double PI = 3.14159265358979323846;
    return ang < PI - ang ? ang : PI - ang;
}

static int is_inlier(const double* T, const double* uv, const double* K, const double* X, float max_err) {
    const double px = T[0] * X[0] + T[1] * X[1] + T[2] * X[2] + T[3];
    const double py = T[4] * X[0] + T[5] * X[1] + T[6] * X[2] + T[7];
    const double pz = T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11];
    const double u = (K[0] * px) / pz + K[2], v = (K[1] * py) / pz + K[3];
    const float err = (float)sqrt((uv[0] - u) * (uv[0] - u) + (uv[1] - v) * (uv[1] - v));
    const int depth_ok = (T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11]) >= 2.220446049250313e-16;
    return err < max_err && depth_ok;
}

/* TriangulatePointRansac for one track of m observations (frame transform index + pixel each).
 * Returns status bits: 1 = TriangulatePointRansac returned true; 2 = the mask is non-empty and every
 * observation is an inlier. TriangulateFrame adds the point iff status == 3 (:270-275).
 * point/mask are always filled as the reference leaves them. */
int oracle_triangulate_track(const double* transforms, const uint32_t* frame, const double* uv, int m, const double* K,
                             float max_err, float min_angle, double* point, uint8_t* mask) {
    memset(mask, 0, (size_t)(m > 0 ? m : 0));
    point[0] = point[1] = point[2] = 0.0;
    if (m < 2) return 0;
    int ransac_ok, mask_len;
    if (m < 3) {
        const double *T0 = transforms + 16 * (size_t)frame[0], *T1 = transforms + 16 * (size_t)frame[1];
        oracle_triangulate_point(T0, T1, uv, uv + 2, K, point);
        if (oracle_triangulation_angle(T0, T1, point) < (double)min_angle) return 0;
        for (int i = 0; i < m; ++i) mask[i] = (uint8_t)is_inlier(transforms + 16 * (size_t)frame[i], uv + 2 * i, K, point, max_err);
        ransac_ok = point[2] > 0.0;
        mask_len = m;
    } else {
        int best = 0;
        uint8_t* loc = (uint8_t*)malloc((size_t)m);
        mask_len = 0;
        for (int r1 = 0; r1 < m - 1; ++r1)
            for (int r2 = r1 + 1; r2 < m; ++r2) {
                const double *T1 = transforms + 16 * (size_t)frame[r1], *T2 = transforms + 16 * (size_t)frame[r2];
                oracle_triangulate_point(T1, T2, uv + 2 * r1, uv + 2 * r2, K, point);
                if (oracle_triangulation_angle(T1, T2, point) >= (double)min_angle) {
                    int inl = 0;
                    for (int i = 0; i < m; ++i) {
                        loc[i] = (uint8_t)is_inlier(transforms + 16 * (size_t)frame[i], uv + 2 * i, K, point, max_err);
                        inl += loc[i];
                    }
                    if (inl > best) {
                        best = inl;
                        memcpy(mask, loc, (size_t)m);
                        mask_len = m;
                    }
                }
            }
        free(loc);
        ransac_ok = point[2] > 0.0 && best > 2;
    }
    int cnt = 0;
    for (int i = 0; i < m; ++i) cnt += mask[i];
    return (ransac_ok ? 1 : 0) | ((mask_len > 0 && cnt == m) ? 2 : 0);
}

/* batch: tracks in CSR form. status[t] (bits above), points[t][3], masks per observation (any track length). */
int oracle_triangulate_tracks(const double* transforms, int n_tracks, const int32_t* track_ptr, const uint32_t* obs_frame,
                              const double* obs_uv, const double* K, float max_err, float min_angle, double* points,
                              int32_t* accept, uint8_t* masks) {
    for (int t = 0; t < n_tracks; ++t)
        if (track_ptr[t + 1] < track_ptr[t]) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int t = 0; t < n_tracks; ++t) {
        const int o0 = track_ptr[t], m = track_ptr[t + 1] - o0;
        accept[t] = oracle_triangulate_track(transforms, obs_frame + o0, obs_uv + 2 * (size_t)o0, m, K, max_err, min_angle,
                                             points + 3 * (size_t)t, masks + o0);
    }
    return 0;
}

/* CalcReprojectionError(uv, transformPoint3d(X, T), K) per item: float, as the reference returns it. */
void oracle_reprojection_errors(const double* transforms, int n, const uint32_t* frame, const double* points,
                                const double* uv, const double* K, float* err) {
    for (int i = 0; i < n; ++i) {
        const double *T = transforms + 16 * (size_t)frame[i], *X = points + 3 * (size_t)i;
        const double px = T[0] * X[0] + T[1] * X[1] + T[2] * X[2] + T[3];
        const double py = T[4] * X[0] + T[5] * X[1] + T[6] * X[2] + T[7];
        const double pz = T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11];
        const double u = (K[0] * px) / pz + K[2], v = (K[1] * py) / pz + K[3];
        err[i] = (float)sqrt((uv[2 * i] - u) * (uv[2 * i] - u) + (uv[2 * i + 1] - v) * (uv[2 * i + 1] - v));
    }
}

/* Two-view structure for candidate relative poses: the per-match loops of RecoverPoseTwoView
 * (/root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:118-143 homography branch,
 * :162-186 essential-matrix branch). Camera 1 is the identity, `transforms[k]` maps camera-1 to camera-2
 * coordinates. keep = z > 0 && reprojection error in camera 1 (rounded to float) < max_err &&
 * angle > min_angle (angle_strict, the homography branch) or angle >= min_angle (the other branch). */
void oracle_two_view_points(int n, const double* uv1, const double* uv2, const double* K, int nt, const double* transforms,
                            float max_err, float min_angle, int angle_strict, double* points, uint8_t* keep, int32_t* counts) {
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int k = 0; k < nt; ++k) {
        const double* T = transforms + 16 * (size_t)k;
        int32_t cnt = 0;
        for (int i = 0; i < n; ++i) {
            double* X = points + 3 * ((size_t)k * n + i);
            oracle_triangulate_point(I4, T, uv1 + 2 * (size_t)i, uv2 + 2 * (size_t)i, K, X);
            int ok = 0;
            if (!(X[2] <= 0.0)) {
                const double u = (K[0] * X[0]) / X[2] + K[2], v = (K[1] * X[1]) / X[2] + K[3];
                const float err = (float)sqrt((uv1[2 * i] - u) * (uv1[2 * i] - u) + (uv1[2 * i + 1] - v) * (uv1[2 * i + 1] - v));
                const double ang = oracle_triangulation_angle(I4, T, X);
                ok = err < max_err && (angle_strict ? ang > (double)min_angle : !(ang < (double)min_angle));
            }
            keep[(size_t)k * n + i] = (uint8_t)ok;
            cnt += ok;
        }
        counts[k] = cnt;
    }
}
