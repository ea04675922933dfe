// solve.hip — batched MINIMAL SOLVERS of the robust estimators eacham calls (the scoring half is
// score.hip): one thread per caller-supplied minimal sample.
//
//   cv::findHomography(pts1, pts2, cv::LMEDS, 4.0, mask2, 100, 0.999)              ReconstructionManager.cpp:75
//       -> EACHAM_SOLVE_HOMOGRAPHY4: OpenCV 4.5.5 HomographyEstimatorCallback::runKernel (fundam.cpp): per-set point
//          normalisation, LtL of the 2 x 9 constraint rows, eigenvector of the smallest eigenvalue (cyclic Jacobi), denormalised, / H[8]
//   cv::findEssentialMat(pts1, pts2, focal, pp, cv::LMEDS, 0.99, 4.0, 1000, mask)  ReconstructionManager.cpp:57-61
//       -> EACHAM_SOLVE_ESSENTIAL5: EMEstimatorCallback::runKernel (five-point.cpp), Nister's five-point algorithm: null space of
//          the 5 x 9 epipolar system (Householder), the ten cubic constraints as a 10 x 20 matrix, Gauss-Jordan, det B(z) = a
//          degree-10 polynomial, its real roots (Durand-Kerner + Newton polish), up to ten unit-norm E per sample
//   cv::solvePnPRansac(pts3d, pts2d, K, dist, rvec, t, false, 10000, 4.0f, 0.999f, inliers, cv::SOLVEPNP_EPNP)   :227-228
//       -> eacham_solve_pnp: EPnP on every 5-point sample (the RANSAC kernel) and on the inlier set (the final refit): four control
//          points, M^T M of the projection system, its four smallest eigenvectors (Jacobi 12 x 12), the 6 x 10 distance system, three
//          linearised starts + Gauss-Newton, absolute orientation (Horn), smallest reprojection error; the point passes recompute the
//          barycentric coordinates, so a thread's state does not grow with the sample size
// OpenCV draws the samples from its own RNG: the sample INDICES are an argument here (what RANSACPointSetRegistrator /
// LMeDSPointSetRegistrator::getSubset produce), so "these correspondences -> these models" is what can be held against the
// CPU restatement the tests keep (solve_oracle.c, bit for bit: this file is compiled with -ffp-contract=off and uses only
// + - * / sqrt) — end-to-end parity with cv::findHomography / findEssentialMat cannot be pinned without that RNG stream.
// A sample is ~10^4-10^5 flops of branchy fp64 with kilobytes of private state: latency-bound, no roofline claim; the
// 1000 / 100 iterations the reference asks for are one launch.
#include "context.hpp"

#include <algorithm>
#include <cstdint>

namespace eacham {
namespace {

__device__ __forceinline__ void wave_sync_lds() {  // a wave's own LDS traffic: order its writes before its reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* cyclic Jacobi on a symmetric N x N matrix (N <= 12): A is destroyed, V's COLUMNS are the eigenvectors, w the eigenvalues.
 * Small N (3, 4) unrolls completely and stays in registers. */
template <int N, class MA, class MV>
__device__ __forceinline__ void jacobi_eig(MA A, MV V, double* w) {
    constexpr int n = N;
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    auto rotate = [&](int p, int q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) return;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < n; ++k) {  /* columns p, q */
            const double akp = A[k * n + p], akq = A[k * n + q];
            A[k * n + p] = c * akp - s * akq;
            A[k * n + q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < n; ++k) {  /* rows p, q */
            const double apk = A[p * n + k], aqk = A[q * n + k];
            A[p * n + k] = c * apk - s * aqk;
            A[q * n + k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const double vkp = V[k * n + p], vkq = V[k * n + q];
            V[k * n + p] = c * vkp - s * vkq;
            V[k * n + q] = s * vkp + c * vkq;
        }
    };
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int p = 0; p < n; ++p) {
            diag += A[p * n + p] * A[p * n + p];
#pragma unroll
            for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        if constexpr (N <= 4) {
#pragma unroll
            for (int p = 0; p < n - 1; ++p)
#pragma unroll
                for (int q = p + 1; q < n; ++q) rotate(p, q);
        } else {
            for (int p = 0; p < n - 1; ++p)
                for (int q = p + 1; q < n; ++q) rotate(p, q);
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

/* ---- five-point ---------------------------------------------------------------------------------------------- */
/* column of the monomial x^ex y^ey z^ez (total degree <= 3) in Nister's elimination order */
__device__ static int mono_col(int ex, int ey, int ez) {
    constexpr int8_t order[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                                        {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
    for (int k = 0; k < 20; ++k)
        if (order[k][0] == ex && order[k][1] == ey && order[k][2] == ez) return k;
    return -1;
}

/* ---- five-point, one WAVE per sample ---------------------------------------------------------------------------------------
 * The arithmetic of essential5 / oracle_essential5 entry by entry, spread over the lanes wherever entries are independent:
 *   null space      the Householder reflections of the 9 x 5 system: a lane per column of Q, a lane per row of P
 *   constraints     the 10 x 20 matrix: an entry per lane (four rounds), each adding ITS monomial's terms in the order of the
 *                   sequential triple loop (a table lists, per monomial, the (a, b, c) factor choices that produce it)
 *   Gauss-Jordan    on one shared copy in LDS: pivot search by every lane (same values), row swap / scale by 20 lanes, the
 *                   180 eliminated entries of a step over the wave
 *   det B(z)        every lane, in registers (a few hundred operations on identical values)
 *   roots           Durand-Kerner in its simultaneous form: root k on lane k, the other iterates by lane shuffles
 *   x, y, polish    a real root per lane: the 3 Gauss-Newton steps read the assembled constraints from LDS
 * Nothing is indexed at run time outside LDS: no scratch. A wave's LDS operations execute in program order; the fences only
 * pin the compiler. */
struct E5Lds {
    double A[200], A0[200];  // the constraints as eliminated / as assembled
    double Q[45], P[81];     // Q^T (9 x 5) and the orthogonal factor
    double lin[36];          // entry e of E as a linear form in (x, y, z, 1)
    double poly[11], mon[11];
    double B[45];            // B(z): [row][column][power of z]
    unsigned char mcol[64];     // column of the monomial x^ex y^ey z^ez at [16 ex + 4 ey + ez]
    unsigned char term[20][8];  // per monomial: the (a, b, c) choices of mul3acc that produce it, packed a | b << 2 | c << 4, in loop order
    unsigned char nterm[20];
};

__device__ __forceinline__ double readlane_f64(double v, int lane_uniform) {  // the value of lane `lane_uniform` (a wave-uniform index)
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane_uniform), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_uniform);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_max_lanes(double v) {  // max over the wave (order-independent)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

__device__ static int essential5_wave(const double* p1, const double* p2, bool has_K, double fx, double fy, double cx, double cy,
                                      double* __restrict__ Eout /* global: 10 x 9 */, E5Lds& S) {
    const int lane = threadIdx.x & 63;
    // ---- the monomial table (lane 0..19: its own monomial) ----
    if (lane < 20) {
        int n = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ex = (a == 0) + (b == 0) + (c == 0), ey = (a == 1) + (b == 1) + (c == 1), ez = (a == 2) + (b == 2) + (c == 2);
                    if (mono_col(ex, ey, ez) == lane) S.term[lane][n++] = (unsigned char)(a | (b << 2) | (c << 4));
                }
        S.nterm[lane] = (unsigned char)n;
    }
    {
        const int ex = lane >> 4, ey = (lane >> 2) & 3, ez = lane & 3;
        S.mcol[lane] = ex + ey + ez <= 3 ? (unsigned char)mono_col(ex, ey, ez) : (unsigned char)0;
    }
    // ---- Q^T ----
    if (lane < 5) {
        const int i = lane;
        double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        if (has_K) {
            x1 = (x1 - cx) / fx; y1 = (y1 - cy) / fy;
            x2 = (x2 - cx) / fx; y2 = (y2 - cy) / fy;
        }
        const double row[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
#pragma unroll
        for (int k = 0; k < 9; ++k) S.Q[k * 5 + i] = row[k];
    }
    for (int e = lane; e < 81; e += 64) S.P[e] = (e / 9 == e % 9) ? 1.0 : 0.0;
    wave_sync_lds();
    // ---- Householder QR of Q^T: lanes 0..4 own a column of Q, lanes 16..24 a row of P ----
    for (int k = 0; k < 5; ++k) {
        double norm = 0.0;
        for (int r = k; r < 9; ++r) norm += S.Q[r * 5 + k] * S.Q[r * 5 + k];
        norm = sqrt(norm);
        if (!(norm > 0.0)) return 0;  // (wave-uniform)
        double v[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) v[r] = r < k ? 0.0 : S.Q[r * 5 + k];
        {
            const double add = S.Q[k * 5 + k] >= 0.0 ? norm : -norm;
#pragma unroll
            for (int r = 0; r < 9; ++r)
                if (r == k) v[r] += add;
        }
        double vv = 0.0;
#pragma unroll
        for (int r = 0; r < 9; ++r)
            if (r >= k) vv += v[r] * v[r];
        if (!(vv > 0.0)) return 0;
        wave_sync_lds();  // every lane has read column k before anybody rewrites it
        if (lane < 5 && lane >= k) {  /* Q <- (I - 2 v v^T / vv) Q, column `lane` */
            const int c = lane;
            double d = 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r)
                if (r >= k) d += v[r] * S.Q[r * 5 + c];
            d = 2.0 * d / vv;
#pragma unroll
            for (int r = 0; r < 9; ++r)
                if (r >= k) S.Q[r * 5 + c] -= d * v[r];
        }
        if (lane >= 16 && lane < 25) {  /* P <- P (I - 2 v v^T / vv), row `lane - 16` */
            const int r = lane - 16;
            double d = 0.0;
#pragma unroll
            for (int c = 0; c < 9; ++c)
                if (c >= k) d += S.P[r * 9 + c] * v[c];
            d = 2.0 * d / vv;
#pragma unroll
            for (int c = 0; c < 9; ++c)
                if (c >= k) S.P[r * 9 + c] -= d * v[c];
        }
        wave_sync_lds();
    }
    if (lane < 36) S.lin[lane] = S.P[(lane >> 2) * 9 + 5 + (lane & 3)];
    wave_sync_lds();
    // ---- the ten cubic constraints: entry (row, col) of the 10 x 20 matrix per lane ----
    for (int e = lane; e < 200; e += 64) {
        const int row = e / 20, col = e % 20;
        const int nt = S.nterm[col];
        double acc = 0.0;
        auto mul3 = [&](int e1, int e2, int e3, double sgn) {  // acc += the terms of sgn * l(e1) l(e2) l(e3) that fall on monomial `col`
            for (int t = 0; t < nt; ++t) {
                const int tc = S.term[col][t];
                acc += sgn * (S.lin[4 * e1 + (tc & 3)] * S.lin[4 * e2 + ((tc >> 2) & 3)]) * S.lin[4 * e3 + (tc >> 4)];
            }
        };
        if (row == 0) {  /* det E */
            const int perm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
#pragma unroll
            for (int p = 0; p < 6; ++p) mul3(perm[p][0], 3 + perm[p][1], 6 + perm[p][2], p < 3 ? 1.0 : -1.0);
        } else {         /* 2 E E^T E - tr(E E^T) E */
            const int i = (row - 1) / 3, j = (row - 1) % 3;
            for (int k = 0; k < 3; ++k)
                for (int l = 0; l < 3; ++l) {
                    mul3(3 * i + l, 3 * k + l, 3 * k + j, 2.0);
                    mul3(3 * k + l, 3 * k + l, 3 * i + j, -1.0);
                }
        }
        S.A[e] = acc;
        S.A0[e] = acc;
    }
    wave_sync_lds();
    // ---- Gauss-Jordan, partial pivoting ----
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        for (int r = col + 1; r < 10; ++r)
            if (fabs(S.A[r * 20 + col]) > fabs(S.A[piv * 20 + col])) piv = r;
        if (!(fabs(S.A[piv * 20 + col]) > 1e-300)) return 0;  // (wave-uniform)
        wave_sync_lds();
        if (piv != col && lane < 20) {
            const double t = S.A[piv * 20 + lane];
            S.A[piv * 20 + lane] = S.A[col * 20 + lane];
            S.A[col * 20 + lane] = t;
        }
        wave_sync_lds();
        const double inv = 1.0 / S.A[col * 20 + col];
        wave_sync_lds();
        if (lane < 20) S.A[col * 20 + lane] *= inv;
        wave_sync_lds();
        double f[3];
        int at[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {  // the 9 x 20 entries of the other rows, three per lane: factors first, then the update
            const int e = lane + 64 * t;
            int r = e / 20;
            if (r >= col) ++r;
            at[t] = e < 180 ? r * 20 + e % 20 : -1;
            f[t] = e < 180 ? S.A[r * 20 + col] : 0.0;
        }
        wave_sync_lds();
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (at[t] >= 0 && f[t] != 0.0) S.A[at[t]] -= f[t] * S.A[col * 20 + at[t] % 20];
        wave_sync_lds();
    }
    /* B(z): rows k = e - z f, l = g - z h, m = i - z j; entries = polynomials in z (ascending), degrees 3, 3, 4. An entry per lane:
     * column j of row r reads e[top_j - k] and f[top_j - k + 1] with top = 2, 5, 9 and 3, 3, 4 coefficients of e */
    if (lane < 45) {
        const int r = lane / 15, j = (lane % 15) / 5, k = lane % 5;
        const double* e = S.A + (4 + 2 * r) * 20 + 10;
        const double* f = S.A + (5 + 2 * r) * 20 + 10;
        const int top = j == 0 ? 2 : (j == 1 ? 5 : 9), ne = j == 2 ? 4 : 3;
        double v;
        if (k == 0) v = e[top];
        else if (k < ne) v = e[top - k] - f[top - k + 1];
        else if (k == ne) v = -f[top - k + 1];
        else v = 0.0;
        S.B[lane] = v;
    }
    wave_sync_lds();
    /* det B(z) by cofactor expansion along row 0, coefficient k on lane k: for every coefficient the products enter in the order the
     * sequential polynomial multiplications add them (first factor's power ascending) */
    if (lane <= 10) {
        const int k = lane;
        double pk = 0.0;
        for (int c0 = 0; c0 < 3; ++c0) {
            const int c1 = c0 == 0 ? 1 : (c0 == 1 ? 2 : 0), c2 = c0 == 0 ? 2 : (c0 == 1 ? 0 : 1);
            const int d1 = c1 == 2 ? 4 : 3, d2 = c2 == 2 ? 4 : 3, d0 = c0 == 2 ? 4 : 3;
            if (k > d0 + d1 + d2) continue;
            double tk = 0.0;  /* term[k] = sum_i B[0][c0][i] * minor[k - i] */
            for (int i = 0; i <= d0; ++i) {
                const int j = k - i;
                if (j < 0 || j > d1 + d2) continue;
                double m1 = 0.0, m2 = 0.0;  /* minor[j] = (B[1][c1] B[2][c2] - B[1][c2] B[2][c1])[j] */
                for (int a = 0; a <= d1; ++a) {
                    const int b = j - a;
                    if (b >= 0 && b <= d2) m1 += S.B[15 + 5 * c1 + a] * S.B[30 + 5 * c2 + b];
                }
                for (int a = 0; a <= d2; ++a) {
                    const int b = j - a;
                    if (b >= 0 && b <= d1) m2 += S.B[15 + 5 * c2 + a] * S.B[30 + 5 * c1 + b];
                }
                tk += S.B[5 * c0 + i] * (m1 - m2);
            }
            pk += tk;
        }
        S.poly[k] = pk;
    }
    wave_sync_lds();
    // ---- the real roots of poly: Durand-Kerner, root k on lane k ----
    double cmax = 0.0;
    for (int k = 0; k <= 10; ++k) cmax = fmax(cmax, fabs(S.poly[k]));
    if (!(cmax > 0.0)) return 0;
    int deg = 10;
    while (deg > 0 && fabs(S.poly[deg]) <= 1e-14 * cmax) --deg;
    if (deg == 0) return 0;
    const double lead = S.poly[deg];
    if (lane <= deg) S.mon[lane] = S.poly[lane] / lead;  /* monic */
    wave_sync_lds();
    double bound = 0.0;
    for (int k = 0; k < deg; ++k) bound = fmax(bound, fabs(S.mon[k]));
    bound += 1.0;
    double zr = 0.0, zi = 0.0;
    {
        double r0 = 1.0;
        const double a0 = fabs(S.mon[0]);
        if (a0 > 0.0) {
            double y = a0 > 1.0 ? a0 : 1.0;
            for (int it = 0; it < 80; ++it) {
                double yp = 1.0;
                for (int j = 0; j < deg - 1; ++j) yp *= y;
                y = ((deg - 1) * y + a0 / yp) / deg;
            }
            r0 = y;
        }
        r0 = fmin(fmax(r0, 0.5), bound);
        double cr = 1.0, ci = 0.0;
        for (int k = 0; k < deg; ++k) {
            if (k == lane) zr = r0 * cr, zi = r0 * ci;
            const double tr = cr * 0.4 - ci * 0.9, ti = cr * 0.9 + ci * 0.4;
            cr = tr, ci = ti;
        }
    }
    const bool mine = lane < deg;
#ifndef E5_EXP_DK_SWEEPS
#define E5_EXP_DK_SWEEPS 200
#endif
    // one sweep: this lane's correction from the iterates the sweep starts with. `change <= 1e-11 bound` of the restatement = no
    // lane's correction above it (a NaN correction is ignored by fmax there and by the comparison here); tolerance and sweep
    // limit: see the CPU restatement (solve_oracle.c)
    const double stop = 1e-11 * bound;
    auto correct = [&](double pr, double pi, double dr, double di) {
        const double den = dr * dr + di * di;
        double ch = 0.0;
        if (mine && den > 0.0) {
            const double qr = (pr * dr + pi * di) / den, qi = (pi * dr - pr * di) / den;
            zr -= qr;
            zi -= qi;
            ch = fabs(qr) + fabs(qi);
        }
        return __ballot(ch > stop) == 0ull;
    };
    if (deg == 10) {  // the usual case: coefficients in registers, the other iterates by v_readlane with constant lanes
        double mm[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) mm[j] = S.mon[j];
        for (int it = 0; it < E5_EXP_DK_SWEEPS; ++it) {
            double pr = 1.0, pi = 0.0;  /* Horner on the monic polynomial */
#pragma unroll
            for (int j = 9; j >= 0; --j) {
                const double tr = pr * zr - pi * zi + mm[j], ti = pr * zi + pi * zr;
                pr = tr, pi = ti;
            }
            double dr = 1.0, di = 0.0;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const double zrj = readlane_f64(zr, j), zij = readlane_f64(zi, j);
                if (j != lane) {
                    const double ar = zr - zrj, ai = zi - zij;
                    const double tr = dr * ar - di * ai, ti = dr * ai + di * ar;
                    dr = tr, di = ti;
                }
            }
            if (correct(pr, pi, dr, di)) break;
        }
    } else {
        for (int it = 0; it < E5_EXP_DK_SWEEPS; ++it) {
            double pr = 1.0, pi = 0.0;
            for (int j = deg - 1; j >= 0; --j) {
                const double mj = S.mon[j];
                const double tr = pr * zr - pi * zi + mj, ti = pr * zi + pi * zr;
                pr = tr, pi = ti;
            }
            double dr = 1.0, di = 0.0;
            for (int j = 0; j < deg; ++j) {
                const double zrj = readlane_f64(zr, j), zij = readlane_f64(zi, j);  // (j is wave-uniform: two v_readlane, no LDS permute)
                if (j != lane) {
                    const double ar = zr - zrj, ai = zi - zij;
                    const double tr = dr * ar - di * ai, ti = dr * ai + di * ar;
                    dr = tr, di = ti;
                }
            }
            if (correct(pr, pi, dr, di)) break;
        }
    }
    bool real = mine && !(fabs(zi) > 1e-7 * (1.0 + fabs(zr)));
    double z = zr;
    if (real)
        for (int it = 0; it < 4; ++it) {  /* Newton polish on the real polynomial */
            double p = S.poly[deg], d = 0.0;
            for (int j = deg - 1; j >= 0; --j) {
                d = d * z + p;
                p = p * z + S.poly[j];
            }
            if (!(fabs(d) > 0.0)) break;
            z -= p / d;
        }
    // position of this root in the ascending list (the sequential insertion sort is stable)
    int rank = 0;
    for (int j = 0; j < deg; ++j) {
        const double zj = __shfl(z, j);
        const int rj = __shfl((int)real, j);
        if (rj && (zj < z || (zj == z && j < lane))) ++rank;
    }
    // ---- x, y and the polish, a real root per lane ----
    bool ok = false;
    double Ev[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) Ev[e] = 0.0;
    if (real) {
        double b[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double v = S.B[15 * i + 5 * j + 4];
#pragma unroll
                for (int k = 3; k >= 0; --k) v = v * z + S.B[15 * i + 5 * j + k];
                b[i][j] = v;
            }
        /* [x y 1]^T spans the null space of b: two of its rows, the pair with the largest 2 x 2 determinant */
        const double d01 = b[0][0] * b[1][1] - b[0][1] * b[1][0], d02 = b[0][0] * b[2][1] - b[0][1] * b[2][0], d12 = b[1][0] * b[2][1] - b[1][1] * b[2][0];
        int bp = 0;
        double bd = 0.0;
        if (fabs(d01) > fabs(bd)) bd = d01, bp = 0;
        if (fabs(d02) > fabs(bd)) bd = d02, bp = 1;
        if (fabs(d12) > fabs(bd)) bd = d12, bp = 2;
        if (fabs(bd) > 0.0) {
            double u0[3], u1[3];  // the two rows chosen
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                u0[c] = bp == 2 ? b[1][c] : b[0][c];
                u1[c] = bp == 0 ? b[1][c] : b[2][c];
            }
            double x = (u0[1] * u1[2] - u0[2] * u1[1]) / bd;
            double y = (u0[2] * u1[0] - u0[0] * u1[2]) / bd;
            double zz = z;
#ifndef E5_EXP_POLISH
#define E5_EXP_POLISH 3
#endif
            for (int it = 0; it < E5_EXP_POLISH; ++it) {  /* three Gauss-Newton steps on the ten constraints themselves, in (x, y, z) */
                // (rolled loops and powers by selection: unrolled, the 20 monomials and their derivatives are ~100 live registers
                // per lane and the kernel spills; the monomial's column comes from a 64-byte table)
                const double x2 = x * x, x3 = x * x * x, y2 = y * y, y3 = y * y * y, z2 = zz * zz, z3 = zz * zz * zz;
                auto pw = [](double v1, double v2, double v3, int e) { return e == 0 ? 1.0 : (e == 1 ? v1 : (e == 2 ? v2 : v3)); };
                double JtJ[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Jtr[3] = {0, 0, 0};
#pragma clang loop unroll(disable)
                for (int row = 0; row < 10; ++row) {
                    double rv = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma clang loop unroll(disable)
                    for (int ex = 0; ex <= 3; ++ex)
#pragma clang loop unroll(disable)
                        for (int ey = 0; ex + ey <= 3; ++ey)
#pragma clang loop unroll(disable)
                            for (int ez = 0; ex + ey + ez <= 3; ++ez) {
                                const double cf = S.A0[row * 20 + S.mcol[16 * ex + 4 * ey + ez]];
                                const double pxe = pw(x, x2, x3, ex), pye = pw(y, y2, y3, ey), pze = pw(zz, z2, z3, ez);
                                rv += cf * (pxe * pye) * pze;
                                if (ex) g0 += cf * (ex * pw(x, x2, x3, ex - 1) * pye) * pze;
                                if (ey) g1 += cf * (pxe * (ey * pw(y, y2, y3, ey - 1))) * pze;
                                if (ez) g2 += cf * (pxe * pye) * (ez * pw(zz, z2, z3, ez - 1));
                            }
                    const double g[3] = {g0, g1, g2};
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        Jtr[u] += g[u] * rv;
#pragma unroll
                        for (int v = 0; v < 3; ++v) JtJ[u][v] += g[u] * g[v];
                    }
                }
                /* 3 x 3 solve by cofactors */
                const double c00 = JtJ[1][1] * JtJ[2][2] - JtJ[1][2] * JtJ[2][1], c01 = JtJ[1][2] * JtJ[2][0] - JtJ[1][0] * JtJ[2][2],
                             c02 = JtJ[1][0] * JtJ[2][1] - JtJ[1][1] * JtJ[2][0];
                const double dt = JtJ[0][0] * c00 + JtJ[0][1] * c01 + JtJ[0][2] * c02;
                if (!(fabs(dt) > 0.0)) break;
                const double c10 = JtJ[0][2] * JtJ[2][1] - JtJ[0][1] * JtJ[2][2], c11 = JtJ[0][0] * JtJ[2][2] - JtJ[0][2] * JtJ[2][0],
                             c12 = JtJ[0][1] * JtJ[2][0] - JtJ[0][0] * JtJ[2][1];
                const double c20 = JtJ[0][1] * JtJ[1][2] - JtJ[0][2] * JtJ[1][1], c21 = JtJ[0][2] * JtJ[1][0] - JtJ[0][0] * JtJ[1][2],
                             c22 = JtJ[0][0] * JtJ[1][1] - JtJ[0][1] * JtJ[1][0];
                const double dx = (c00 * Jtr[0] + c10 * Jtr[1] + c20 * Jtr[2]) / dt;
                const double dy = (c01 * Jtr[0] + c11 * Jtr[1] + c21 * Jtr[2]) / dt;
                const double dz = (c02 * Jtr[0] + c12 * Jtr[1] + c22 * Jtr[2]) / dt;
                if (!(fabs(dx) + fabs(dy) + fabs(dz) < 1e300)) break;
                x -= dx, y -= dy, zz -= dz;
            }
            double nrm = 0.0;
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                Ev[e] = S.lin[4 * e] * x + S.lin[4 * e + 1] * y + S.lin[4 * e + 2] * zz + S.lin[4 * e + 3];
                nrm += Ev[e] * Ev[e];
            }
            nrm = sqrt(nrm);
            if (nrm > 0.0 && nrm < 1e300) {
                ok = true;
#pragma unroll
                for (int e = 0; e < 9; ++e) Ev[e] = Ev[e] / nrm;
            }
        }
    }
    // the models leave in ascending root order, the failed ones squeezed out
    int slot = 0, n = 0;
    for (int j = 0; j < deg; ++j) {
        const int okj = __shfl((int)ok, j), rkj = __shfl(rank, j);
        n += okj;
        if (okj && rkj < rank) ++slot;
    }
    if (ok)
#pragma unroll
        for (int e = 0; e < 9; ++e) Eout[9 * slot + e] = Ev[e];
    return n;
}

/* least squares min |A x - b| for an R x C system (C <= R <= 6, C <= 5), Householder QR on a copy, every loop unrolled: the working array and the
 * callers' matrices stay in registers (as run-time-indexed arrays they are scratch memory). Returns 0 if a column collapses. Same
 * operations in the same order as lsq_small(R, C, ...) of the CPU restatement (solve_oracle.c). */
template <int R, int C>
__device__ __forceinline__ int lsq_rc(const double* A, const double* b, double* x) {
    constexpr int r = R, c = C;
    static_assert(C <= R && R <= 6, "an over-determined or square system of at most six rows");
    double Q[R][C + 1];
#pragma unroll
    for (int i = 0; i < r; ++i) {
#pragma unroll
        for (int j = 0; j < c; ++j) Q[i][j] = A[i * c + j];
        Q[i][c] = b[i];
    }
#pragma unroll
    for (int j = 0; j < c; ++j) {
        double nrm = 0.0;
#pragma unroll
        for (int i = j; i < r; ++i) nrm += Q[i][j] * Q[i][j];
        nrm = sqrt(nrm);
        if (!(nrm > 0.0)) return 0;
        const double alpha = Q[j][j] > 0.0 ? -nrm : nrm;
        double v[6];
#pragma unroll
        for (int i = j; i < r; ++i) v[i] = Q[i][j];
        v[j] -= alpha;
        double vv = 0.0;
#pragma unroll
        for (int i = j; i < r; ++i) vv += v[i] * v[i];
        if (!(vv > 0.0)) return 0;
#pragma unroll
        for (int k = j; k <= c; ++k) {
            double d = 0.0;
#pragma unroll
            for (int i = j; i < r; ++i) d += v[i] * Q[i][k];
            d = 2.0 * d / vv;
#pragma unroll
            for (int i = j; i < r; ++i) Q[i][k] -= d * v[i];
        }
        Q[j][j] = alpha;
    }
#pragma unroll
    for (int j = c - 1; j >= 0; --j) {
        double s = Q[j][c];
#pragma unroll
        for (int k = j + 1; k < c; ++k) s -= Q[j][k] * x[k];
        x[j] = s / Q[j][j];
    }
#pragma unroll
    for (int j = 0; j < c; ++j)
        if (!(fabs(x[j]) < 1e300)) return 0;
    return 1;
}
template <int C>
__device__ __forceinline__ int lsq6(const double* A, const double* b, double* x) { return lsq_rc<6, C>(A, b, x); }

// The N x N eigenproblem (N = 9, 12) by ONE WAVE on one shared copy of A and V, in the ROUND-ROBIN ordering of
// the CPU restatement (solve_oracle.c)'s jacobi_eig_rr: a sweep is N' - 1 rounds of N' / 2 disjoint rotations (N' = N rounded up to even; position 0
// holds index 0, position j >= 1 holds 1 + ((j - 1 - round) mod (N' - 1)), pair i = positions i and N' - 1 - i, index N is a bye).
// The unit of work is an ITEM (pair i, k): a lane owns item `lane` and, for N = 12, item 64 + lane (72 items). The lane forms the rotation of
// its items' pairs ITSELF from the matrix the round starts with (the same operations on the same values in every lane that
// needs them: the same bits, and no trip through LDS and no barrier to hand six rotations round), then the two stages —
// columns p, q of row k; rows p, q at column k together with the eigenvector columns — run with every read of a stage issued before
// its first write: no element is written twice inside a stage and none is read by another item after it was written, and element by
// element the arithmetic is the restatement's, so the result is the same bits. A wave's LDS operations execute in program order:
// the barriers only pin the compiler. (Until round 5 lanes 0..5 formed the rotations and handed them over through LDS, and a
// stage made two dependent passes over its 72 items: ~3.7 k cycles per round, 133 us for the front half of a five-point EPnP sample.)
struct JacRound {   // (kept for the callers' LDS layouts; the rotations no longer pass through it)
    double c[8], s[8];
    int p[8], q[8], on[8];
};
template <int N>
__device__ __forceinline__ void jacobi_wave(double* A, double* V, double* w, JacRound& R) {
    (void)R;
    constexpr int n = N, np = N + (N & 1), half = np / 2;
    constexpr int ITEMS = half * n, PASSES = (ITEMS + 63) / 64;
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < n * n; e += 64) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    wave_sync_lds();
#ifndef JAC_EXP_SWEEPS
#define JAC_EXP_SWEEPS 60
#endif
    for (int sweep = 0; sweep < JAC_EXP_SWEEPS; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < n; ++p) {
            diag += A[p * n + p] * A[p * n + p];
            for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int round = 0; round < np - 1; ++round) {
            int ip[PASSES], iq[PASSES], ik[PASSES];
            bool on[PASSES];
            double rc[PASSES], rs[PASSES];
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int e = lane + 64 * ps, i = e / n, j1 = i, j2 = np - 1 - i;
                ik[ps] = e % n;
                const int a = j1 == 0 ? 0 : 1 + ((j1 - 1 - round) % (np - 1) + (np - 1)) % (np - 1);
                const int b = 1 + ((j2 - 1 - round) % (np - 1) + (np - 1)) % (np - 1);
                const int p = a < b ? a : b, q = a < b ? b : a;
                ip[ps] = p, iq[ps] = q;
                on[ps] = false, rc[ps] = 1.0, rs[ps] = 0.0;
                if (e < ITEMS && q < n) {
                    const double apq = A[p * n + q];
                    if (apq != 0.0) {
                        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        rc[ps] = 1.0 / sqrt(t * t + 1.0), rs[ps] = t * rc[ps];
                        on[ps] = true;
                    }
                }
            }
            {   /* columns p, q of every pair: row k of the item */
                double akp[PASSES], akq[PASSES];
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps)
                    if (on[ps]) akp[ps] = A[ik[ps] * n + ip[ps]], akq[ps] = A[ik[ps] * n + iq[ps]];
                wave_sync_lds();   // (every lane's reads of the round's start, the pivots above included, before the first write)
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps)
                    if (on[ps]) {
                        A[ik[ps] * n + ip[ps]] = rc[ps] * akp[ps] - rs[ps] * akq[ps];
                        A[ik[ps] * n + iq[ps]] = rs[ps] * akp[ps] + rc[ps] * akq[ps];
                    }
            }
            wave_sync_lds();
            {   /* rows p, q of every pair at column k; the eigenvector columns */
                double apk[PASSES], aqk[PASSES], vkp[PASSES], vkq[PASSES];
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps)
                    if (on[ps]) {
                        apk[ps] = A[ip[ps] * n + ik[ps]], aqk[ps] = A[iq[ps] * n + ik[ps]];
                        vkp[ps] = V[ik[ps] * n + ip[ps]], vkq[ps] = V[ik[ps] * n + iq[ps]];
                    }
                wave_sync_lds();
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps)
                    if (on[ps]) {
                        A[ip[ps] * n + ik[ps]] = rc[ps] * apk[ps] - rs[ps] * aqk[ps];
                        A[iq[ps] * n + ik[ps]] = rs[ps] * apk[ps] + rc[ps] * aqk[ps];
                        V[ik[ps] * n + ip[ps]] = rc[ps] * vkp[ps] - rs[ps] * vkq[ps];
                        V[ik[ps] * n + iq[ps]] = rs[ps] * vkp[ps] + rc[ps] * vkq[ps];
                    }
            }
            wave_sync_lds();
        }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

/* homography4 by one wave: L^T L (shared, an upper-triangle entry per lane, the four points added in order), the 9 x 9 eigenproblem by
 * jacobi_wave, the rest on identical values in every lane. a, b: this sample's 4 x 2 points. Returns 1 / 0; H (9) valid in every lane. */
__device__ static int homography4_wave(const double* a, const double* b, double* H, double* LtL /* 81 */, double* V /* 81 */, JacRound& R) {
    const int lane = threadIdx.x & 63;
    const int count = 4;
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < count; ++i) {
        cM[0] += a[2 * i]; cM[1] += a[2 * i + 1];
        cm[0] += b[2 * i]; cm[1] += b[2 * i + 1];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) cM[k] /= count, cm[k] /= count;
#pragma unroll
    for (int i = 0; i < count; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            sM[k] += fabs(a[2 * i + k] - cM[k]);
            sm[k] += fabs(b[2 * i + k] - cm[k]);
        }
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (fabs(sM[k]) < 2.220446049250313e-16 || fabs(sm[k]) < 2.220446049250313e-16) return 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) sM[k] = count / sM[k], sm[k] = count / sm[k];
    for (int e = lane; e < 81; e += 64) {
        const int j = e / 9, k = e % 9;
        if (k < j) continue;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < count; ++i) {
            const double x = (b[2 * i] - cm[0]) * sm[0], y = (b[2 * i + 1] - cm[1]) * sm[1];
            const double X = (a[2 * i] - cM[0]) * sM[0], Y = (a[2 * i + 1] - cM[1]) * sM[1];
            const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
            const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
            double lxj = 0, lxk = 0, lyj = 0, lyk = 0;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                if (q == j) lxj = Lx[q], lyj = Ly[q];
                if (q == k) lxk = Lx[q], lyk = Ly[q];
            }
            acc += lxj * lxk + lyj * lyk;
        }
        LtL[j * 9 + k] = acc;
        LtL[k * 9 + j] = acc;
    }
    wave_sync_lds();
    double w[9];
    jacobi_wave<9>(LtL, V, w, R);
    int best = 0;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        double wb = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if (q == best) wb = w[q];
        if (w[i] < wb) best = i;
    }
    double H0[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) H0[k] = V[k * 9 + best];
    const double inv[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1};
    const double n2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double T[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[3 * r + c] = inv[3 * r] * H0[c] + inv[3 * r + 1] * H0[3 + c] + inv[3 * r + 2] * H0[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) H[3 * r + c] = T[3 * r] * n2[c] + T[3 * r + 1] * n2[3 + c] + T[3 * r + 2] * n2[6 + c];
    if (!(fabs(H[8]) > 0.0)) return 0;
    const double s = 1.0 / H[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] *= s;
    return 1;
}

// EPnP by one wave. Lane l accumulates the points l, l + 64, ... of every pass over the points; the partial sums are added in
// lane order — for at most 64 points that is the sequential sum over the points, bit for bit, and the CPU restatement defines the
// sums of more than 64 points the same way (64 strided partials) — and the small dense algebra then runs on identical values in
// every lane, with everything that is indexed at run time (M^T M, its eigenvectors, the 6 x 10 distance system) in one SHARED
// copy in LDS. BIG = false: samples of at most 64 points (the RANSAC loop's 5-point samples: a point per lane, its two rows
// of the projection system parked in LDS, an entry of M^T M per lane) — 16 KB of LDS per wave, ten waves per CU; BIG = true: the
// all-inlier refit, any number of points, a private partial M^T M per lane (40 KB).
struct PnpLds {
    double red[64];
    double A[144], V[144];   // M^T M and its eigenvectors
    double ev[48], L[60];    // the four null vectors, the distance system
    JacRound R;
};

// sum of one value per lane in LANE ORDER over the first `terms` lanes, ((v0 + v1) + v2) + ..., returned to every lane
__device__ __forceinline__ double ordered_wave_sum(double v, double* red, int terms) {
    red[threadIdx.x & 63] = v;
    wave_sync_lds();
    double t = red[0];
    for (int l = 1; l < terms; ++l) t += red[l];
    wave_sync_lds();
    return t;
}

// What the front half leaves in registers (identical in every lane) for the back half: the control-point frame and the six squared
// control-point distances; the four null vectors (S.ev) and the distance system (S.L) stay in LDS.
struct PnpFrame {
    double c0[3], ax[3][3], sc[3], rho[6];
    bool planar;  // a coplanar point set: three control points (sc[2] == 0 marks it in the stored frame), see epnp_front
};

// Front half, ONE WAVE per sample: control points, M^T M (shared), its eigenvectors by jacobi_wave, the distance system.
template <bool BIG>
__device__ static int epnp_front(int m, const int* idx, const double* obj, const double* img, const double* K, PnpFrame& F, PnpLds& S,
                                 double* rows /* !BIG: 64 x 24 */, double* part /* BIG: 78 x 64 */) {
    if (m < 4 || (!BIG && m > 64)) return 0;
    const int lane = threadIdx.x & 63;
    const int terms = m < 64 ? m : 64;  // lanes that hold a partial sum
    auto total = [&](double v) { return ordered_wave_sum(v, S.red, terms); };
    const double fu = K[0], uc = K[2], fv = K[1], vc = K[3];
    double (&c0)[3] = F.c0;
    double (&ax)[3][3] = F.ax;
    double (&sc)[3] = F.sc;
    double (&rho)[6] = F.rho;
    /* control points: centroid + principal axes scaled by the spread along them */
    c0[0] = c0[1] = c0[2] = 0.0;
    for (int k = lane; k < m; k += 64)
#pragma unroll
        for (int e = 0; e < 3; ++e) c0[e] += obj[3 * (size_t)idx[k] + e];
#pragma unroll
    for (int e = 0; e < 3; ++e) c0[e] = total(c0[e]) / (double)m;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, V3[9], w3[3];
    for (int k = lane; k < m; k += 64) {
        double d[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) d[e] = obj[3 * (size_t)idx[k] + e] - c0[e];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) C[3 * i + j] += d[i] * d[j];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) C[i] = total(C[i]);
    jacobi_eig<3>(C, V3, w3);
    double wmax = w3[0] > w3[1] ? w3[0] : w3[1];
    wmax = wmax > w3[2] ? wmax : w3[2];
    /* ax[k] = unit axis k, sc[k] = its length: control point k+1 = c0 + sc[k] ax[k] */
    if (!(wmax > 0.0)) return 0;
    const int kmin = w3[1] < w3[0] ? (w3[2] < w3[1] ? 2 : 1) : (w3[2] < w3[0] ? 2 : 0);  // the axis of the smallest spread (ties: the lower index)
    if ((kmin != 0 && !(w3[0] > 1e-12 * wmax)) || (kmin != 1 && !(w3[1] > 1e-12 * wmax)) || (kmin != 2 && !(w3[2] > 1e-12 * wmax))) return 0;  // collinear / coincident
    const double wflat = kmin == 0 ? w3[0] : (kmin == 1 ? w3[1] : w3[2]);
    const bool planar = !(wflat > 1e-12 * wmax);
    F.planar = planar;
    /* A COPLANAR set takes the paper's three-control-point form inside the same arrays (solve_oracle.c's header): the flat axis goes
     * last with length 0 — control point 3 coincides with the centroid and carries barycentric coordinate 0 — and its three
     * diagonal entries of M^T M are set above every eigenvalue of the 9 x 9 part below. A set with volume keeps the axes as the
     * eigenproblem leaves them. */
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int src = !planar ? k : (k == 0 ? (kmin == 0 ? 1 : 0) : (k == 1 ? (kmin == 2 ? 1 : 2) : kmin));
        const double wk = src == 0 ? w3[0] : (src == 1 ? w3[1] : w3[2]);
        sc[k] = (planar && k == 2) ? 0.0 : sqrt(wk / (double)m);
#pragma unroll
        for (int e = 0; e < 3; ++e) ax[k][e] = src == 0 ? V3[3 * e] : (src == 1 ? V3[3 * e + 1] : V3[3 * e + 2]);
    }
#define EPNP_ALPHAS(i, al)                                                                        \
    {                                                                                             \
        double d_[3];                                                                             \
        _Pragma("unroll") for (int e_ = 0; e_ < 3; ++e_) d_[e_] = obj[3 * (size_t)(i) + e_] - c0[e_];               \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                           \
            (al)[k_ + 1] = (planar && k_ == 2) ? 0.0 : (ax[k_][0] * d_[0] + ax[k_][1] * d_[1] + ax[k_][2] * d_[2]) / sc[k_]; \
        (al)[0] = 1.0 - (al)[1] - (al)[2] - (al)[3];                                              \
    }
    /* M^T M of the 2m x 12 projection system  sum_j alpha_j (fu Xc_j + (uc - u) Zc_j) = 0, same with v */
    if constexpr (!BIG) {
        if (lane < m) {  // this lane's point: its two rows
            double al[4];
            EPNP_ALPHAS(idx[lane], al);
            const double du = uc - img[2 * (size_t)idx[lane]], dv = vc - img[2 * (size_t)idx[lane] + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rows[24 * lane + 3 * j] = al[j] * fu, rows[24 * lane + 3 * j + 1] = 0.0, rows[24 * lane + 3 * j + 2] = al[j] * du;
                rows[24 * lane + 12 + 3 * j] = 0.0, rows[24 * lane + 12 + 3 * j + 1] = al[j] * fv, rows[24 * lane + 12 + 3 * j + 2] = al[j] * dv;
            }
        }
        wave_sync_lds();
        for (int e = lane; e < 144; e += 64) {  // an entry per lane: the points in order, as the sequential loop adds them
            const int i = e / 12, j = e % 12;
            if (j < i) continue;
            double acc = 0.0;
            for (int k = 0; k < m; ++k) acc += rows[24 * k + i] * rows[24 * k + j] + rows[24 * k + 12 + i] * rows[24 * k + 12 + j];
            S.A[12 * i + j] = acc;
            S.A[12 * j + i] = acc;
        }
        wave_sync_lds();
    } else {
        for (int e = 0; e < 78; ++e) part[64 * e + lane] = 0.0;
        for (int k = lane; k < m; k += 64) {
            double al[4], r1[12], r2[12];
            EPNP_ALPHAS(idx[k], al);
            const double du = uc - img[2 * (size_t)idx[k]], dv = vc - img[2 * (size_t)idx[k] + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r1[3 * j] = al[j] * fu, r1[3 * j + 1] = 0.0, r1[3 * j + 2] = al[j] * du;
                r2[3 * j] = 0.0, r2[3 * j + 1] = al[j] * fv, r2[3 * j + 2] = al[j] * dv;
            }
            int e = 0;
#pragma unroll
            for (int i = 0; i < 12; ++i)
#pragma unroll
                for (int j = i; j < 12; ++j, ++e) part[64 * e + lane] += r1[i] * r1[j] + r2[i] * r2[j];
        }
        wave_sync_lds();
        for (int e = lane; e < 78; e += 64) {  // entry e of the upper triangle: the 64 partials in lane order
            double t = part[64 * e];
            for (int l = 1; l < 64; ++l) t += part[64 * e + l];
            int i = 0, rem = e;
            while (rem >= 12 - i) rem -= 12 - i, ++i;
            const int j = i + rem;
            S.A[12 * i + j] = t;
            S.A[12 * j + i] = t;
        }
        wave_sync_lds();
    }
    if (planar) {  // rows / columns 9..11 are exact zeros: their diagonal goes above every eigenvalue of the 9 x 9 part
        if (lane == 0) {
            double tr = 0.0;
            for (int i = 0; i < 9; ++i) tr += S.A[13 * i];
            for (int i = 9; i < 12; ++i) S.A[13 * i] = 2.0 * tr + 1.0;
        }
        wave_sync_lds();
    }
    double w[12];
    jacobi_wave<12>(S.A, S.V, w, S.R);
    int ord[4];  /* the four smallest eigenvalues, ascending (ties: lower index first) */
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int best = -1;
        double wbest = 0.0;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            bool used = false;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < k) used |= ord[q] == i;
            if (!used && (best < 0 || w[i] < wbest)) best = i, wbest = w[i];
        }
        ord[k] = best;
    }
    // ev[k][i] = V[12 i + ord[k]]: the shared copy, an element per lane
    if (lane < 48) {
        const int k = lane / 12, i = lane % 12;
        int o = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q == k) o = ord[q];
        S.ev[12 * k + i] = S.V[12 * i + o];
    }
    wave_sync_lds();
    /* the six control-point distance constraints, quadratic in beta: L (6 x 10) over [b00 b01 b11 b02 b12 b22 b03 b13 b23 b33] */
    double cw[4][3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        cw[0][e] = c0[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) cw[k + 1][e] = c0[e] + sc[k] * ax[k][e];
    }
    {
        constexpr int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
        if (lane < 60) {  // entry (p, col) of L per lane
            const int p = lane / 10, col = lane % 10;
            int i = 0, j = 0, cc = 0;
            for (int jj = 0; jj < 4; ++jj)
                for (int ii = 0; ii <= jj; ++ii, ++cc)
                    if (cc == col) i = ii, j = jj;
            int a_ = 0, b_ = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (q == p) a_ = pa[q], b_ = pb[q];
            double di[3], dj[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                di[e] = S.ev[12 * i + 3 * a_ + e] - S.ev[12 * i + 3 * b_ + e];
                dj[e] = S.ev[12 * j + 3 * a_ + e] - S.ev[12 * j + 3 * b_ + e];
            }
            const double d = di[0] * dj[0] + di[1] * dj[1] + di[2] * dj[2];
            S.L[10 * p + col] = i == j ? d : 2.0 * d;
        }
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            rho[p] = 0.0;
#pragma unroll
            for (int e = 0; e < 3; ++e) rho[p] += (cw[pa[p]][e] - cw[pb[p]][e]) * (cw[pa[p]][e] - cw[pb[p]][e]);
        }
        wave_sync_lds();
    }
    return 1;
}

// One linearised start of the back half (variant 0 / 1 / 2 = the first 1 / 2 / 3 null vectors' leading terms): least squares for
// the start, five Gauss-Newton steps, Horn's absolute orientation, the reprojection error. Returns the error (< 0: no pose) and
// the pose in cand[12].
template <bool LANE, int variant>
__device__ static double epnp_back_variant(int m, const int* idx, const double* obj, const double* img, const double* K, const PnpFrame& F,
                                           const double* ev, const double* L, size_t es, double* red, double* cand) {
    const int lane = LANE ? 0 : (int)(threadIdx.x & 63);
    const int kstep = LANE ? 1 : 64;
    const int terms = m < 64 ? m : 64;
    auto total = [&](double v) { return LANE ? v : ordered_wave_sum(v, red, terms); };
    const double fu = K[0], fv = K[1], uc = K[2], vc = K[3];
    const double (&c0)[3] = F.c0;
    const double (&ax)[3][3] = F.ax;
    const double (&sc)[3] = F.sc;
    const double (&rho)[6] = F.rho;
    const bool planar = F.planar;
    {
        /* linearised start: the products b_i b_j that involve only the first 1 / 2 / 3 null vectors' leading terms */
        constexpr int ncol[3] = {4, 3, 5};
        constexpr int cols[3][5] = {{0, 1, 3, 6, 0}, {0, 1, 2, 0, 0}, {0, 1, 2, 3, 4}};
        double A[30], x[5], beta[4] = {0, 0, 0, 0};
        if (planar) {
            /* three control points: the distance equations of the pairs (0,1), (0,2), (1,2) = rows 0, 1, 3; start 0 takes the first
             * null vector alone (x = b00), start 1 the first two (x = b00 b01 b11, a square system); there is no third start.
             * Gauss-Newton runs on the unknowns of the start (1 or 2 betas against three equations). */
            if constexpr (variant == 2) {
                return -1.0;
            } else {
                constexpr int rows3[3] = {0, 1, 3}, nb = variant + 1, nc3 = variant == 0 ? 1 : 3;
                double l3[3][3], rho3[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    rho3[p] = rho[rows3[p]];
#pragma unroll
                    for (int j = 0; j < 3; ++j) l3[p][j] = L[(size_t)(10 * rows3[p] + j) * es];
#pragma unroll
                    for (int j = 0; j < nc3; ++j) A[p * nc3 + j] = l3[p][j];
                }
                if (!lsq_rc<3, nc3>(A, rho3, x)) return -1.0;
                const double s = x[0] < 0.0 ? -1.0 : 1.0;
                beta[0] = sqrt(s * x[0]);
                if (variant == 1) {
                    beta[1] = s * x[2] > 0.0 ? sqrt(s * x[2]) : 0.0;
                    if (x[1] < 0.0) beta[0] = -beta[0];
                }
                if (!(beta[0] != 0.0)) return -1.0;
                for (int it = 0; it < 5; ++it) {
                    double J[6], r[3], dx[2];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const double* l = l3[p];
                        J[nb * p] = 2.0 * l[0] * beta[0] + l[1] * beta[1];
                        if (nb == 2) J[nb * p + 1] = l[1] * beta[0] + 2.0 * l[2] * beta[1];
                        r[p] = rho3[p] - (l[0] * beta[0] * beta[0] + l[1] * beta[0] * beta[1] + l[2] * beta[1] * beta[1]);
                    }
                    if (!lsq_rc<3, nb>(J, r, dx)) break;
#pragma unroll
                    for (int k = 0; k < nb; ++k) beta[k] += dx[k];
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (j < ncol[variant]) A[p * ncol[variant] + j] = L[(size_t)(10 * p + cols[variant][j]) * es];
            const int solved = variant == 0 ? lsq6<4>(A, rho, x) : (variant == 1 ? lsq6<3>(A, rho, x) : lsq6<5>(A, rho, x));
            if (!solved) return -1.0;
            if (variant == 0) {  /* x = b00 b01 b02 b03 */
                const double s = x[0] < 0.0 ? -1.0 : 1.0;
                beta[0] = sqrt(s * x[0]);
                if (!(beta[0] > 0.0)) return -1.0;
#pragma unroll
                for (int k = 1; k < 4; ++k) beta[k] = s * x[k] / beta[0];
            } else {             /* x = b00 b01 b11 (b02 b12) */
                const double s = x[0] < 0.0 ? -1.0 : 1.0;
                beta[0] = sqrt(s * x[0]);
                beta[1] = s * x[2] > 0.0 ? sqrt(s * x[2]) : 0.0;
                if (x[1] < 0.0) beta[0] = -beta[0];
                if (!(beta[0] != 0.0)) return -1.0;
                if (variant == 2) beta[2] = x[3] / beta[0];
            }
            for (int it = 0; it < 5; ++it) {  /* Gauss-Newton on the six distance equations */
                double J[24], r[6], dx[4];
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    double l[10];
#pragma unroll
                    for (int q = 0; q < 10; ++q) l[q] = L[(size_t)(10 * p + q) * es];
                    J[4 * p + 0] = 2.0 * l[0] * beta[0] + l[1] * beta[1] + l[3] * beta[2] + l[6] * beta[3];
                    J[4 * p + 1] = l[1] * beta[0] + 2.0 * l[2] * beta[1] + l[4] * beta[2] + l[7] * beta[3];
                    J[4 * p + 2] = l[3] * beta[0] + l[4] * beta[1] + 2.0 * l[5] * beta[2] + l[8] * beta[3];
                    J[4 * p + 3] = l[6] * beta[0] + l[7] * beta[1] + l[8] * beta[2] + 2.0 * l[9] * beta[3];
                    r[p] = rho[p] - (l[0] * beta[0] * beta[0] + l[1] * beta[0] * beta[1] + l[2] * beta[1] * beta[1] + l[3] * beta[0] * beta[2] +
                                     l[4] * beta[1] * beta[2] + l[5] * beta[2] * beta[2] + l[6] * beta[0] * beta[3] + l[7] * beta[1] * beta[3] +
                                     l[8] * beta[2] * beta[3] + l[9] * beta[3] * beta[3]);
                }
                if (!lsq6<4>(J, r, dx)) break;
#pragma unroll
                for (int k = 0; k < 4; ++k) beta[k] += dx[k];
            }
        }
        /* control points in the camera frame, sign from the first point's depth */
        double cc[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 3; ++e)
                cc[j][e] = beta[0] * ev[(size_t)(3 * j + e) * es] + beta[1] * ev[(size_t)(12 + 3 * j + e) * es] + beta[2] * ev[(size_t)(24 + 3 * j + e) * es] + beta[3] * ev[(size_t)(36 + 3 * j + e) * es];
        {
            double al[4];
            EPNP_ALPHAS(idx[0], al);
            const double z0 = al[0] * cc[0][2] + al[1] * cc[1][2] + al[2] * cc[2][2] + al[3] * cc[3][2];
            if (z0 < 0.0)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 3; ++e) cc[j][e] = -cc[j][e];
        }
        /* absolute orientation world -> camera (Horn's quaternion form): S = sum pc (pw - c0)^T */
        double Sm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pcm[3] = {0, 0, 0};
        for (int k = lane; k < m; k += kstep) {
            double al[4], pc[3];
            EPNP_ALPHAS(idx[k], al);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                pc[e] = al[0] * cc[0][e] + al[1] * cc[1][e] + al[2] * cc[2][e] + al[3] * cc[3][e];
                pcm[e] += pc[e];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) Sm[3 * i + j] += pc[i] * (obj[3 * (size_t)idx[k] + j] - c0[j]);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) Sm[i] = total(Sm[i]);
#pragma unroll
        for (int e = 0; e < 3; ++e) pcm[e] = total(pcm[e]) / (double)m;
        /* Sm[i][j] = sum camera_i world_j; the rotation maximising tr(R^T S) is the top eigenvector of Horn's 4 x 4 matrix
         * written for the map world -> camera (its "left" set is the world points: Sxy = sum world_x camera_y = S[y][x]) */
        const double Sxx = Sm[0], Sxy = Sm[3], Sxz = Sm[6], Syx = Sm[1], Syy = Sm[4], Syz = Sm[7], Szx = Sm[2], Szy = Sm[5], Szz = Sm[8];
        double N[16] = {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx,
                        Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz,
                        Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy,
                        Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz};
        double V4[16], w4[4];
        jacobi_eig<4>(N, V4, w4);
        double q0 = V4[0], qx = V4[4], qy = V4[8], qz = V4[12], wtop = w4[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (w4[k] > wtop) wtop = w4[k], q0 = V4[k], qx = V4[4 + k], qy = V4[8 + k], qz = V4[12 + k];
        cand[0] = q0 * q0 + qx * qx - qy * qy - qz * qz, cand[1] = 2.0 * (qx * qy - q0 * qz), cand[2] = 2.0 * (qx * qz + q0 * qy);
        cand[3] = 2.0 * (qy * qx + q0 * qz), cand[4] = q0 * q0 - qx * qx + qy * qy - qz * qz, cand[5] = 2.0 * (qy * qz - q0 * qx);
        cand[6] = 2.0 * (qz * qx - q0 * qy), cand[7] = 2.0 * (qz * qy + q0 * qx), cand[8] = q0 * q0 - qx * qx - qy * qy + qz * qz;
#pragma unroll
        for (int i = 0; i < 3; ++i) cand[9 + i] = pcm[i] - (cand[3 * i] * c0[0] + cand[3 * i + 1] * c0[1] + cand[3 * i + 2] * c0[2]);
        double err = 0.0;
        for (int k = lane; k < m; k += kstep) {
            const double* X = obj + 3 * (size_t)idx[k];
            const double xc = cand[0] * X[0] + cand[1] * X[1] + cand[2] * X[2] + cand[9];
            const double yc = cand[3] * X[0] + cand[4] * X[1] + cand[5] * X[2] + cand[10];
            const double zc = cand[6] * X[0] + cand[7] * X[1] + cand[8] * X[2] + cand[11];
            const double eu = uc + fu * xc / zc - img[2 * (size_t)idx[k]], evv = vc + fv * yc / zc - img[2 * (size_t)idx[k] + 1];
            err += sqrt(eu * eu + evv * evv);
        }
        err = total(err);
        if (!(err < 1e300)) return -1.0;
        return err;
    }
}
#undef EPNP_ALPHAS

constexpr int SOLVE_WAVES = 4;   // samples (waves) per workgroup of the minimal-sample kernels

// Samples of at most 64 points (the RANSAC loop's five-point samples) in two launches. Front: ONE WAVE per sample, SOLVE_WAVES samples
// per workgroup (no workgroup barrier anywhere: waves return on their own), leaving the sample's frame — 130 doubles: c0 3, axes 9,
// lengths 3, rho 6, null vectors 48, distance system 60, valid 1 — in the batch's arrays, field-major (element e of sample s at
// frame[e * n_samples + s]: the back half's lanes read neighbouring words). Back: ONE LANE per sample. The back half is scalar work
// with ~380 live registers: run by a whole wave per sample it held the kernel at one wave per SIMD and 64 lanes repeated every
// operation (2.0 ms for 10 000 samples); by lanes, 10 000 samples are 157 waves.
constexpr int PNP_FRAME = 130, PNP_F_EV = 21, PNP_F_L = 69, PNP_F_VALID = 129;

__global__ __launch_bounds__(64 * SOLVE_WAVES) void solve_pnp_front_kernel(const double* __restrict__ obj, const double* __restrict__ img,
                                                                          const double* __restrict__ K, int sample_size, int n_samples,
                                                                          const int* __restrict__ idx, double* __restrict__ frame) {
    __shared__ PnpLds lds[SOLVE_WAVES];
    extern __shared__ double rows_dyn[];   // SOLVE_WAVES x min(sample_size, 64) x 24: the two rows of every point of a wave's sample
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * SOLVE_WAVES + wave;
    if (s >= n_samples) return;
    double* rows = rows_dyn + (size_t)wave * 24 * (sample_size < 64 ? sample_size : 64);
    const double K4[4] = {K[0], K[1], K[2], K[3]};
    PnpFrame F;
    PnpLds& S = lds[wave];
    const int ok = epnp_front<false>(sample_size, idx + (size_t)s * sample_size, obj, img, K4, F, S, rows, nullptr);
    const size_t ns = (size_t)n_samples;
    double* dst = frame + s;
    if (lane == 0) {
        dst[PNP_F_VALID * ns] = ok ? 1.0 : 0.0;
        if (ok) {
#pragma unroll
            for (int e = 0; e < 3; ++e) dst[e * ns] = F.c0[e], dst[(12 + e) * ns] = F.sc[e];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < 3; ++e) dst[(3 + 3 * k + e) * ns] = F.ax[k][e];
#pragma unroll
            for (int q = 0; q < 6; ++q) dst[(15 + q) * ns] = F.rho[q];
        }
    }
    if (ok) {
        if (lane < 48) dst[(PNP_F_EV + lane) * ns] = S.ev[lane];
        if (lane < 60) dst[(PNP_F_L + lane) * ns] = S.L[lane];
    }
}

// A lane per (sample, linearised start): blockIdx.y is the start (0..2), ONE launch — as one lane per sample with the three starts in
// a row the kernel needed ~380 registers, spilled 122 of them (324 B of scratch per lane) and ran one wave per SIMD; as three launches
// (one instantiation each) the starts waited for one another on the stream: 36 + 32 + 32 us per RANSAC chunk of the incremental loop,
// where a chunk is four waves per start. The starts' errors and poses go to `tmp` ([start][sample][13]); solve_pnp_select_kernel
// takes the first strictly smallest, as the CPU restatement's loop over the starts does.
template <int variant>
__device__ __forceinline__ void solve_pnp_back_body(const double* __restrict__ obj, const double* __restrict__ img,
                                                    const double* __restrict__ K, int sample_size, int n_samples,
                                                    const int* __restrict__ idx, const double* __restrict__ frame,
                                                    double* __restrict__ tmp) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_samples) return;
    const size_t ns = (size_t)n_samples;
    const double* src = frame + s;
    const double K4[4] = {K[0], K[1], K[2], K[3]};
    double cand[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) cand[k] = 0.0;
    double err = -1.0;
    if (src[PNP_F_VALID * ns] != 0.0) {
        PnpFrame F;
#pragma unroll
        for (int e = 0; e < 3; ++e) F.c0[e] = src[e * ns], F.sc[e] = src[(12 + e) * ns];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 3; ++e) F.ax[k][e] = src[(3 + 3 * k + e) * ns];
#pragma unroll
        for (int q = 0; q < 6; ++q) F.rho[q] = src[(15 + q) * ns];
        F.planar = F.sc[2] == 0.0;
        const int* rows_idx = idx + (size_t)s * sample_size;
        err = epnp_back_variant<true, variant>(sample_size, rows_idx, obj, img, K4, F, src + PNP_F_EV * ns, src + PNP_F_L * ns, ns, nullptr, cand);
    }
    double* dst = tmp + ((size_t)variant * ns + s) * 13;
    dst[0] = err;
#pragma unroll
    for (int k = 0; k < 12; ++k) dst[1 + k] = cand[k];
}
__global__ __launch_bounds__(64) void solve_pnp_back_kernel(const double* __restrict__ obj, const double* __restrict__ img,
                                                            const double* __restrict__ K, int sample_size, int n_samples,
                                                            const int* __restrict__ idx, const double* __restrict__ frame,
                                                            double* __restrict__ tmp) {
    if (blockIdx.y == 0) solve_pnp_back_body<0>(obj, img, K, sample_size, n_samples, idx, frame, tmp);        // (workgroup-uniform)
    else if (blockIdx.y == 1) solve_pnp_back_body<1>(obj, img, K, sample_size, n_samples, idx, frame, tmp);
    else solve_pnp_back_body<2>(obj, img, K, sample_size, n_samples, idx, frame, tmp);
}
__global__ __launch_bounds__(256) void solve_pnp_select_kernel(int n_samples, const double* __restrict__ tmp, double* __restrict__ models, int* __restrict__ n_models) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_samples) return;
    double best = -1.0;
    int which = -1;
    for (int v = 0; v < 3; ++v) {
        const double err = tmp[((size_t)v * n_samples + s) * 13];
        if (err >= 0.0 && (best < 0.0 || err < best)) best = err, which = v;
    }
    for (int k = 0; k < 12; ++k) models[12 * (size_t)s + k] = which >= 0 ? tmp[((size_t)which * n_samples + s) * 13 + 1 + k] : 0.0;
    n_models[s] = which >= 0 ? 1 : 0;
}

// samples of more than 64 points (the all-inlier refit): one workgroup of three waves per sample. Wave 0 runs the front half (a
// partial M^T M per lane) and leaves the frame, the null vectors and the distance system in LDS; then every wave takes ONE
// linearised start of the back half (its sums over the points spread over the wave's lanes, each wave with its own reduction
// scratch) and thread 0 picks the first strictly smallest error — the order of epnp_back. One wave running the three starts in a row
// was 234 us per refit of the incremental loop.
__global__ __launch_bounds__(192) void solve_pnp_big_kernel(const double* __restrict__ obj, const double* __restrict__ img,
                                                            const double* __restrict__ K, int sample_size, const int* __restrict__ idx,
                                                            double* __restrict__ models, int* __restrict__ n_models) {
    __shared__ PnpLds lds;
    __shared__ double part[78 * 64];
    __shared__ PnpFrame frame;
    __shared__ int front_ok;
    __shared__ double red[3][64], result[3][13];
    const int s = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double K4[4] = {K[0], K[1], K[2], K[3]};
    const int* rows_idx = idx + (size_t)s * sample_size;
    if (wave == 0) {
        PnpFrame F;
        const int n = epnp_front<true>(sample_size, rows_idx, obj, img, K4, F, lds, nullptr, part);
        if (lane == 0) frame = F, front_ok = n;
    }
    __syncthreads();
    if (front_ok) {   // (workgroup-uniform)
        const PnpFrame F = frame;
        double cand[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) cand[k] = 0.0;
        double err;
        if (wave == 0) err = epnp_back_variant<false, 0>(sample_size, rows_idx, obj, img, K4, F, lds.ev, lds.L, 1, red[0], cand);
        else if (wave == 1) err = epnp_back_variant<false, 1>(sample_size, rows_idx, obj, img, K4, F, lds.ev, lds.L, 1, red[1], cand);
        else err = epnp_back_variant<false, 2>(sample_size, rows_idx, obj, img, K4, F, lds.ev, lds.L, 1, red[2], cand);
        if (lane == 0) {
            result[wave][0] = err;
#pragma unroll
            for (int k = 0; k < 12; ++k) result[wave][1 + k] = cand[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double best = -1.0;
        int which = -1;
        if (front_ok)
            for (int v = 0; v < 3; ++v)
                if (result[v][0] >= 0.0 && (best < 0.0 || result[v][0] < best)) best = result[v][0], which = v;
        for (int k = 0; k < 12; ++k) models[12 * (size_t)s + k] = which >= 0 ? result[which][1 + k] : 0.0;
        n_models[s] = which >= 0 ? 1 : 0;
    }
}

__global__ __launch_bounds__(64 * SOLVE_WAVES) void solve_h4_kernel(const double* __restrict__ a, const double* __restrict__ b, int n_samples,
                                                                   const int* __restrict__ idx, double* __restrict__ models, int* __restrict__ n_models) {
    __shared__ double LtL[SOLVE_WAVES][81], V[SOLVE_WAVES][81];
    __shared__ JacRound R[SOLVE_WAVES];
    const int wave = threadIdx.x >> 6;
    const int s = blockIdx.x * SOLVE_WAVES + wave;
    if (s >= n_samples) return;
    double pa[8], pb[8], out[9];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = idx[s * 4 + k];
        pa[2 * k] = a[2 * (size_t)i]; pa[2 * k + 1] = a[2 * (size_t)i + 1];
        pb[2 * k] = b[2 * (size_t)i]; pb[2 * k + 1] = b[2 * (size_t)i + 1];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) out[k] = 0.0;
    const int n = homography4_wave(pa, pb, out, LtL[wave], V[wave], R[wave]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) models[9 * (size_t)s + k] = out[k];
        n_models[s] = n;
    }
}

__global__ __launch_bounds__(64 * SOLVE_WAVES) void solve_e5_kernel(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ K,
                                                                   int has_K, int n_samples, const int* __restrict__ idx, double* __restrict__ models,
                                                                   int* __restrict__ n_models) {
    __shared__ E5Lds lds[SOLVE_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * SOLVE_WAVES + wave;
    if (s >= n_samples) return;
    double pa[10], pb[10];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int i = idx[s * 5 + k];
        pa[2 * k] = a[2 * (size_t)i]; pa[2 * k + 1] = a[2 * (size_t)i + 1];
        pb[2 * k] = b[2 * (size_t)i]; pb[2 * k + 1] = b[2 * (size_t)i + 1];
    }
    double fx = 1, fy = 1, cx = 0, cy = 0;
    if (has_K) fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    double* dst = models + (size_t)s * 90;
    for (int k = lane; k < 90; k += 64) dst[k] = 0.0;  // (this wave's own stores below follow in program order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const int n = essential5_wave(pa, pb, has_K != 0, fx, fy, cx, cy, dst, lds[wave]);
    if (lane == 0) n_models[s] = n;
}

}  // namespace
}  // namespace eacham

using namespace eacham;

extern "C" int eacham_solve_minimal(eacham_ctx* ctx, int kind, int n_points, const double* a, const double* b, const double* K,
                                    int n_samples, const int32_t* sample_idx, double* models, int32_t* n_models) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (kind != EACHAM_SOLVE_HOMOGRAPHY4 && kind != EACHAM_SOLVE_ESSENTIAL5) return ctx->fail(EACHAM_ERR_INVALID, "solve_minimal: unknown kind %d", kind);
    if (n_points < 0 || n_samples < 0 || (n_samples > 0 && (!a || !b || !sample_idx || !models || !n_models)))
        return ctx->fail(EACHAM_ERR_INVALID, "solve_minimal: null argument or negative size");
    if (n_samples == 0) return EACHAM_OK;
    const int m = kind == EACHAM_SOLVE_HOMOGRAPHY4 ? 4 : 5, maxm = kind == EACHAM_SOLVE_HOMOGRAPHY4 ? 1 : 10;
    for (long long k = 0; k < (long long)n_samples * m; ++k)
        if (sample_idx[k] < 0 || sample_idx[k] >= n_points)
            return ctx->fail(EACHAM_ERR_INVALID, "solve_minimal: sample index %d of %d points", (int)sample_idx[k], n_points);
    EACHAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto align256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t o_a = take(sizeof(double) * 2 * (size_t)n_points), o_b = take(sizeof(double) * 2 * (size_t)n_points), o_K = take(sizeof(double) * 4);
    const size_t o_i = take(sizeof(int) * (size_t)n_samples * m), o_m = take(sizeof(double) * 9 * (size_t)maxm * n_samples);
    const size_t o_n = take(sizeof(int) * (size_t)n_samples);
    if (int rc = ensure_io(ctx, off)) return rc;
    if (int rc = ensure_io_host(ctx, off)) return rc;
    char* base = (char*)ctx->io;
    hipStream_t st = ctx->stream;
    IoPack io(ctx, st);
    if (int rc = io.in(o_a, a, sizeof(double) * 2 * (size_t)n_points)) return rc;
    if (int rc = io.in(o_b, b, sizeof(double) * 2 * (size_t)n_points)) return rc;
    if (K)
        if (int rc = io.in(o_K, K, sizeof(double) * 4)) return rc;
    if (int rc = io.in(o_i, sample_idx, sizeof(int) * (size_t)n_samples * m)) return rc;
    if (int rc = io.flush_in()) return rc;
    {
        ProfileScope scope(ctx, EACHAM_KERNEL_SCORE);
        const unsigned grid = (unsigned)((n_samples + SOLVE_WAVES - 1) / SOLVE_WAVES);
        if (kind == EACHAM_SOLVE_HOMOGRAPHY4)
            solve_h4_kernel<<<grid, 64 * SOLVE_WAVES, 0, st>>>((const double*)(base + o_a), (const double*)(base + o_b), n_samples,
                                                               (const int*)(base + o_i), (double*)(base + o_m), (int*)(base + o_n));
        else
            solve_e5_kernel<<<grid, 64 * SOLVE_WAVES, 0, st>>>((const double*)(base + o_a), (const double*)(base + o_b), (const double*)(base + o_K),
                                                               K ? 1 : 0, n_samples, (const int*)(base + o_i), (double*)(base + o_m), (int*)(base + o_n));
    }
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    if (int rc = io.out(models, o_m, sizeof(double) * 9 * (size_t)maxm * n_samples)) return rc;
    if (int rc = io.out(n_models, o_n, sizeof(int) * (size_t)n_samples)) return rc;
    if (int rc = io.finish()) return rc;
    return EACHAM_OK;
}

extern "C" int eacham_solve_pnp(eacham_ctx* ctx, int n_points, const double* object_points, const double* image_points, const double* K,
                                int sample_size, int n_samples, const int32_t* sample_idx, double* models, int32_t* n_models) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (n_points < 0 || n_samples < 0 || (n_samples > 0 && (!object_points || !image_points || !K || !sample_idx || !models || !n_models)))
        return ctx->fail(EACHAM_ERR_INVALID, "solve_pnp: null argument or negative size");
    if (n_samples == 0) return EACHAM_OK;
    if (sample_size < 5) return ctx->fail(EACHAM_ERR_INVALID, "solve_pnp: EPnP needs at least 5 points per sample, got %d", sample_size);
    const long long total = (long long)n_samples * sample_size;
    for (long long k = 0; k < total; ++k)
        if (sample_idx[k] < 0 || sample_idx[k] >= n_points)
            return ctx->fail(EACHAM_ERR_INVALID, "solve_pnp: sample index %d of %d points", (int)sample_idx[k], n_points);
    EACHAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto align256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t o_a = take(sizeof(double) * 3 * (size_t)n_points), o_b = take(sizeof(double) * 2 * (size_t)n_points), o_K = take(sizeof(double) * 4);
    const size_t o_i = take(sizeof(int) * (size_t)total), o_m = take(sizeof(double) * 12 * (size_t)n_samples), o_n = take(sizeof(int) * (size_t)n_samples);
    const size_t o_f = take(sample_size <= 64 ? sizeof(double) * PNP_FRAME * (size_t)n_samples : 0);   // the samples' frames between the two launches
    const size_t o_t = take(sample_size <= 64 ? sizeof(double) * 3 * 13 * (size_t)n_samples : 0);        // error + pose of the three starts
    if (int rc = ensure_io(ctx, off)) return rc;
    if (int rc = ensure_io_host(ctx, o_f)) return rc;   // everything but the samples' frames, which never leave the device
    char* base = (char*)ctx->io;
    hipStream_t st = ctx->stream;
    IoPack io(ctx, st);
    if (int rc = io.in(o_a, object_points, sizeof(double) * 3 * (size_t)n_points)) return rc;
    if (int rc = io.in(o_b, image_points, sizeof(double) * 2 * (size_t)n_points)) return rc;
    if (int rc = io.in(o_K, K, sizeof(double) * 4)) return rc;
    if (int rc = io.in(o_i, sample_idx, sizeof(int) * (size_t)total)) return rc;
    if (int rc = io.flush_in()) return rc;
    {
        ProfileScope scope(ctx, EACHAM_KERNEL_SCORE);
        // Bit-identical with the CPU restatement either way: samples of at most 64 points — the RANSAC loop's — a wave per sample for the
        // shared front half, a lane per sample for the scalar back half; larger ones — the all-inlier refit — one wave for both.
        if (sample_size <= 64) {
            solve_pnp_front_kernel<<<(unsigned)((n_samples + SOLVE_WAVES - 1) / SOLVE_WAVES), 64 * SOLVE_WAVES,
                                     sizeof(double) * SOLVE_WAVES * 24 * (size_t)std::min(sample_size, 64), st>>>(
                (const double*)(base + o_a), (const double*)(base + o_b), (const double*)(base + o_K), sample_size, n_samples,
                (const int*)(base + o_i), (double*)(base + o_f));
            const unsigned gb = (unsigned)((n_samples + 63) / 64);
            solve_pnp_back_kernel<<<dim3(gb, 3), 64, 0, st>>>((const double*)(base + o_a), (const double*)(base + o_b), (const double*)(base + o_K),
                                                              sample_size, n_samples, (const int*)(base + o_i), (const double*)(base + o_f), (double*)(base + o_t));
            solve_pnp_select_kernel<<<(unsigned)((n_samples + 255) / 256), 256, 0, st>>>(n_samples, (const double*)(base + o_t), (double*)(base + o_m), (int*)(base + o_n));
        }
        else
            solve_pnp_big_kernel<<<(unsigned)n_samples, 192, 0, st>>>((const double*)(base + o_a), (const double*)(base + o_b), (const double*)(base + o_K),
                                                                     sample_size, (const int*)(base + o_i), (double*)(base + o_m), (int*)(base + o_n));
    }
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    if (int rc = io.out(models, o_m, sizeof(double) * 12 * (size_t)n_samples)) return rc;
    if (int rc = io.out(n_models, o_n, sizeof(int) * (size_t)n_samples)) return rc;
    if (int rc = io.finish()) return rc;
    return EACHAM_OK;
}
