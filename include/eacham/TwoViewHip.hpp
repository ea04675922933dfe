// TwoViewHip.hpp — the rest of RecoverPoseTwoView's OpenCV calls on top of the C-ABI
// (/root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:47-183):
//
//   cv::findEssentialMat(pts1, pts2, focal, pp, cv::LMEDS, 0.99, 4.0, 1000, mask)   :57-61   -> FindEssentialMat
//   cv::findHomography(pts1, pts2, cv::LMEDS, 4.0, mask2, 100, 0.999)               :75      -> FindHomography
//   cv::decomposeHomographyMat(H, K, Rs, ts, normals)                               :92      -> DecomposeHomographyMat
//   cv::recoverPose(E, pts1, pts2, K, R, t, 50.0f, mask)                            :150     -> RecoverPose
//
// The robust estimators are the pair (eacham_solve_minimal: every minimal sample -> its models, one launch) +
// (eacham_score_hypotheses: every model against every correspondence, one launch) with the LMedS rule of
// LMeDSPointSetRegistrator::run in between: smallest median wins, sigma = 2.5 * 1.4826 * (1 + 5 / (n - m)) * sqrt(median),
// inliers = err <= sigma^2 (ptsetreg.cpp). The SAMPLES: by default OpenCV's own stream (Sampling::OpenCV, CvSampling.hpp: cv::RNG
// seeded (uint64)-1 per call, getSubset, the homography's checkSubset — restated from memory of the 4.5.5 sources, unverified
// here like the rest: parity unpinned); Sampling::Counter is the library's counter-based generator seeded by the caller (the
// rounds 1-3 form). The tests hold the result against ground truth under both. Neither estimator is followed by OpenCV's final
// Levenberg-Marquardt polish on the inliers except findHomography's (RefitHomography below; findEssentialMat has none).
// DecomposeHomographyMat returns the four {R, t, n} of the Faugeras decomposition — the solution set of
// cv::decomposeHomographyMat (OpenCV computes it with the Malis-Vargas closed form and in another order: the reference
// keeps "the first solution with the strictly largest number of good points", so the order only matters on ties).
// RecoverPose follows cv::recoverPose: the four (R, t) of decomposeEssentialMat, every point triangulated for each
// (eacham_two_view_points), a point is good iff its depth is positive and below distanceThresh in BOTH cameras, the
// candidate with the most good points wins (first on ties).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "CvSampling.hpp"
#include "TriangulatorHip.hpp"

namespace eacham {
namespace hip {

typedef std::array<double, 9> Mat3;   // row-major
typedef std::array<double, 3> Vec3;

namespace twoview_detail {

inline Mat3 mul(const Mat3& A, const Mat3& B) {
    Mat3 C{};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
    return C;
}
inline Mat3 transpose(const Mat3& A) { return {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]}; }
inline double det(const Mat3& A) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
inline Vec3 cross(const Vec3& a, const Vec3& b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
inline double norm(const Vec3& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// A = U diag(S) V^T, S descending, U and V orthogonal (columns). One-sided Jacobi on the columns of A.
inline void svd3(const Mat3& A, Mat3& U, Vec3& S, Mat3& V) {
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a[r][c] = A[3 * r + c];
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) alpha += a[r][p] * a[r][p], beta += a[r][q] * a[r][q], gamma += a[r][p] * a[r][q];
                off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
                if (gamma == 0.0) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double x = a[r][p], y = a[r][q];
                    a[r][p] = c * x - s * y, a[r][q] = s * x + c * y;
                    const double vx = v[r][p], vy = v[r][q];
                    v[r][p] = c * vx - s * vy, v[r][q] = s * vx + c * vy;
                }
            }
        if (off < 1e-15) break;
    }
    double s[3];
    int ord[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) s[c] = std::sqrt(a[0][c] * a[0][c] + a[1][c] * a[1][c] + a[2][c] * a[2][c]);
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (s[ord[j]] > s[ord[i]]) std::swap(ord[i], ord[j]);
    Vec3 u[3];
    for (int k = 0; k < 3; ++k) {
        const int c = ord[k];
        S[k] = s[c];
        for (int r = 0; r < 3; ++r) V[3 * r + k] = v[r][c];
        if (s[c] > 1e-12 * s[ord[0]] && s[c] > 0.0)
            u[k] = {a[0][c] / s[c], a[1][c] / s[c], a[2][c] / s[c]};
        else if (k == 2)
            u[k] = cross(u[0], u[1]);  // a rank-2 matrix (an essential matrix): complete the basis
        else
            u[k] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, 0.0};
    }
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < 3; ++r) U[3 * r + k] = u[k][r];
}

// counter-based sample indices (splitmix64): sample s = m distinct indices out of n
inline uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline std::vector<int32_t> draw_samples(int n, int m, int count, uint64_t seed, int first = 0) {   // samples first .. first + count - 1
    std::vector<int32_t> idx((size_t)count * m);
    for (int s = 0; s < count; ++s) {
        uint64_t ctr = 0;
        for (int k = 0; k < m;) {
            const int v = (int)(mix(seed * 0x100000001B3ull + ((uint64_t)(first + s) << 20) + ctr++) % (uint64_t)n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup = dup || idx[(size_t)s * m + j] == v;
            if (!dup) idx[(size_t)s * m + k++] = v;
        }
    }
    return idx;
}

struct RobustModel {
    Mat3 model{};
    std::vector<uint8_t> mask;   // inliers of the LMedS rule
    int inliers = 0;
    float median = std::numeric_limits<float>::quiet_NaN();
    int iterations = 0;          // minimal samples drawn (see lmeds)
    bool ok = false;
};

// RANSACUpdateNumIters (ptsetreg.cpp): iterations after which a sample of m inliers has been drawn with probability p at
// outlier ratio ep, capped by maxIters
inline int ransac_update_num_iters(double p, double ep, int m, int maxIters) {
    p = std::min(std::max(p, 0.0), 1.0);
    ep = std::min(std::max(ep, 0.0), 1.0);
    const double num = std::max(1.0 - p, std::numeric_limits<double>::min());
    const double denom = 1.0 - std::pow(1.0 - ep, m);
    if (denom < std::numeric_limits<double>::min()) return 0;
    const double ln = std::log(num), ld = std::log(denom);
    return ld >= 0 || -ln >= maxIters * (-ld) ? maxIters : (int)std::lround(ln / ld);
}

// the LMedS loop: solve all samples, score all models, keep the smallest median, classify by sigma.
// LMeDSPointSetRegistrator::run fixes its iteration count up front from the confidence at an assumed outlier ratio of 0.45
// (at least 3, at most maxIters): 1000 asked for with confidence 0.99 and 5-point samples are 89 iterations, 100 asked for
// with 0.999 and 4-point samples are 72.
// The minimal samples of one LMeDSPointSetRegistrator::run: `iterations` subsets of m out of n. Sampling::OpenCV replays the
// registrator's own stream — RNG rng((uint64)-1), getSubset with at most 1000 attempts per subset, the homography's
// checkSubset (CvSampling.hpp) — and may return fewer subsets when getSubset gives up (run() then stops iterating);
// Sampling::Counter is the library's counter-based generator seeded by the caller.
inline std::vector<int32_t> lmeds_samples(int n, int m, int iterations, bool homography, const std::vector<double>& uv1,
                                          const std::vector<double>& uv2, uint64_t seed, Sampling sampling) {
    if (sampling == Sampling::Counter) return draw_samples(n, m, iterations, seed);
    std::vector<int32_t> idx;
    idx.reserve((size_t)iterations * m);
    CvRNG rng(0xffffffffffffffffull);
    std::vector<int32_t> sub(m);
    for (int it = 0; it < iterations; ++it) {
        const bool found = cv_get_subset(rng, n, m, sub.data(), 1000, [&](const int32_t* s) {
            if (!homography) return true;
            float a[8], b[8];
            for (int k = 0; k < 4; ++k) {
                a[2 * k] = (float)uv1[2 * (size_t)s[k]], a[2 * k + 1] = (float)uv1[2 * (size_t)s[k] + 1];
                b[2 * k] = (float)uv2[2 * (size_t)s[k]], b[2 * k + 1] = (float)uv2[2 * (size_t)s[k] + 1];
            }
            return cv_check_subset_homography(a, b, 4);
        });
        if (!found) break;
        idx.insert(idx.end(), sub.begin(), sub.end());
    }
    return idx;
}

inline RobustModel lmeds(Context& ctx, int solve_kind, int score_kind, int m, const std::vector<double>& uv1, const std::vector<double>& uv2,
                         const double* K4, int maxIters, double confidence, uint64_t seed, Sampling sampling) {
    RobustModel out;
    const int n = (int)(uv1.size() / 2), maxm = solve_kind == EACHAM_SOLVE_ESSENTIAL5 ? 10 : 1;
    if (n < m || uv2.size() != uv1.size() || maxIters <= 0) return out;
    int iterations = std::min(maxIters, std::max(ransac_update_num_iters(confidence, 0.45, m, maxIters), 3));
    const std::vector<int32_t> idx = lmeds_samples(n, m, iterations, solve_kind == EACHAM_SOLVE_HOMOGRAPHY4, uv1, uv2, seed, sampling);
    iterations = (int)(idx.size() / m);
    out.iterations = iterations;
    if (iterations == 0) return out;
    std::vector<double> models((size_t)iterations * maxm * 9);
    std::vector<int32_t> counts(iterations);
    ctx.check(eacham_solve_minimal(ctx.get(), solve_kind, n, uv1.data(), uv2.data(), K4, iterations, idx.data(), models.data(), counts.data()));
    std::vector<double> cand;
    for (int s = 0; s < iterations; ++s)
        for (int k = 0; k < counts[s]; ++k) cand.insert(cand.end(), &models[((size_t)s * maxm + k) * 9], &models[((size_t)s * maxm + k) * 9] + 9);
    const int nm = (int)(cand.size() / 9);
    if (nm == 0) return out;
    std::vector<float> med(nm);
    std::vector<int32_t> inl(nm);
    ctx.check(eacham_score_hypotheses(ctx.get(), score_kind, n, uv1.data(), uv2.data(), nm, cand.data(), K4, 0.0f, nullptr, inl.data(), med.data()));
    int best = -1;
    for (int k = 0; k < nm; ++k)
        if (med[k] == med[k] && (best < 0 || med[k] < med[best])) best = k;
    if (best < 0) return out;
    for (int e = 0; e < 9; ++e) out.model[e] = cand[(size_t)best * 9 + e];
    out.median = med[best];
    // sigma of LMeDSPointSetRegistrator::run, then the errors of the winner alone
    // (`sigma = MAX(sigma, 0.001)` before findInliers squares it: on noise-free data the median is ~0 and without the floor
    // the mask would shrink to the below-median half; for E the unit is the K-normalised one OpenCV uses too)
    const double sigma = std::max(2.5 * 1.4826 * (1.0 + 5.0 / std::max(n - m, 1)) * std::sqrt((double)med[best]), 0.001);
    const float thr = (float)(sigma * sigma);
    std::vector<float> err(n);
    int32_t cnt = 0;
    float m1 = 0;
    ctx.check(eacham_score_hypotheses(ctx.get(), score_kind, n, uv1.data(), uv2.data(), 1, out.model.data(), K4, thr, err.data(), &cnt, &m1));
    out.mask.resize(n);
    for (int i = 0; i < n; ++i) out.mask[i] = err[i] <= thr ? 1 : 0;
    out.inliers = cnt;
    out.ok = true;
    return out;
}

}  // namespace twoview_detail

using twoview_detail::RobustModel;

// cv::findEssentialMat(pts1, pts2, focal, pp, LMEDS, prob, threshold, maxIters, mask): pixels in, K4 = fx fy cx cy
// (the reference passes focal = K(0,0) and pp = (K(0,2), K(1,2)): fx = fy = focal). The model is a unit-norm E.
inline RobustModel FindEssentialMat(Context& ctx, const std::vector<double>& uv1, const std::vector<double>& uv2, const double* K4,
                                    int maxIters = 1000, uint64_t seed = 12345, double prob = 0.99, Sampling sampling = Sampling::OpenCV) {
    return twoview_detail::lmeds(ctx, EACHAM_SOLVE_ESSENTIAL5, EACHAM_SCORE_ESSENTIAL, 5, uv1, uv2, K4, maxIters, prob, seed, sampling);
}
// What cv::findHomography does with the inliers of the robust stage (fundam.cpp, "if (result && npoints > 4 ...)"): the
// normalised DLT over ALL inliers (HomographyEstimatorCallback::runKernel with count = inliers), then at most 10
// Levenberg-Marquardt iterations on the 8 free entries (H[8] = 1) of the forward transfer error. Host arithmetic: a 9 x 9
// eigenproblem and 8 x 8 solves over a few thousand points. false: fewer than 4 inliers or a degenerate configuration.
inline bool RefitHomography(const std::vector<double>& uv1, const std::vector<double>& uv2, const std::vector<uint8_t>& mask, Mat3& H) {
    const int n = (int)(uv1.size() / 2);
    std::vector<int> in;
    for (int i = 0; i < n; ++i)
        if (mask.empty() || mask[i]) in.push_back(i);
    const int m = (int)in.size();
    if (m < 4) return false;
    double cm[2] = {0, 0}, cM[2] = {0, 0}, sm[2] = {0, 0}, sM[2] = {0, 0};
    for (int i : in) { cM[0] += uv1[2 * i], cM[1] += uv1[2 * i + 1], cm[0] += uv2[2 * i], cm[1] += uv2[2 * i + 1]; }
    for (int e = 0; e < 2; ++e) cM[e] /= m, cm[e] /= m;
    for (int i : in) {
        sM[0] += std::fabs(uv1[2 * i] - cM[0]), sM[1] += std::fabs(uv1[2 * i + 1] - cM[1]);
        sm[0] += std::fabs(uv2[2 * i] - cm[0]), sm[1] += std::fabs(uv2[2 * i + 1] - cm[1]);
    }
    for (int e = 0; e < 2; ++e)
        if (!(sM[e] > 1e-12) || !(sm[e] > 1e-12)) return false;
    for (int e = 0; e < 2; ++e) sM[e] = m / sM[e], sm[e] = m / sm[e];
    double LtL[81] = {0};
    for (int i : in) {
        const double x = (uv2[2 * i] - cm[0]) * sm[0], y = (uv2[2 * i + 1] - cm[1]) * sm[1];
        const double X = (uv1[2 * i] - cM[0]) * sM[0], Y = (uv1[2 * i + 1] - cM[1]) * sM[1];
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x}, Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; ++j)
            for (int k = j; k < 9; ++k) LtL[9 * j + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; ++j)
        for (int k = 0; k < j; ++k) LtL[9 * j + k] = LtL[9 * k + j];
    // eigenvector of the smallest eigenvalue: cyclic Jacobi
    double V[81];
    for (int i = 0; i < 81; ++i) V[i] = (i % 10 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int p = 0; p < 9; ++p) {
            diag += LtL[10 * p] * LtL[10 * p];
            for (int q = p + 1; q < 9; ++q) off += LtL[9 * p + q] * LtL[9 * p + q];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double apq = LtL[9 * p + q];
                if (apq == 0.0) continue;
                const double theta = (LtL[10 * q] - LtL[10 * p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 9; ++k) { const double a = LtL[9 * k + p], b = LtL[9 * k + q]; LtL[9 * k + p] = c * a - sn * b, LtL[9 * k + q] = sn * a + c * b; }
                for (int k = 0; k < 9; ++k) { const double a = LtL[9 * p + k], b = LtL[9 * q + k]; LtL[9 * p + k] = c * a - sn * b, LtL[9 * q + k] = sn * a + c * b; }
                for (int k = 0; k < 9; ++k) { const double a = V[9 * k + p], b = V[9 * k + q]; V[9 * k + p] = c * a - sn * b, V[9 * k + q] = sn * a + c * b; }
            }
    }
    int lo = 0;
    for (int k = 1; k < 9; ++k)
        if (LtL[10 * k] < LtL[10 * lo]) lo = k;
    Mat3 H0;
    for (int k = 0; k < 9; ++k) H0[k] = V[9 * k + lo];
    const Mat3 invHnorm{1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1}, Hnorm2{sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    Mat3 Hd = twoview_detail::mul(twoview_detail::mul(invHnorm, H0), Hnorm2);
    if (!(std::fabs(Hd[8]) > 1e-300)) return false;
    for (double& v : Hd) v /= Hd[8];
    // Levenberg-Marquardt on h[0..7], residuals = projected - observed
    auto cost = [&](const Mat3& G) {
        double c = 0;
        for (int i : in) {
            const double X = uv1[2 * i], Y = uv1[2 * i + 1], w = G[6] * X + G[7] * Y + 1.0, iw = std::fabs(w) > 1e-300 ? 1.0 / w : 0.0;
            const double ex = (G[0] * X + G[1] * Y + G[2]) * iw - uv2[2 * i], ey = (G[3] * X + G[4] * Y + G[5]) * iw - uv2[2 * i + 1];
            c += ex * ex + ey * ey;
        }
        return c;
    };
    double lambda = 1e-3, c0 = cost(Hd);
    for (int it = 0; it < 10; ++it) {
        double JtJ[64] = {0}, Jte[8] = {0};
        for (int i : in) {
            const double X = uv1[2 * i], Y = uv1[2 * i + 1], w = Hd[6] * X + Hd[7] * Y + 1.0, iw = std::fabs(w) > 1e-300 ? 1.0 / w : 0.0;
            const double xi = (Hd[0] * X + Hd[1] * Y + Hd[2]) * iw, yi = (Hd[3] * X + Hd[4] * Y + Hd[5]) * iw;
            const double Jx[8] = {X * iw, Y * iw, iw, 0, 0, 0, -X * iw * xi, -Y * iw * xi}, Jy[8] = {0, 0, 0, X * iw, Y * iw, iw, -X * iw * yi, -Y * iw * yi};
            const double ex = xi - uv2[2 * i], ey = yi - uv2[2 * i + 1];
            for (int j = 0; j < 8; ++j) {
                Jte[j] += Jx[j] * ex + Jy[j] * ey;
                for (int k = j; k < 8; ++k) JtJ[8 * j + k] += Jx[j] * Jx[k] + Jy[j] * Jy[k];
            }
        }
        for (int j = 0; j < 8; ++j)
            for (int k = 0; k < j; ++k) JtJ[8 * j + k] = JtJ[8 * k + j];
        bool improved = false;
        for (int attempt = 0; attempt < 8 && !improved; ++attempt) {
            double A[64], b[8];   // (JtJ + lambda diag) d = -Jte by Cholesky
            for (int j = 0; j < 64; ++j) A[j] = JtJ[j];
            for (int j = 0; j < 8; ++j) A[9 * j] *= 1.0 + lambda, b[j] = -Jte[j];
            bool spd = true;
            for (int j = 0; j < 8 && spd; ++j) {
                for (int k = 0; k <= j; ++k) {
                    double v = A[8 * j + k];
                    for (int q = 0; q < k; ++q) v -= A[8 * j + q] * A[8 * k + q];
                    if (j == k) { spd = v > 0; A[9 * j] = spd ? std::sqrt(v) : 1.0; } else A[8 * j + k] = v / A[9 * k];
                }
            }
            if (spd) {
                for (int j = 0; j < 8; ++j) { for (int q = 0; q < j; ++q) b[j] -= A[8 * j + q] * b[q]; b[j] /= A[9 * j]; }
                for (int j = 7; j >= 0; --j) { for (int q = j + 1; q < 8; ++q) b[j] -= A[8 * q + j] * b[q]; b[j] /= A[9 * j]; }
                Mat3 G = Hd;
                for (int j = 0; j < 8; ++j) G[j] += b[j];
                const double c1 = cost(G);
                if (c1 < c0) { Hd = G, c0 = c1, lambda *= 0.1, improved = true; }
            }
            if (!improved) lambda *= 10.0;
        }
        if (!improved) break;
    }
    H = Hd;
    return true;
}

// cv::findHomography(pts1, pts2, LMEDS, ransacReprojThreshold, mask, maxIters, confidence). The model has H[8] = 1:
// the LMedS winner refitted on its inliers as above (the mask stays the winner's).
inline RobustModel FindHomography(Context& ctx, const std::vector<double>& uv1, const std::vector<double>& uv2, int maxIters = 100,
                                  uint64_t seed = 12345, double confidence = 0.999, Sampling sampling = Sampling::OpenCV) {
    RobustModel r = twoview_detail::lmeds(ctx, EACHAM_SOLVE_HOMOGRAPHY4, EACHAM_SCORE_HOMOGRAPHY, 4, uv1, uv2, nullptr, maxIters, confidence, seed, sampling);
    if (r.ok && uv1.size() / 2 > 4) {
        Mat3 H;
        if (RefitHomography(uv1, uv2, r.mask, H)) r.model = H;
    }
    return r;
}

// cv::decomposeEssentialMat: E = U diag(1, 1, 0) V^T -> R1 = U W V^T, R2 = U W^T V^T, t = U[:, 2] (|t| = 1)
inline void DecomposeEssentialMat(const Mat3& E, Mat3& R1, Mat3& R2, Vec3& t) {
    using namespace twoview_detail;
    Mat3 U, V;
    Vec3 S;
    svd3(E, U, S, V);
    if (det(U) < 0) for (double& x : U) x = -x;
    if (det(V) < 0) for (double& x : V) x = -x;
    const Mat3 W{0, 1, 0, -1, 0, 0, 0, 0, 1}, Vt = transpose(V);
    R1 = mul(mul(U, W), Vt);
    R2 = mul(mul(U, transpose(W)), Vt);
    t = {U[2], U[5], U[8]};
}

struct RecoveredPose {
    Mat3 R{};
    Vec3 t{};
    int good = 0;
    std::vector<uint8_t> mask;
    std::array<double, 16> transform() const {  // camera-1 -> camera-2, row-major 4x4 (ConvertToTransform of the reference)
        return {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2], 0, 0, 0, 1};
    }
};

// cv::recoverPose(E, pts1, pts2, K, R, t, distanceThresh, mask): K = 3x3 row-major
inline RecoveredPose RecoverPose(Context& ctx, const Mat3& E, const std::vector<double>& uv1, const std::vector<double>& uv2,
                                 const double* K, double distanceThresh = 50.0, const std::vector<uint8_t>* maskIn = nullptr) {
    Mat3 R1, R2;
    Vec3 t;
    DecomposeEssentialMat(E, R1, R2, t);
    const Mat3* Rs[4] = {&R1, &R2, &R1, &R2};
    const double sg[4] = {1, 1, -1, -1};
    std::vector<double> T(4 * 16);
    for (int k = 0; k < 4; ++k) {
        RecoveredPose p;
        p.R = *Rs[k];
        p.t = {sg[k] * t[0], sg[k] * t[1], sg[k] * t[2]};
        const auto M = p.transform();
        std::copy(M.begin(), M.end(), T.begin() + 16 * k);
    }
    const int n = (int)(uv1.size() / 2);
    const double K4[4] = {K[0], K[4], K[2], K[5]};
    std::vector<double> pts((size_t)12 * n + 3);
    std::vector<uint8_t> keep((size_t)4 * n + 1);
    std::vector<int32_t> counts(5);
    // every (candidate, point) triangulated on the device; the cheirality rule of recoverPose is applied here
    ctx.check(eacham_two_view_points(ctx.get(), n, uv1.data(), uv2.data(), K4, 4, T.data(), std::numeric_limits<float>::max(), 0.0f, 0,
                                     pts.data(), keep.data(), counts.data()));
    RecoveredPose best;
    for (int k = 0; k < 4; ++k) {
        const double* M = &T[16 * k];
        std::vector<uint8_t> mask(n);
        int good = 0;
        for (int i = 0; i < n; ++i) {
            const double* X = &pts[3 * ((size_t)k * n + i)];
            const double z1 = X[2], z2 = M[8] * X[0] + M[9] * X[1] + M[10] * X[2] + M[11];
            const bool ok = (!maskIn || (*maskIn)[i]) && z1 > 0 && z1 < distanceThresh && z2 > 0 && z2 < distanceThresh;
            mask[i] = ok;
            good += ok;
        }
        if (k == 0 || good > best.good) {
            best.R = *Rs[k];
            best.t = {sg[k] * t[0], sg[k] * t[1], sg[k] * t[2]};
            best.good = good;
            best.mask = std::move(mask);
        }
    }
    return best;
}

struct HomographyMotion {
    Mat3 R{};
    Vec3 t{}, n{};   // H_normalised ~ R + t n^T, |n| = 1, t scaled by the plane distance
};

// cv::decomposeHomographyMat(H, K, rotations, translations, normals): K = 3x3 row-major. Up to four solutions.
inline std::vector<HomographyMotion> DecomposeHomographyMat(const Mat3& H, const double* K) {
    using namespace twoview_detail;
    const Mat3 Km{K[0], K[1], K[2], K[3], K[4], K[5], K[6], K[7], K[8]};
    // K^-1 for an upper-triangular camera matrix
    const double fx = K[0], s = K[1], cx = K[2], fy = K[4], cy = K[5];
    const Mat3 Ki{1 / fx, -s / (fx * fy), (s * cy - cx * fy) / (fx * fy), 0, 1 / fy, -cy / fy, 0, 0, 1};
    Mat3 Hn = mul(mul(Ki, H), Km);
    Mat3 U, V;
    Vec3 S;
    svd3(Hn, U, S, V);
    std::vector<HomographyMotion> out;
    if (!(S[1] > 0)) return out;
    for (double& x : Hn) x /= S[1];                 // the middle singular value of a Euclidean homography is 1
    const double s1 = S[0] / S[1], s3 = S[2] / S[1];
    // H is only known up to scale AND sign: a Euclidean homography has det > 0 (both cameras see the plane from the same
    // side), so a negative determinant means the scale was negative — flip Hn and, with it, U (Hn = U S V^T stays true)
    if (det(U) * det(V) < 0) {
        for (double& x : Hn) x = -x;
        for (double& x : U) x = -x;
    }
    const Mat3 Vt = transpose(V);
    if (s1 - s3 < 1e-12) {  // pure rotation: one solution, t = 0
        HomographyMotion m;
        m.R = Hn;
        m.n = {0, 0, 1};
        out.push_back(m);
        return out;
    }
    const double a = std::sqrt(std::max(0.0, (s1 * s1 - 1.0) / (s1 * s1 - s3 * s3)));
    const double b = std::sqrt(std::max(0.0, (1.0 - s3 * s3) / (s1 * s1 - s3 * s3)));
    const double sth = std::sqrt(std::max(0.0, (s1 * s1 - 1.0) * (1.0 - s3 * s3))) / (s1 + s3);
    const double cth = (1.0 + s1 * s3) / (s1 + s3);
    for (int e1 = 1; e1 >= -1; e1 -= 2)
        for (int e3 = 1; e3 >= -1; e3 -= 2) {
            const double x1 = e1 * a, x3 = e3 * b, st = e1 * e3 * sth;
            const Mat3 Rp{cth, 0, -st, 0, 1, 0, st, 0, cth};
            const Vec3 tp{(s1 - s3) * x1, 0, -(s1 - s3) * x3}, np{x1, 0, x3};
            HomographyMotion m;
            m.R = mul(mul(U, Rp), Vt);              // det U det V = +1 here: a proper rotation
            m.t = {U[0] * tp[0] + U[1] * tp[1] + U[2] * tp[2], U[3] * tp[0] + U[4] * tp[1] + U[5] * tp[2], U[6] * tp[0] + U[7] * tp[1] + U[8] * tp[2]};
            m.n = {V[0] * np[0] + V[1] * np[1] + V[2] * np[2], V[3] * np[0] + V[4] * np[1] + V[5] * np[2], V[6] * np[0] + V[7] * np[1] + V[8] * np[2]};
            out.push_back(m);
        }
    return out;
}

}  // namespace hip
}  // namespace eacham
