// PnPHip.hpp — cv::solvePnPRansac as the reference calls it, on the device library.
//
//   cv::solvePnPRansac(pts3d1, pts2d2, K, distCoeffs /* zeros */, rvec, t, false, 10000, 4.0f, 0.999f, inliersPnP,
//                      cv::SOLVEPNP_EPNP);              /root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:227-228
//   cv::Rodrigues(rvec, R); factor.transform = ConvertToTransform(R, t);                                          :236-238
//
// OpenCV's loop is sequential: draw 5 points, EPnP, count the points within 4 px, shrink the iteration budget from the
// best inlier ratio and the confidence, and at the end run EPnP once more on the inliers of the winner. Here the three
// steps are three launches over ALL candidate samples: eacham_solve_pnp on `iterations` five-point rows, one
// eacham_score_hypotheses(kind PNP) call for every model against every point, eacham_solve_pnp on the one row of the
// winner's inliers. The early exit is only a cost saving in OpenCV (a model found later can only have MORE inliers), so
// evaluating the full budget returns a model at least as good; the iteration count OpenCV WOULD have stopped at is
// reported for reference (`opencv_iterations`). The samples come from a counter-based generator seeded by the caller —
// OpenCV's own RNG stream is not reproduced (parity unpinned; tests hold the result against the ground truth).
#pragma once

#include <algorithm>
#include <cmath>

#include "TwoViewHip.hpp"

namespace eacham {
namespace hip {

struct PnPResult {
    bool ok = false;              // false: fewer than 5 points, or no sample gave a model with >= 5 inliers (cv returns false)
    Mat3 R{};                     // cv::Rodrigues(rvec)
    Vec3 rvec{};                  // axis * angle, what solvePnPRansac hands back
    Vec3 t{};
    std::vector<int> inliers;     // indices within 4 px of the RANSAC winner (OpenCV's `inliers` output)
    int opencv_iterations = 0;    // where RANSACUpdateNumIters would have stopped the sequential loop
};

inline Vec3 RodriguesFromMatrix(const Mat3& R) {   // rotation matrix -> axis * angle
    const double c = std::min(1.0, std::max(-1.0, (R[0] + R[4] + R[8] - 1.0) * 0.5));
    const double theta = std::acos(c);
    Vec3 ax{R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double s = twoview_detail::norm(ax) * 0.5;   // sin(theta)
    if (s > 1e-9) {
        for (double& x : ax) x *= theta / (2.0 * s);
        return ax;
    }
    if (c > 0.0) return {0.0, 0.0, 0.0};
    // theta = pi: axis from the diagonal of (R + I) / 2
    Vec3 v{std::sqrt(std::max(0.0, (R[0] + 1.0) * 0.5)), std::sqrt(std::max(0.0, (R[4] + 1.0) * 0.5)), std::sqrt(std::max(0.0, (R[8] + 1.0) * 0.5))};
    if (R[1] + R[3] < 0.0) v[1] = -v[1];
    if (R[2] + R[6] < 0.0) v[2] = -v[2];
    if (v[0] == 0.0 && R[5] + R[7] < 0.0) v[2] = -v[2];
    const double n = twoview_detail::norm(v);
    for (double& x : v) x *= theta / (n > 0.0 ? n : 1.0);
    return v;
}

// object: n x 3, image: n x 2 (pixels), K9: row-major 3 x 3 (no distortion — the reference passes zeros).
inline PnPResult SolvePnPRansac(Context& ctx, const std::vector<double>& object, const std::vector<double>& image, const double* K9,
                                int iterations = 10000, float reprojectionError = 4.0f, double confidence = 0.999, uint64_t seed = 1) {
    PnPResult out;
    const int n = (int)(image.size() / 2), m = 5;
    if (n < m || object.size() != (size_t)3 * n || iterations <= 0) return out;
    const double K4[4] = {K9[0], K9[4], K9[2], K9[5]};
    const std::vector<int32_t> idx = twoview_detail::draw_samples(n, m, iterations, seed);
    std::vector<double> models((size_t)iterations * 12);
    std::vector<int32_t> okv(iterations), inl(iterations);
    ctx.check(eacham_solve_pnp(ctx.get(), n, object.data(), image.data(), K4, m, iterations, idx.data(), models.data(), okv.data()));
    const float thr = reprojectionError * reprojectionError;   // PnPRansacCallback::computeError returns squared pixels
    ctx.check(eacham_score_hypotheses(ctx.get(), EACHAM_SCORE_PNP, n, object.data(), image.data(), iterations, models.data(), K4, thr, nullptr,
                                      inl.data(), nullptr));
    int best = -1, budget = iterations;
    for (int s = 0; s < iterations; ++s) {
        if (!okv[s]) continue;
        if (best < 0 || inl[s] > inl[best]) {   // strictly more inliers replaces the model, as in RANSACPointSetRegistrator::run
            best = s;
            // RANSACUpdateNumIters(confidence, outlier ratio, model points, budget)
            const double ep = std::min(1.0, std::max(0.0, (double)(n - inl[s]) / n));
            const double num = std::log(std::max(1.0 - confidence, 1e-300)), denom = std::log(std::max(1.0 - std::pow(1.0 - ep, m), 1e-300));
            if (denom < 0.0 && -num < (double)budget * -denom) budget = (int)std::lround(num / denom);
        }
        if (out.opencv_iterations == 0 && s + 1 >= budget) out.opencv_iterations = s + 1;
    }
    if (out.opencv_iterations == 0) out.opencv_iterations = iterations;
    if (best < 0 || inl[best] < m) return out;
    std::vector<float> err(n);
    int32_t cnt = 0;
    ctx.check(eacham_score_hypotheses(ctx.get(), EACHAM_SCORE_PNP, n, object.data(), image.data(), 1, &models[(size_t)best * 12], K4, thr,
                                      err.data(), &cnt, nullptr));
    std::vector<int32_t> rows;
    for (int i = 0; i < n; ++i)
        if (err[i] <= thr) rows.push_back(i);
    out.inliers.assign(rows.begin(), rows.end());
    double refit[12];
    int32_t rok = 0;
    ctx.check(eacham_solve_pnp(ctx.get(), n, object.data(), image.data(), K4, (int)rows.size(), 1, rows.data(), refit, &rok));
    const double* pose = rok ? refit : &models[(size_t)best * 12];   // (a coplanar inlier set cannot be refitted: keep the winner)
    for (int e = 0; e < 9; ++e) out.R[e] = pose[e];
    for (int e = 0; e < 3; ++e) out.t[e] = pose[9 + e];
    out.rvec = RodriguesFromMatrix(out.R);
    out.ok = true;
    return out;
}

}  // namespace hip
}  // namespace eacham
