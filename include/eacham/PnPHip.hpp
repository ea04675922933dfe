// PnPHip.hpp — cv::solvePnPRansac as the reference calls it, on the device library.
//
//   cv::solvePnPRansac(pts3d1, pts2d2, K, distCoeffs /* zeros */, rvec, t, false, 10000, 4.0f, 0.999f, inliersPnP,
//                      cv::SOLVEPNP_EPNP);              /root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:227-228
//   cv::Rodrigues(rvec, R); factor.transform = ConvertToTransform(R, t);                                          :236-238
//
// OpenCV's loop is sequential: draw 5 points, EPnP, count the points within 4 px, shrink the iteration budget from the
// best inlier ratio and the confidence (RANSACUpdateNumIters), stop when the budget is used up, and run EPnP once more on
// the inliers of the winner. Here the samples are processed in CHUNKS of 256 — eacham_solve_pnp on the chunk's five-point
// rows, one eacham_score_hypotheses(kind PNP) call for its models against every point — and the sequential rule is replayed
// over the chunk's inlier counts in sample order, so the loop ends at exactly the sample OpenCV's would (`iterations`) and
// the winner is the best model among the samples before it; with 70 % inliers that is one chunk instead of the 10 000
// samples asked for. The last launch is EPnP on the one row of the winner's inliers. The samples are OpenCV's own stream by
// default (Sampling::OpenCV, CvSampling.hpp: cv::RNG seeded (uint64)-1, getSubset; from memory of the 4.5.5 sources, unverified:
// parity unpinned) or the counter-based generator seeded by the caller (Sampling::Counter); tests hold the result against the
// ground truth under both.
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
    int iterations = 0;           // samples the sequential rule consumed before RANSACUpdateNumIters stopped it
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
                                int iterations = 10000, float reprojectionError = 4.0f, double confidence = 0.999, uint64_t seed = 1,
                                Sampling sampling = Sampling::OpenCV) {
    PnPResult out;
    const int n = (int)(image.size() / 2), m = 5;
    if (n < m || object.size() != (size_t)3 * n || iterations <= 0) return out;
    const double K4[4] = {K9[0], K9[4], K9[2], K9[5]};
    const float thr = reprojectionError * reprojectionError;   // PnPRansacCallback::computeError returns squared pixels
    const int chunk = 256;
    std::vector<double> models((size_t)chunk * 12), best_model(12, 0.0);
    std::vector<int32_t> okv(chunk), inl(chunk);
    int best_inl = -1, budget = iterations, done = 0;
    CvRNG rng(0xffffffffffffffffull);   // RANSACPointSetRegistrator::run: RNG rng((uint64)-1), one stream for the whole call
    for (int first = 0; first < budget; first += chunk) {
        const int cnt = std::min(chunk, iterations - first);
        // sample s of the whole run is row (s - first) of this chunk: OpenCV's stream is drawn in sample order (a chunk draws ahead
        // of the budget, which only shrinks: the samples the sequential rule consumes are the stream's prefix); the counter-based
        // generator is indexed by the global sample number
        std::vector<int32_t> idx;
        if (sampling == Sampling::OpenCV) {
            idx.resize((size_t)cnt * m);
            for (int k = 0; k < cnt; ++k) (void)cv_get_subset(rng, n, m, &idx[(size_t)k * m], 10000, [](const int32_t*) { return true; });
        } else {
            idx = twoview_detail::draw_samples(n, m, cnt, seed, first);
        }
        ctx.check(eacham_solve_pnp(ctx.get(), n, object.data(), image.data(), K4, m, cnt, idx.data(), models.data(), okv.data()));
        ctx.check(eacham_score_hypotheses(ctx.get(), EACHAM_SCORE_PNP, n, object.data(), image.data(), cnt, models.data(), K4, thr, nullptr,
                                          inl.data(), nullptr));
        for (int k = 0; k < cnt && first + k < budget; ++k) {
            done = first + k + 1;
            if (!okv[k]) continue;
            if (inl[k] > std::max(best_inl, m - 1)) {   // strictly more inliers (and at least a sample's worth) replaces the model
                best_inl = inl[k];
                std::copy(&models[(size_t)k * 12], &models[(size_t)k * 12] + 12, best_model.begin());
                budget = twoview_detail::ransac_update_num_iters(confidence, (double)(n - inl[k]) / n, m, budget);
            }
        }
    }
    out.iterations = done;
    if (best_inl < m) return out;
    std::vector<float> err(n);
    int32_t cnt = 0;
    ctx.check(eacham_score_hypotheses(ctx.get(), EACHAM_SCORE_PNP, n, object.data(), image.data(), 1, best_model.data(), K4, thr,
                                      err.data(), &cnt, nullptr));
    std::vector<int32_t> rows;
    for (int i = 0; i < n; ++i)
        if (err[i] <= thr) rows.push_back(i);
    out.inliers.assign(rows.begin(), rows.end());
    double refit[12];
    int32_t rok = 0;
    ctx.check(eacham_solve_pnp(ctx.get(), n, object.data(), image.data(), K4, (int)rows.size(), 1, rows.data(), refit, &rok));
    const double* pose = rok ? refit : best_model.data();   // (a collinear inlier set cannot be refitted: keep the winner)
    for (int e = 0; e < 9; ++e) out.R[e] = pose[e];
    for (int e = 0; e < 3; ++e) out.t[e] = pose[9 + e];
    out.rvec = RodriguesFromMatrix(out.R);
    out.ok = true;
    return out;
}

}  // namespace hip
}  // namespace eacham
