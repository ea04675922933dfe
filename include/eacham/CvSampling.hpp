// CvSampling.hpp — the SAMPLE STREAM of OpenCV's robust estimators, restated.
//
// cv::findEssentialMat / cv::findHomography (cv::LMEDS) and cv::solvePnPRansac, as the reference calls them
// (/root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:57-61, :75, :227-228), draw their minimal samples
// inside LMeDSPointSetRegistrator::run / RANSACPointSetRegistrator::run (OpenCV 4.5.5, modules/calib3d/src/ptsetreg.cpp):
//   * both seed a fresh generator per call:          RNG rng((uint64)-1);
//   * cv::RNG is a multiply-with-carry generator:    state = (uint64)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
//                                                    next() = (unsigned)state;   uniform(a, b) = a + next() % (b - a)
//   * getSubset draws modelPoints indices one by one with rng.uniform(0, count), drawing again while the index repeats one
//     already in the subset; the finished subset goes through the callback's checkSubset and is drawn afresh when that
//     refuses it (at most 1000 attempts under LMedS, 10 000 under RANSAC). Only HomographyEstimatorCallback has a
//     checkSubset (fundam.cpp): the last point must not be collinear with any two earlier ones, in either image, and the four
//     correspondences must keep their orientation (the sign test of Marquez-Neila et al.).
// OpenCV is not in this image and none of it could be run here: everything above is written FROM MEMORY of the 4.5.5 sources
// and is unverified ("parity unpinned", like the rest of the estimators). What it buys is that the documented deviation
// "the samples come from the library's own counter-based generator" is gone by default: with OpenCV at hand, E / H / the PnP
// pose can be compared model for model. tests/test_cv_sampling.py pins the generator's first draws against an independent
// statement of the recurrence.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

namespace eacham {
namespace hip {

enum class Sampling {
    OpenCV,   // cv::RNG((uint64)-1) + getSubset (+ checkSubset for the homography): the default
    Counter   // the counter-based generator of rounds 1-3 (splitmix64 of (seed, sample, draw)), seeded by the caller
};

struct CvRNG {   // cv::RNG (core/include/opencv2/core.hpp, core/operations.hpp)
    uint64_t state;
    explicit CvRNG(uint64_t s = 0xffffffffu) : state(s ? s : 0xffffffffu) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

namespace cvsampling_detail {

// haveCollinearPoints (fundam.cpp): only the LAST point of the subset is tested against the lines through two earlier ones
inline bool have_collinear_points(const float* p /* count x 2 */, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];   // (float differences, widened)
        for (int k = 0; k < j; ++k) {
            const double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
        }
    }
    return false;
}
inline double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

}  // namespace cvsampling_detail

// HomographyEstimatorCallback::checkSubset for a subset of `count` correspondences (src, dst: count x 2, cv::Point2f values)
inline bool cv_check_subset_homography(const float* src, const float* dst, int count) {
    using namespace cvsampling_detail;
    if (have_collinear_points(src, count) || have_collinear_points(dst, count)) return false;
    if (count == 4) {
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; ++i) {
            const int* t = tt[i];
            const double A[9] = {src[2 * t[0]], src[2 * t[0] + 1], 1.0, src[2 * t[1]], src[2 * t[1] + 1], 1.0, src[2 * t[2]], src[2 * t[2] + 1], 1.0};
            const double B[9] = {dst[2 * t[0]], dst[2 * t[0] + 1], 1.0, dst[2 * t[1]], dst[2 * t[1] + 1], 1.0, dst[2 * t[2]], dst[2 * t[2] + 1], 1.0};
            negative += det3(A) * det3(B) < 0;
        }
        if (negative != 0 && negative != 4) return false;
    }
    return true;
}

// One getSubset call: m distinct indices out of n into idx[0..m). `check(idx)` is the callback's checkSubset (may be empty).
// Returns false when maxAttempts subsets were refused (the registrator then stops: found == false).
template <class Check>
inline bool cv_get_subset(CvRNG& rng, int n, int m, int32_t* idx, int maxAttempts, Check check) {
    for (int iters = 0; iters < maxAttempts; ++iters) {
        for (int i = 0; i < m; ++i) {
            int v = rng.uniform(0, n);
            for (;;) {
                bool dup = false;
                for (int j = 0; j < i; ++j) dup = dup || idx[j] == v;
                if (!dup) break;
                v = rng.uniform(0, n);
            }
            idx[i] = v;
        }
        if (check(idx)) return true;
    }
    return false;
}

}  // namespace hip
}  // namespace eacham
