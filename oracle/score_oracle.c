/*
 * score_oracle.c — CPU restatement of the per-hypothesis scoring inside the robust estimators eacham calls
 * (SURVEY.md §8(f) rank 3).
 *
 * TEST INFRASTRUCTURE ONLY. PARITY UNPINNED: the arithmetic lives in OpenCV 4.5.5 (conanfile.txt:3), which is not in
 * the reference tree; restated from its published sources (modules/calib3d/src: five-point.cpp, fundam.cpp,
 * solvepnp.cpp, ptsetreg.cpp), anchored on the reference's call sites:
 *   cv::findEssentialMat(pts1, pts2, focal, pp, cv::LMEDS, 0.99, 4.0, 1000, mask)  ReconstructionManager.cpp:57-61
 *   cv::findHomography(pts1, pts2, cv::LMEDS, 4.0, mask2, 100, 0.999)              ReconstructionManager.cpp:75
 *   cv::solvePnPRansac(pts3d, pts2d, K, dist = 0, ..., 10000, 4.0f, 0.999f, inliers, SOLVEPNP_EPNP)   :227-228
 * What is restated is the part that is data-parallel over (hypothesis, correspondence): the error every model
 * assigns to every point, the inlier count under a threshold (RANSAC's criterion, findInliers: err <= t) and the
 * median (LMedS' criterion). Minimal-sample drawing and the minimal solvers stay with the caller: they are tied to
 * OpenCV's RNG state, so no parity could be stated for a re-implementation.
 *   kind 0 ESSENTIAL   EMEstimatorCallback::computeError: x = (u - cx)/fx, (v - cy)/fy when K is given;
 *                      err = (float)( (x2' E x1)^2 / (Ex1[0]^2 + Ex1[1]^2 + Etx2[0]^2 + Etx2[1]^2) ), double arithmetic
 *   kind 1 HOMOGRAPHY  HomographyEstimatorCallback::computeError: H and the points converted to float, then
 *                      ww = 1/(H6 x + H7 y + 1); dx = (H0 x + H1 y + H2) ww - x'; err = dx dx + dy dy, float arithmetic
 *   kind 2 PNP         PnPRansacCallback::computeError: projectPoints in double (no distortion), the projection and
 *                      the image point as Point2f, err = |diff|^2 accumulated in float; model = R (row-major 9) | t (3)
 * Products and sums are NOT contracted into FMAs (a baseline x86-64 OpenCV build has none).
 */
#pragma GCC optimize("fp-contract=off")
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int fcmp(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

float oracle_score_one(int kind, const double* a, const double* b, const double* M, const double* K) {
    if (kind == 0) {
        double x1[3] = {a[0], a[1], 1.0}, x2[3] = {b[0], b[1], 1.0};
        if (K) {
            x1[0] = (a[0] - K[2]) / K[0]; x1[1] = (a[1] - K[3]) / K[1];
            x2[0] = (b[0] - K[2]) / K[0]; x2[1] = (b[1] - K[3]) / K[1];
        }
        double Ex1[3], Etx2[3];
        for (int r = 0; r < 3; ++r) {
            Ex1[r] = M[3 * r] * x1[0] + M[3 * r + 1] * x1[1] + M[3 * r + 2] * x1[2];
            Etx2[r] = M[r] * x2[0] + M[3 + r] * x2[1] + M[6 + r] * x2[2];
        }
        const double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
        const double d = Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1] + Etx2[0] * Etx2[0] + Etx2[1] * Etx2[1];
        return (float)(x2tEx1 * x2tEx1 / d);
    }
    if (kind == 1) {
        const float H[8] = {(float)M[0], (float)M[1], (float)M[2], (float)M[3], (float)M[4], (float)M[5], (float)M[6], (float)M[7]};
        const float x = (float)a[0], y = (float)a[1], mx = (float)b[0], my = (float)b[1];
        const float ww = 1.f / (H[6] * x + H[7] * y + 1.f);
        const float dx = (H[0] * x + H[1] * y + H[2]) * ww - mx;
        const float dy = (H[3] * x + H[4] * y + H[5]) * ww - my;
        return dx * dx + dy * dy;
    }
    {
        const double X = M[0] * a[0] + M[1] * a[1] + M[2] * a[2] + M[9];
        const double Y = M[3] * a[0] + M[4] * a[1] + M[5] * a[2] + M[10];
        double Z = M[6] * a[0] + M[7] * a[1] + M[8] * a[2] + M[11];
        Z = Z ? 1.0 / Z : 1.0; /* cvProjectPoints2: z = z ? 1./z : 1 */
        const float u = (float)(X * Z * K[0] + K[2]), v = (float)(Y * Z * K[1] + K[3]);
        const float dx = (float)b[0] - u, dy = (float)b[1] - v;
        return dx * dx + dy * dy;
    }
}

/* a: n x 2 (kinds 0, 1) or n x 3 object points (kind 2); b: n x 2; models: nm x 9 (kinds 0, 1) or nm x 12 (kind 2).
 * errors (optional): nm x n. counts[m] = #{err <= threshold}; medians[m] = LMedS median
 * (sorted, odd: the middle, even: the mean of the two middle values; NaN when n == 0). */
void oracle_score_hypotheses(int kind, int n, const double* a, const double* b, int nm, const double* models, const double* K,
                             float threshold, float* errors, int32_t* counts, float* medians) {
    const int ma = kind == 2 ? 3 : 2, mm = kind == 2 ? 12 : 9;
    float* e = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int m = 0; m < nm; ++m) {
        int32_t c = 0;
        for (int i = 0; i < n; ++i) {
            e[i] = oracle_score_one(kind, a + (size_t)ma * i, b + 2 * (size_t)i, models + (size_t)mm * m, K);
            c += e[i] <= threshold;
        }
        if (errors) memcpy(errors + (size_t)m * n, e, sizeof(float) * (size_t)n);
        if (counts) counts[m] = c;
        if (medians) {
            qsort(e, (size_t)n, sizeof(float), fcmp);
            medians[m] = n == 0 ? NAN : (n % 2 ? e[n / 2] : (e[n / 2 - 1] + e[n / 2]) * 0.5f);
        }
    }
    free(e);
}
