// score.hip — batch scoring of pose hypotheses for the robust estimators of eacham's two-view / PnP stage
// (SURVEY.md §8(f) rank 3), gfx950.
//
//   cv::findEssentialMat(..., cv::LMEDS, 0.99, 4.0, 1000, mask)      /root/reference/modules/sfm/reconstruction/ReconstructionManager.cpp:57-61
//   cv::findHomography(pts1, pts2, cv::LMEDS, 4.0, mask2, 100, 0.999)  :75
//   cv::solvePnPRansac(..., 10000, 4.0f, 0.999f, inliers, SOLVEPNP_EPNP)  :227-228   (10 000 hypotheses x every correspondence)
//
// What runs here is the part of those estimators that is data-parallel over (hypothesis, correspondence): the
// error each model assigns to each point — OpenCV 4.5.5's EMEstimatorCallback / HomographyEstimatorCallback /
// PnPRansacCallback ::computeError (the tests hold a CPU restatement of the same formulas) —, the inlier count under a threshold
// (RANSAC) and the median (LMedS). Drawing minimal samples and solving them stays with the caller.
// Layout: correspondences resident once (n x 2 / n x 3 doubles), models nm x 9 / nm x 12 doubles; one workgroup per
// model sweeps the points (coalesced), keeps its n errors as ordered-uint keys in LDS (n <= 16384) or re-reads them
// from the optional error matrix, counts inliers with a fixed-order block sum and finds the median by a 4-pass
// 8-bit radix select — no sort, no atomics on data, bit-identical with the oracle (no FMA contraction: every
// product and sum is an explicit round-to-nearest intrinsic, as a baseline x86-64 OpenCV build computes them).
// HBM-bound in principle (n x 32 B read per model from L2), latency-bound at the sizes the reference has
// (<= 15 000 matches): one launch scores all 10 000 PnP hypotheses.
#include "context.hpp"

#include <cmath>
#include <vector>

// HIP's __fmul_rn / __dadd_rn are plain operators to the compiler, and hipcc contracts a * b + c into an FMA by
// default: this file is compiled with -ffp-contract=off (csrc/Makefile); the reference arithmetic has no fused
// operations (the GPU parity tests are bit-exact, so dropping the flag shows at once).

namespace eacham {
namespace {

constexpr int SC_BLOCK = 256;
constexpr int SC_MAX_LDS = 16384;  // errors of one model kept in LDS (64 KB); larger n re-reads the error matrix

__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }

template <int KIND>
__device__ __forceinline__ float score_one(const double* __restrict__ a, const double* __restrict__ b, const double* M,
                                           const double* K, bool normalise) {
    if (KIND == 0) {
        double x1[3] = {a[0], a[1], 1.0}, x2[3] = {b[0], b[1], 1.0};
        if (normalise) {
            x1[0] = __ddiv_rn(dadd(a[0], -K[2]), K[0]); x1[1] = __ddiv_rn(dadd(a[1], -K[3]), K[1]);
            x2[0] = __ddiv_rn(dadd(b[0], -K[2]), K[0]); x2[1] = __ddiv_rn(dadd(b[1], -K[3]), K[1]);
        }
        double Ex1[3], Etx2[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Ex1[r] = dadd(dadd(dmul(M[3 * r], x1[0]), dmul(M[3 * r + 1], x1[1])), dmul(M[3 * r + 2], x1[2]));
            Etx2[r] = dadd(dadd(dmul(M[r], x2[0]), dmul(M[3 + r], x2[1])), dmul(M[6 + r], x2[2]));
        }
        const double x2tEx1 = dadd(dadd(dmul(x2[0], Ex1[0]), dmul(x2[1], Ex1[1])), dmul(x2[2], Ex1[2]));
        const double d = dadd(dadd(dadd(dmul(Ex1[0], Ex1[0]), dmul(Ex1[1], Ex1[1])), dmul(Etx2[0], Etx2[0])), dmul(Etx2[1], Etx2[1]));
        return (float)__ddiv_rn(dmul(x2tEx1, x2tEx1), d);
    } else if (KIND == 1) {
        const float x = (float)a[0], y = (float)a[1], mx = (float)b[0], my = (float)b[1];
        const float H0 = (float)M[0], H1 = (float)M[1], H2 = (float)M[2], H3 = (float)M[3], H4 = (float)M[4], H5 = (float)M[5],
                    H6 = (float)M[6], H7 = (float)M[7];
        const float ww = __fdiv_rn(1.f, fadd(fadd(fmul(H6, x), fmul(H7, y)), 1.f));
        const float dx = fadd(fmul(fadd(fadd(fmul(H0, x), fmul(H1, y)), H2), ww), -mx);
        const float dy = fadd(fmul(fadd(fadd(fmul(H3, x), fmul(H4, y)), H5), ww), -my);
        return fadd(fmul(dx, dx), fmul(dy, dy));
    } else {
        const double X = dadd(dadd(dadd(dmul(M[0], a[0]), dmul(M[1], a[1])), dmul(M[2], a[2])), M[9]);
        const double Y = dadd(dadd(dadd(dmul(M[3], a[0]), dmul(M[4], a[1])), dmul(M[5], a[2])), M[10]);
        double Z = dadd(dadd(dadd(dmul(M[6], a[0]), dmul(M[7], a[1])), dmul(M[8], a[2])), M[11]);
        Z = Z != 0.0 ? __ddiv_rn(1.0, Z) : 1.0;
        const float u = (float)dadd(dmul(dmul(X, Z), K[0]), K[2]), v = (float)dadd(dmul(dmul(Y, Z), K[1]), K[3]);
        const float dx = fadd((float)b[0], -u), dy = fadd((float)b[1], -v);
        return fadd(fmul(dx, dx), fmul(dy, dy));
    }
}

// total order of floats as unsigned keys (negatives reversed, NaN with the sign bit clear sorts last)
__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// k-th smallest (0-based) of the n keys: 4 passes of an 8-bit radix histogram, block-wide. `load(i)` returns key i.
template <class Load>
__device__ unsigned radix_select(Load load, int n, int k, unsigned* hist /* [256] LDS */, unsigned* sh /* [2] LDS */) {
    unsigned prefix = 0, mask = 0;
    int want = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += SC_BLOCK) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += SC_BLOCK) {
            const unsigned key = load(i);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);  // integer counts: order-free
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0, d = 0;
            for (; d < 255; ++d) {
                if (acc + (int)hist[d] > want) break;
                acc += (int)hist[d];
            }
            sh[0] = (unsigned)d;
            sh[1] = (unsigned)acc;
        }
        __syncthreads();
        prefix |= sh[0] << shift;
        mask |= 255u << shift;
        want -= (int)sh[1];
        __syncthreads();
    }
    return prefix;
}

template <int KIND>
__global__ __launch_bounds__(SC_BLOCK) void score_kernel(int n, const double* __restrict__ a, const double* __restrict__ b,
                                                         const double* __restrict__ models, const double* __restrict__ Kdev,
                                                         int normalise, float threshold, float* __restrict__ errors,
                                                         int* __restrict__ counts, float* __restrict__ medians, int keys_in_lds) {
    extern __shared__ unsigned keys[];  // [n] when keys_in_lds
    __shared__ unsigned hist[256], sh[2];
    __shared__ int wsum[SC_BLOCK / 64];
    constexpr int MA = KIND == 2 ? 3 : 2, MM = KIND == 2 ? 12 : 9;
    const int m = blockIdx.x;
    double M[MM], K[4] = {1, 1, 0, 0};
#pragma unroll
    for (int k = 0; k < MM; ++k) M[k] = models[(size_t)MM * m + k];
    if (Kdev) {
#pragma unroll
        for (int k = 0; k < 4; ++k) K[k] = Kdev[k];
    }
    float* erow = errors ? errors + (size_t)m * n : nullptr;
    int c = 0;
    for (int i = threadIdx.x; i < n; i += SC_BLOCK) {
        double pa[MA], pb[2] = {b[2 * (size_t)i], b[2 * (size_t)i + 1]};
#pragma unroll
        for (int k = 0; k < MA; ++k) pa[k] = a[(size_t)MA * i + k];
        const float e = score_one<KIND>(pa, pb, M, K, normalise != 0);
        c += e <= threshold;
        if (erow) erow[i] = e;
        if (keys_in_lds) keys[i] = fkey(e);
    }
    // inlier count: shuffle tree per wave, then the waves in order (integers: exact in any order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && counts) {
        int tot = 0;
        for (int w = 0; w < SC_BLOCK / 64; ++w) tot += wsum[w];
        counts[m] = tot;
    }
    if (!medians) return;
    if (n == 0) {
        if (threadIdx.x == 0) medians[m] = __uint_as_float(0x7fc00000u);
        return;
    }
    __syncthreads();  // keys / the error row are complete (the row was written by this workgroup: visible after the barrier)
    auto load = [&](int i) { return keys_in_lds ? keys[i] : fkey(erow[i]); };
    const unsigned hi = radix_select(load, n, n / 2, hist, sh);
    float med = fkey_inv(hi);
    if (n % 2 == 0) {
        const unsigned lo = radix_select(load, n, n / 2 - 1, hist, sh);
        med = fmul(fadd(fkey_inv(lo), med), 0.5f);
    }
    if (threadIdx.x == 0) medians[m] = med;
}

}  // namespace
}  // namespace eacham

using namespace eacham;

extern "C" int eacham_score_hypotheses(eacham_ctx* ctx, int kind, int n_points, const double* a, const double* b, int n_models,
                                       const double* models, const double* K, float threshold, float* errors,
                                       int32_t* inlier_counts, float* medians) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (kind < EACHAM_SCORE_ESSENTIAL || kind > EACHAM_SCORE_PNP || n_points < 0 || n_models < 0)
        return ctx->fail(EACHAM_ERR_INVALID, "score: bad kind or negative size");
    if (n_models == 0) return EACHAM_OK;
    if (!models || (n_points > 0 && (!a || !b)) || (kind == EACHAM_SCORE_PNP && !K))
        return ctx->fail(EACHAM_ERR_INVALID, "score: null array");
    const long long total = (long long)n_points * n_models;
    const bool keys_in_lds = n_points <= SC_MAX_LDS;
    if (!keys_in_lds && medians && total > (1ll << 32)) return ctx->fail(EACHAM_ERR_CAPACITY, "score: error matrix too large");
    EACHAM_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int ma = kind == EACHAM_SCORE_PNP ? 3 : 2, mm = kind == EACHAM_SCORE_PNP ? 12 : 9;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align(off + bytes); return o; };
    const bool need_err = errors != nullptr || (!keys_in_lds && medians != nullptr);
    const size_t o_a = take(sizeof(double) * ma * (size_t)n_points), o_b = take(sizeof(double) * 2 * (size_t)n_points);
    const size_t o_m = take(sizeof(double) * mm * (size_t)n_models), o_K = take(sizeof(double) * 4);
    const size_t o_c = take(sizeof(int) * (size_t)n_models), o_med = take(sizeof(float) * (size_t)n_models);
    const size_t o_e = take(need_err ? sizeof(float) * (size_t)total : 0);
    if (int rc = ensure_io(ctx, off)) return rc;
    if (int rc = ensure_io_host(ctx, o_e)) return rc;   // everything but the error matrix
    char* base = (char*)ctx->io;
    hipStream_t st = ctx->stream;
    IoPack io(ctx, st);
    if (n_points > 0) {
        if (int rc = io.in(o_a, a, sizeof(double) * ma * (size_t)n_points)) return rc;
        if (int rc = io.in(o_b, b, sizeof(double) * 2 * (size_t)n_points)) return rc;
    }
    if (int rc = io.in(o_m, models, sizeof(double) * mm * (size_t)n_models)) return rc;
    if (K)
        if (int rc = io.in(o_K, K, sizeof(double) * 4)) return rc;
    if (int rc = io.flush_in()) return rc;
    const size_t smem = keys_in_lds && medians ? sizeof(unsigned) * (size_t)(n_points > 0 ? n_points : 1) : 0;
    {
        ProfileScope scope(ctx, EACHAM_KERNEL_SCORE);
#define EACHAM_SCORE_LAUNCH(KIND)                                                                                              \
    do {                                                                                                                       \
        if (smem > 48 * 1024)                                                                                                  \
            EACHAM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)score_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        score_kernel<KIND><<<n_models, SC_BLOCK, smem, st>>>(n_points, (const double*)(base + o_a), (const double*)(base + o_b),  \
                                                             (const double*)(base + o_m), K ? (const double*)(base + o_K) : nullptr, \
                                                             K && kind == EACHAM_SCORE_ESSENTIAL ? 1 : 0, threshold,             \
                                                             need_err ? (float*)(base + o_e) : nullptr, (int*)(base + o_c),      \
                                                             medians ? (float*)(base + o_med) : nullptr, keys_in_lds && medians ? 1 : 0); \
    } while (0)
        if (kind == EACHAM_SCORE_ESSENTIAL) EACHAM_SCORE_LAUNCH(0);
        else if (kind == EACHAM_SCORE_HOMOGRAPHY) EACHAM_SCORE_LAUNCH(1);
        else EACHAM_SCORE_LAUNCH(2);
#undef EACHAM_SCORE_LAUNCH
    }
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    if (int rc = io.out(inlier_counts, o_c, sizeof(int) * (size_t)n_models)) return rc;
    if (int rc = io.out(medians, o_med, sizeof(float) * (size_t)n_models)) return rc;
    if (total > 0)
        if (int rc = io.out(errors, o_e, sizeof(float) * (size_t)total)) return rc;
    if (int rc = io.finish()) return rc;
    return EACHAM_OK;
}
