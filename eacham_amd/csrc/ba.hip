// ba.hip — Levenberg-Marquardt bundle adjustment for gfx950 (MI355X), all fp64.
//
// Replaces the GTSAM part of   RefineBA   modules/sfm/reconstruction/BundleAdjuster.cpp:40-250
//   factors + noise models (:57-121, :171-178), LevenbergMarquardtOptimizer::optimize (:182-216),
//   graph.error (:218-219). Semantics follow SURVEY.md Appendix A (GTSAM 4.1.1, Ceres-default LM,
//   first-order Pose3 chart with the Cayley map on Rot3).
//
// Layout in HBM (everything resident for the whole LM loop; only scalars cross PCIe per try):
//   cameras   pose[nc][12] = R (camera->world, row-major) | t          prior means pose0
//   points    pt[nl][3], prior means pt0, lmprior[nl][2] = {sigma, huber k}
//   obs       grouped by landmark (CSR lm_ptr): obs_cam[no], obs_uv[no][2]; a second CSR
//             (cam_ptr, cam_obs) lists each camera's observations, with camera-ordered copies of their
//             landmark ids / measurements (cam_lm, cam_uv) and the inverse map obs_pos; the per-observation
//             blocks E and Et are stored in CAMERA order (a camera's records are contiguous)
//   pairs     for every camera block (c <= c') the list of observation pairs (o, o') of one landmark
//             seen by both: the sparsity structure of the Schur complement, built once per problem
// Linear algebra of one damped step (H + lambda diag(clamp(diag H))) delta = g:
//   landmarks are eliminated first (3x3 blocks, one thread each), the reduced camera system
//   S (n = 6 nc + 5, the 5 shared-calibration columns are a dense border) is assembled WITHOUT
//   atomics — every block is a deterministic sequential sum over its pair list — into 64x64 TILES of the symbolic
//   pattern of its Cholesky factor under a fill-reducing camera ordering (ba_plan.hpp: what GTSAM's multifrontal solver
//   with COLAMD does for the reference, BundleAdjuster.cpp:182-190), factorised level by level of the elimination tree
//   on v_mfma_f64_16x16x4_f64 (64 columns per panel, the two 32x32 diagonal blocks of a panel factorised by one wave
//   each) with the right-hand side carried as a row of the root panel, and the landmark steps follow by back-substitution.
// Roofline: every kernel except the dense factorisation streams observation-sized arrays once
// (HBM-bound); see DESIGN.md for the per-kernel byte counts.
#include "context.hpp"
#include "ba_plan.hpp"
#include "ba_groups.hpp"
#include "ba_window.hpp"
#include "devprim.hpp"

#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace eacham {

// ---- noise parameters with the reference's float arithmetic (BundleAdjuster.cpp:28-33) ----------
static double rot_sigma_host(float deg) { return (double)(deg * 3.141592f / 180.0f); }
struct Noise {
    double pose_sigma[6], fixed_sigma[6];
    double pose_huber, pix_sigma, pix_huber;
    double k_sigma[5];
};
static Noise make_noise() {
    Noise nz;
    for (int k = 0; k < 3; ++k) {
        nz.pose_sigma[k] = rot_sigma_host(45.0f);
        nz.pose_sigma[3 + k] = (double)0.35f;
        nz.fixed_sigma[k] = rot_sigma_host(0.0001f);
        nz.fixed_sigma[3 + k] = (double)0.0001f;
    }
    nz.pose_huber = (double)2.5f;
    nz.pix_sigma = (double)1.5f;
    nz.pix_huber = (double)3.0f;
    const double ks[5] = {25, 25, 0.00001, 0.0001, 0.0001};
    for (int k = 0; k < 5; ++k) nz.k_sigma[k] = ks[k];
    return nz;
}

#ifndef EXP_WD_STOP
#define EXP_WD_STOP 0
#endif
constexpr int TPB = 256;           // threads per block of the streaming kernels
constexpr int PAIR_CHUNK = 256;    // pair-list entries summed by one wave (lane e takes entries e, e+64, ...)
constexpr int NB = 32;             // Cholesky block size
constexpr int PB = 2 * NB;         // columns of a panel: what one workgroup factorises per launch (sp_level)
constexpr int LMLIN = 24;          // per-landmark linearisation: Hll(6) gl(3) ElK(15)
constexpr int CAMLIN = 72;         // per-camera: Hcc(36) HcK(30) gc(6)
constexpr int KLIN = 30;           // HKK(25) gK(5)
constexpr int SCAL = 16;           // scalar block read back per try
constexpr int N_STATUS = 4;        // flags[0..3] status, flags[4 + P] = hand-off flag of panel P (back-substitution)

typedef double mfma_d4 __attribute__((ext_vector_type(4)));  // accumulator of v_mfma_f64_16x16x4_f64: D[(lane >> 4) + 4 reg][lane & 15]

__device__ __forceinline__ void wave_lds_sync() {  // a wave's own LDS region: order its writes before its reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double huber_weight(double n, double k) { return n <= k ? 1.0 : k / n; }
__device__ __forceinline__ double huber_loss(double n, double k) { return n <= k ? 0.5 * n * n : k * (n - 0.5 * k); }
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// GeneralSFMFactor2<Cal3_S2>::evaluateError. pose = R(9) t(3) camera->world.
// Returns false on a cheirality failure: residual and Jacobians are zero.
template <bool JAC>
__device__ __forceinline__ bool reproj(const double* __restrict__ x, const double* __restrict__ l,
                                       const double* __restrict__ K, double mu, double mv, double* r,
                                       double* Jp, double* Jl, double* Jk) {
    const double dx = l[0] - x[9], dy = l[1] - x[10], dz = l[2] - x[11];
    const double qx = x[0] * dx + x[3] * dy + x[6] * dz;
    const double qy = x[1] * dx + x[4] * dy + x[7] * dz;
    const double qz = x[2] * dx + x[5] * dy + x[8] * dz;
    if (qz <= 0.0) {
        r[0] = r[1] = 0.0;
        if (JAC) {
            for (int k = 0; k < 12; ++k) Jp[k] = 0.0;
            for (int k = 0; k < 6; ++k) Jl[k] = 0.0;
            for (int k = 0; k < 10; ++k) Jk[k] = 0.0;
        }
        return false;
    }
    const double d = 1.0 / qz, u = qx * d, v = qy * d;
    const double fx = K[0], fy = K[1], s = K[2];
    r[0] = fx * u + s * v + K[3] - mu;
    r[1] = fy * v + K[4] - mv;
    if (JAC) {
        const double Dn[12] = {u * v, -1 - u * u, v, -d, 0, d * u, 1 + v * v, -u * v, -u, 0, -d, d * v};
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            Jp[j] = fx * Dn[j] + s * Dn[6 + j];
            Jp[6 + j] = fy * Dn[6 + j];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double a0 = d * (x[3 * j] - u * x[3 * j + 2]);
            const double a1 = d * (x[3 * j + 1] - v * x[3 * j + 2]);
            Jl[j] = fx * a0 + s * a1;
            Jl[3 + j] = fy * a1;
        }
        Jk[0] = u; Jk[1] = 0; Jk[2] = v; Jk[3] = 1; Jk[4] = 0;
        Jk[5] = 0; Jk[6] = v; Jk[7] = 0; Jk[8] = 0; Jk[9] = 1;
    }
    return true;
}

// whitened + Huber-reweighted Jacobian factor of one observation (Robust::WhitenSystem)
__device__ __forceinline__ void obs_factor(const double* x, const double* l, const double* K, double mu,
                                           double mv, double sig, double kh, double* Ap, double* Al,
                                           double* Ak, double* b) {
    double r[2];
    reproj<true>(x, l, K, mu, mv, r, Ap, Al, Ak);
    const double e0 = r[0] / sig, e1 = r[1] / sig;
    const double sw = sqrt(huber_weight(sqrt(e0 * e0 + e1 * e1), kh)), sc = sw / sig;
#pragma unroll
    for (int k = 0; k < 12; ++k) Ap[k] *= sc;
#pragma unroll
    for (int k = 0; k < 6; ++k) Al[k] *= sc;
#pragma unroll
    for (int k = 0; k < 10; ++k) Ak[k] *= sc;
    b[0] = -sw * e0;
    b[1] = -sw * e1;
}

// deterministic block-wide sum of NV values per thread: result valid in thread 0 (out[0..NV))
template <int NV>
__device__ __forceinline__ void block_sum(double* v, double* smem /* [TPB/64][NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double x = v[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        v[i] = x;
    }
    if (lane == 0)
        for (int i = 0; i < NV; ++i) smem[wave * NV + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
            for (int w = 0; w < TPB / 64; ++w) s += smem[w * NV + i];
            v[i] = s;
        }
    }
    __syncthreads();
}

// ---- Cayley chart --------------------------------------------------------
__device__ __forceinline__ void cayley(const double* w, double* R) {
    const double x = w[0], y = w[1], z = w[2];
    const double x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
    const double f = 1.0 / (4.0 + x2 + y2 + z2), f2 = 2.0 * f;
    R[0] = (4 + x2 - y2 - z2) * f; R[1] = (xy - 2 * z) * f2;      R[2] = (xz + 2 * y) * f2;
    R[3] = (xy + 2 * z) * f2;      R[4] = (4 - x2 + y2 - z2) * f; R[5] = (yz - 2 * x) * f2;
    R[6] = (xz - 2 * y) * f2;      R[7] = (yz + 2 * x) * f2;      R[8] = (4 - x2 - y2 + z2) * f;
}
// xi = Local(x, prior): chart coordinates of x^-1 * prior
__device__ __forceinline__ void pose_local(const double* x, const double* p, double* xi) {
    double Rd[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Rd[3 * i + j] = x[i] * p[j] + x[3 + i] * p[3 + j] + x[6 + i] * p[6 + j];
    const double s = 2.0 / (1.0 + Rd[0] + Rd[4] + Rd[8]);
    xi[0] = s * (Rd[7] - Rd[5]);
    xi[1] = s * (Rd[2] - Rd[6]);
    xi[2] = s * (Rd[3] - Rd[1]);
    const double dt0 = p[9] - x[9], dt1 = p[10] - x[10], dt2 = p[11] - x[11];
#pragma unroll
    for (int i = 0; i < 3; ++i) xi[3 + i] = x[i] * dt0 + x[3 + i] * dt1 + x[6 + i] * dt2;
}

// ---- device-side view of a prepared problem --------------------------------------------------------
struct BaDev {
    int nc, nl, no, n;          // n = 6 nc + 5
    // the reduced system in tiles (ba_plan.hpp): camera c starts at padded column sp_pos[c], K at sp_posK, the
    // right-hand side is row sp_rhs_row (row 63 of the root panel); sp_tile_map[I (I + 1) / 2 + J] = tile of (I >= J)
    int sp_npan, sp_ntiles, sp_posK, sp_rhs_row, sp_n_pad;
    const int *sp_pos, *sp_tile_map, *sp_pad_cols, *sp_diag_tile;
    double *T, *Xrow, *zsol;    // tiles [sp_ntiles][64][64]; X of every panel row-major; z of the back-substitution
    // values
    double *pose, *pose0, *pose_new, *pt, *pt0, *pt_new, *Kc, *K0, *K_new;  // Kc: fx fy s u0 v0
    const int* fixed;
    const double* lmprior;  // [nl][2] sigma, k
    // structure
    const int *lm_ptr, *cam_ptr, *cam_obs;
    const int2* cam_chunks; // camera-aligned chunks of <= TPB positions {first, count}; cam_chunk_ptr[c] .. [c+1] = camera c's
    const int* cam_chunk_ptr;
    int n_cam_chunks;
    const int* cam_lm;      // landmark of cam_obs[p], camera order: one hop less in the per-camera gathers
    const int* pos_cam;     // camera of position p (camera order)
    int store_E;            // 1: the linearisation keeps E (DogLeg reads it); 0: the E -> Et kernel recomputes it
    const int* obs_pos;     // inverse of cam_obs: Et is stored in CAMERA order (record of observation o at obs_pos[o])
    const double* cam_uv;   // measurement of cam_obs[p], camera order
    const unsigned *obs_cam, *obs_lm;
    const double* obs_uv;
    const int2* pair_entries;
    const int4* pair_chunks;  // {block id, first entry, count, chunk index within block}
    const int4* blocks;       // {c, c', first chunk, n chunks}
    int n_chunks, n_blocks;
    // the landmark-major form of the Schur stage (ba_groups.hpp; g_rows == 0: the pair lists above serve)
    int g_rows, g_ngroups, g_nblk, g_nparts, g_nchunks, g_nlong;
    long long g_nent4;
    const BaGroup* g_groups;
    const int *g_lmid, *g_lmrow;    // [group][rows / 4]: landmark t of the group (-1: none), local row of its own row
    const int2* g_rowinfo;          // [group][rows]: {camera (nc: a landmark's own row, -1: no row), landmark index inside the group}
    const double* g_uv;             // [group][rows][2]
    const BaChunk* g_chunks;        // {first uint4-row of the chunk's entries, steps of 4 entries}
    const uint32_t* g_ent;          // [uint4-row][lane][4]: r1 | r2 << 16, local rows (null row = g_rows)
    const uint32_t* g_laneinfo;     // [chunk][lane]: lanes after this one in its segment << 28 | (slot + 1 at a segment's first lane)
    const int4* g_blk;              // {c1, c2 (nc = the calibration / right-hand-side pseudo-camera), first slot, count}
    const int* g_longblk;           // blocks with more than GRP_LONG partials
    // the dense form of the Schur stage (ba_window.hpp; w_rows == 0: not in use): padded per-group arrays as above, dense partials
    int w_rows, w_ngroups, w_stride;
    const int2* w_groups;           // {landmarks, rows}
    const int2* w_rowinfo;
    const double* w_uv;
    const int *w_lmid, *w_lmrow;
    double* w_part;                 // [w_ngroups][w_stride]
    // linearisation
    double *E, *lmlin, *camlin, *klin;
    // per try
    double *Et, *lmtry, *Winv, *Wops, *partial, *kk_part, *delta_c, *delta_l, *err_part, *lin_part, *scal;
    // DogLeg: Gauss-Newton step (cameras+K | landmarks) and per-block partial sums of the six forms
    double *dl_nc, *dl_nl, *dl_part;
    double* bpart;  // [n_cam_chunks][36] partial border sums
    // PCG + block-Jacobi (use_preconditioner): x = delta_c | delta_l; r, z, p, q over [cameras + K | landmarks];
    // inverse diagonal blocks Mc (36 per camera), MK (25), Ml (9 per landmark); damping diagonals Dc, Dl;
    // pcg_p1 [n_lm_blocks][6], pcg_p2 [n_cam_chunks][6], pcg_p3 [n_lm_blocks + 1] partial sums;
    // pcg_s = {gamma, threshold, alpha, beta, done, iterations, p.q}
    double *pcg_rc, *pcg_rl, *pcg_zc, *pcg_zl, *pcg_pc, *pcg_pl, *pcg_qc, *pcg_ql, *pcg_Mc, *pcg_MK, *pcg_Ml, *pcg_Dc, *pcg_Dl,
        *pcg_p1, *pcg_p2, *pcg_p3, *pcg_s;
    int* flags;
    double* scal_pinned;  // device view of the pinned host copy of scal[0..2] (final_sums)
    int* sync_counter;    // ticket counter of the "last workgroup makes the final sums" hand-over (self-resetting)
    int n_lm_blocks;  // grid of the per-landmark kernels
    // Small problems (a local window: 1.5 k landmarks = six workgroups) are bound by the LENGTH of a thread's instruction
    // stream, not by bandwidth: a landmark's ~8 observations cost ~3 k fp64 instructions in one lane, ~10 us per kernel
    // whatever the landmark count. There the three kernels that loop over a landmark's observations spread it over
    // lpl = 8 adjacent lanes (observation o0 + sub, o0 + sub + 8, ...; xor-shuffle sums in a fixed order), on a grid of
    // n_ll_blocks workgroups; large problems use lpl = 2 (50 k landmarks are 782 waves on 1024 SIMDs with one lane each).
    int lpl, n_ll_blocks;
    // the same idea for the kernels of the step's tail (back-substitution + error: two passes of dependent gathers per
    // observation): their own lanes-per-landmark count and grid (measured on S200 / config 4: 2 lanes 35.6 / 72 us,
    // 1 lane 41.7 / 75, 4 lanes 41.8 / 71, 8 lanes 55 / 92)
    int lpl_step, n_step_blocks;
    Noise nz;
};

// sum over the LPL adjacent lanes that share a landmark, the same value in all of them (fixed butterfly order)
template <int LPL>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int m = 1; m < LPL; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// ---- K-A: per-landmark linearisation (thread = landmark) --------------------------------------------
// Hll, gl, ElK of the landmark (+ its prior) and E_o = Ap^T Al (6x3) of each of its observations.
template <int LPL>
__device__ __forceinline__ void linearize_landmark(const BaDev& D, int gid) {
    const int j = gid / LPL, sub = gid % LPL;
    if (j >= D.nl) return;  // (the LPL lanes of a landmark leave together)
    const int o0 = D.lm_ptr[j], o1 = D.lm_ptr[j + 1];
    double* out = D.lmlin + (size_t)LMLIN * j;
    double H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, EK[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) EK[k] = 0.0;
    const double l[3] = {D.pt[3 * j], D.pt[3 * j + 1], D.pt[3 * j + 2]};
    double K[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) K[k] = D.Kc[k];
    for (int o = o0 + sub; o < o1; o += LPL) {
        const double* x = D.pose + 12 * (size_t)D.obs_cam[o];
        double xr[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) xr[k] = x[k];
        double Ap[12], Al[6], Ak[10], b[2];
        obs_factor(xr, l, K, D.obs_uv[2 * (size_t)o], D.obs_uv[2 * (size_t)o + 1], D.nz.pix_sigma, D.nz.pix_huber, Ap, Al, Ak, b);
        H[0] += Al[0] * Al[0] + Al[3] * Al[3];
        H[1] += Al[0] * Al[1] + Al[3] * Al[4];
        H[2] += Al[0] * Al[2] + Al[3] * Al[5];
        H[3] += Al[1] * Al[1] + Al[4] * Al[4];
        H[4] += Al[1] * Al[2] + Al[4] * Al[5];
        H[5] += Al[2] * Al[2] + Al[5] * Al[5];
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] += Al[a] * b[0] + Al[3 + a] * b[1];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) EK[3 * a + c] += Ak[a] * Al[c] + Ak[5 + a] * Al[3 + c];
        if (D.store_E) {
            double* E = D.E + 18 * (size_t)D.obs_pos[o];  // camera order
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) E[3 * a + c] = Ap[a] * Al[c] + Ap[6 + a] * Al[3 + c];
        }
    }
    if (LPL > 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) H[k] = group_sum<LPL>(H[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = group_sum<LPL>(g[k]);
#pragma unroll
        for (int k = 0; k < 15; ++k) EK[k] = group_sum<LPL>(EK[k]);
        if (sub != 0) return;
    }
    if (o1 > o0) {  // PriorFactor<Point3>, Robust(Huber(3/obs), Isotropic(1/obs))
        const double sg = D.lmprior[2 * j], kh = D.lmprior[2 * j + 1];
        double e[3], n2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            e[a] = (l[a] - D.pt0[3 * j + a]) / sg;
            n2 += e[a] * e[a];
        }
        const double sw = sqrt(huber_weight(sqrt(n2), kh)), w = sw / sg;
        H[0] += w * w; H[3] += w * w; H[5] += w * w;
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] += w * (-sw * e[a]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] = H[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[6 + k] = g[k];
#pragma unroll
    for (int k = 0; k < 15; ++k) out[9 + k] = EK[k];
}

// ---- K-B: per-camera linearisation (block = one of LSEG segments of a camera's observations) ----------
// Every observation contributes two whitened Jacobian rows r = [Ap (6) | Ak (5) | b] (12 entries) and the camera's
// blocks are the 12 x 12 Gram matrix sum_rows r^T r: Hcc = [0:6, 0:6], HcK = [0:6, 6:11], gc = [0:6, 11],
// the camera's share of HKK = [6:11, 6:11] and of gK = [6:11, 11]. The sum over observations is the K dimension of
// v_mfma_f64_16x16x4_f64 (D += R^T R, four rows per instruction; the same register is the A and the B operand), so
// there is no block-wide reduction of 77 per-thread sums any more — those shuffle trees, not the arithmetic or the
// gathers, were the kernel's time (35 us on S200 with one block per camera, more with four). A wave stages the
// rows of 32 observations at a time in its own LDS region as 64 rows x 16 doubles (padding columns zero): the
// operand of step t is then lds[64 t + lane], one conflict-free 512-byte read.
// -> clpart[(c * LSEG + seg)][256] = the block's 16 x 16 accumulator (waves added in order).
constexpr int LSEG = 4;     // segments per camera
constexpr int CLP = 256;    // 16 x 16 accumulator image per (camera, segment)
__device__ __forceinline__ void linearize_camera_segment(const BaDev& D, double* __restrict__ clpart, int block) {
    __shared__ __attribute__((aligned(16))) double stage[TPB / 64][64 * 16];  // 8 KB per wave
    __shared__ double wsum[TPB / 64][4][64];
    const int c = block / LSEG, seg = block % LSEG;
    const int q0c = D.cam_ptr[c], q1c = D.cam_ptr[c + 1];
    const int len = (q1c - q0c + LSEG - 1) / LSEG;
    const int p0 = q0c + seg * len, p1 = min(p0 + len, q1c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* st = stage[wave];
    double x[12], K[5];
#pragma unroll
    for (int k = 0; k < 12; ++k) x[k] = D.pose[12 * (size_t)c + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) K[k] = D.Kc[k];
    mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int base = p0; base < p1; base += TPB) {  // block-uniform trip count
        const int p = base + (int)threadIdx.x;
        const bool valid = p < p1;
        double row[2][12];
        {
            const int lm = valid ? D.cam_lm[p] : 0;
            const double2 uv = valid ? *reinterpret_cast<const double2*>(&D.cam_uv[2 * (size_t)p]) : double2{0.0, 0.0};
            const double* lp = D.pt + 3 * (size_t)lm;
            const double l[3] = {lp[0], lp[1], lp[2]};
            double Ap[12], Al[6], Ak[10], b[2];
            obs_factor(x, l, K, uv.x, uv.y, D.nz.pix_sigma, D.nz.pix_huber, Ap, Al, Ak, b);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 6; ++k) row[h][k] = valid ? Ap[6 * h + k] : 0.0;
#pragma unroll
                for (int k = 0; k < 5; ++k) row[h][6 + k] = valid ? Ak[5 * h + k] : 0.0;
                row[h][11] = valid ? b[h] : 0.0;
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((lane >> 5) == half) {
                // rows 2 (lane & 31) and + 1; the 16-byte slots of a row are XOR-swizzled with the observation's low
                // bits so that the eight lanes of a store group hit eight different slots (stride 256 bytes otherwise)
                const int o = lane & 31;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double2* dst = reinterpret_cast<double2*>(st + (2 * o + h) * 16);
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl)
                        dst[sl ^ (o & 7)] = sl < 6 ? double2{row[h][2 * sl], row[h][2 * sl + 1]} : double2{0.0, 0.0};
                }
            }
            wave_lds_sync();
            {   // operand of step t: lane (i = lane & 15, kk = lane >> 4) <- row 4 t + kk, entry i
                const int i = lane & 15, kk = lane >> 4;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int r = 4 * t + kk, o = r >> 1;
                    const double v = st[r * 16 + ((((i >> 1) ^ (o & 7)) << 1) | (i & 1))];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
                }
            }
            wave_lds_sync();
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) wsum[wave][r][lane] = acc[r];
    __syncthreads();
    {   // thread -> (reg, lane) of the accumulator image: D[(lane >> 4) + 4 reg][lane & 15]
        const int r = threadIdx.x >> 6, l = threadIdx.x & 63;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) v += wsum[w][r][l];
        clpart[(size_t)CLP * block + 16 * ((l >> 4) + 4 * r) + (l & 15)] = v;
    }
}

// K-A and K-B in ONE launch: the landmark side and the camera side of the linearisation read the same values and
// write disjoint outputs; as two launches the second waited for the first (18 + 23 us on S200). The camera segments
// (the longer role) take the low workgroup ids.
template <int LPL>
__global__ __launch_bounds__(TPB) void ba_linearize(BaDev D, double* __restrict__ clpart, int n_cam_blocks) {
    if ((int)blockIdx.x < n_cam_blocks) linearize_camera_segment(D, clpart, (int)blockIdx.x);
    else linearize_landmark<LPL>(D, ((int)blockIdx.x - n_cam_blocks) * TPB + threadIdx.x);
}

// Second stage, one launch: blocks [0, KLIN) sum the calibration parts over every (camera, segment) in fixed order
// (strided partial sums, then a shuffle tree) and add the Cal3_S2 prior -> klin; blocks [KLIN, KLIN + nc) add a
// camera's segments in order and its pose prior -> camlin.
__device__ __forceinline__ void finish_linearize_block(const BaDev& D, const double* __restrict__ clpart, int block, int lane) {
    if (block < KLIN) {
        const int i = block;
        const int src = i < 25 ? 16 * (6 + min(i / 5, i % 5)) + 6 + max(i / 5, i % 5) : 16 * (6 + (i - 25)) + 11;  // upper triangle
        double s = 0.0;
        for (int e = lane; e < D.nc * LSEG; e += 64) s += clpart[(size_t)CLP * e + src];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) {
            if (i < 25) {
                if (i / 5 == i % 5) s += 1.0 / (D.nz.k_sigma[i / 5] * D.nz.k_sigma[i / 5]);
            } else {
                const int a = i - 25;
                s += (1.0 / D.nz.k_sigma[a]) * (-(D.Kc[a] - D.K0[a]) / D.nz.k_sigma[a]);
            }
            D.klin[i] = s;
        }
        return;
    }
    const int c = block - KLIN;
    if (lane >= 57) return;
    int a, bb;  // entry of the accumulator image this lane finishes: Hcc upper triangle (21), HcK (30), gc (6)
    if (lane < 21) {
        a = 0;
        int rem = lane;
        while (rem >= 6 - a) rem -= 6 - a, ++a;
        bb = a + rem;
    } else if (lane < 51) {
        a = (lane - 21) / 5, bb = 6 + (lane - 21) % 5;
    } else {
        a = lane - 51, bb = 11;
    }
    double v = 0.0;
    for (int seg = 0; seg < LSEG; ++seg) v += clpart[(size_t)CLP * (c * LSEG + seg) + 16 * a + bb];
    // PriorFactor<Pose3>: e = -Local(x, prior), H = I; Robust(Huber 2.5) unless the node is fixed
    double x[12], xi[6], n2 = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) x[k] = D.pose[12 * (size_t)c + k];
    pose_local(x, D.pose0 + 12 * (size_t)c, xi);
    const bool fx = D.fixed[c] != 0;
    const double* sg = fx ? D.nz.fixed_sigma : D.nz.pose_sigma;
#pragma unroll
    for (int k = 0; k < 6; ++k) n2 += (xi[k] / sg[k]) * (xi[k] / sg[k]);
    const double sw = fx ? 1.0 : sqrt(huber_weight(sqrt(n2), D.nz.pose_huber));
    double* out = D.camlin + (size_t)CAMLIN * c;
    if (lane < 21) {
        if (a == bb) v += (sw / sg[a]) * (sw / sg[a]);
        out[6 * a + bb] = v;
        out[6 * bb + a] = v;
    } else if (lane < 51) {
        out[36 + (lane - 21)] = v;
    } else {
        out[66 + a] = v + (sw / sg[a]) * (-sw * (-xi[a] / sg[a]));
    }
}

__global__ __launch_bounds__(64) void ba_finish_linearize(BaDev D, const double* __restrict__ clpart) {
    finish_linearize_block(D, clpart, (int)blockIdx.x, (int)threadIdx.x);
}

// ---- K-C: per-landmark elimination for one lambda (thread = landmark) -------------------------------
// Hd = Hll + lambda clamp(diag), Hd = L L^T, Linv; EKt = ElK Linv^T, gt = Linv gl (Et_o = E_o Linv^T: K-C2).
// Also this block's share of the (K,K) Schur term: sum EKt EKt^T (25) and EKt gt (5) -> kk_part.
// The launch also clears the tiles of S for the assembly kernels behind it (the factorisation works in place, so S is
// rebuilt for every lambda): the stores are issued first and drain under the arithmetic — no fill node per try.
// The FIRST try after a linearisation also carries that linearisation's second stage (finish_linearize_block: n_finish
// extra workgroups, one wave of each at work): it depends on the linearisation only, like this kernel, and as a launch of its own
// it was 4.6 (window) / 6.9 us (S200) on the chain of the iteration.
// the 3x3 elimination of one landmark for one lambda: Linv (lower: m00; m10 m11; m20 m21 m22), gt = Linv gl, EKt[a] = Linv ElK[a].
// ONE definition for every kernel that needs it (the thread that owns the landmark, and — in the pair-list form — every observation
// of it, which recomputes these 24 numbers instead of waiting for a launch that stores them): same instructions, same bits.
struct LmElim {
    double m[6], gt[3], ek[15];
    bool ok;
};
__device__ __forceinline__ LmElim eliminate_landmark(const double* __restrict__ in /* lmlin record */, double lambda) {
    LmElim r;
    double h0 = in[0], h1 = in[1], h2 = in[2], h3 = in[3], h4 = in[4], h5 = in[5];
    h0 += lambda * clampd(h0, 1e-6, 1e32);
    h3 += lambda * clampd(h3, 1e-6, 1e32);
    h5 += lambda * clampd(h5, 1e-6, 1e32);
    // 3x3 Cholesky, lower: [l00; l10 l11; l20 l21 l22]
    bool ok = h0 > 0.0;
    const double l00 = sqrt(ok ? h0 : 1.0);
    const double l10 = h1 / l00, l20 = h2 / l00;
    const double d1 = h3 - l10 * l10;
    ok = ok && d1 > 0.0;
    const double l11 = sqrt(d1 > 0.0 ? d1 : 1.0);
    const double l21 = (h4 - l20 * l10) / l11;
    const double d2 = h5 - l20 * l20 - l21 * l21;
    ok = ok && d2 > 0.0;
    const double l22 = sqrt(d2 > 0.0 ? d2 : 1.0);
    r.ok = ok;
    const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;
    const double m10 = -l10 * m00 * m11;
    const double m21 = -l21 * m11 * m22;
    const double m20 = -(l20 * m00 + l21 * m10) * m22;
    r.m[0] = m00; r.m[1] = m10; r.m[2] = m11; r.m[3] = m20; r.m[4] = m21; r.m[5] = m22;
    const double g0 = in[6], g1 = in[7], g2 = in[8];
    r.gt[0] = m00 * g0; r.gt[1] = m10 * g0 + m11 * g1; r.gt[2] = m20 * g0 + m21 * g1 + m22 * g2;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const double e0 = in[9 + 3 * a], e1 = in[10 + 3 * a], e2 = in[11 + 3 * a];
        r.ek[3 * a] = m00 * e0;
        r.ek[3 * a + 1] = m10 * e0 + m11 * e1;
        r.ek[3 * a + 2] = m20 * e0 + m21 * e1 + m22 * e2;
    }
    return r;
}

__device__ __forceinline__ void eliminate_landmarks_block(const BaDev& D, double lambda, int block, int n_lm_blocks) {
    __shared__ double sm[(TPB / 64) * 30];
    const int j = block * TPB + threadIdx.x;
    {
        double2* S2 = reinterpret_cast<double2*>(D.T);
        const size_t total = (size_t)D.sp_ntiles * (PB * PB / 2);
        for (size_t e = (size_t)j; e < total; e += (size_t)n_lm_blocks * TPB) S2[e] = make_double2(0.0, 0.0);
    }
    double kk[30];
#pragma unroll
    for (int k = 0; k < 30; ++k) kk[k] = 0.0;
    if (j < D.nl) {
        const int o0 = D.lm_ptr[j], o1 = D.lm_ptr[j + 1];
        double* out = D.lmtry + (size_t)LMLIN * j;
        if (o1 > o0) {
            const LmElim e = eliminate_landmark(D.lmlin + (size_t)LMLIN * j, lambda);
            if (!e.ok) atomicOr(D.flags, 1);
#pragma unroll
            for (int k = 0; k < 6; ++k) out[k] = e.m[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) out[6 + k] = e.gt[k];
#pragma unroll
            for (int k = 0; k < 15; ++k) out[9 + k] = e.ek[k];
            const double t0 = e.gt[0], t1 = e.gt[1], t2 = e.gt[2];
            const double* ek = e.ek;
#pragma unroll
            for (int a = 0; a < 5; ++a) {
#pragma unroll
                for (int bb = 0; bb < 5; ++bb)
                    kk[5 * a + bb] = ek[3 * a] * ek[3 * bb] + ek[3 * a + 1] * ek[3 * bb + 1] + ek[3 * a + 2] * ek[3 * bb + 2];
                kk[25 + a] = ek[3 * a] * t0 + ek[3 * a + 1] * t1 + ek[3 * a + 2] * t2;
            }
        } else {
#pragma unroll
            for (int k = 0; k < LMLIN; ++k) out[k] = 0.0;
        }
    }
    block_sum<30>(kk, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 30; ++k) D.kk_part[(size_t)30 * block + k] = kk[k];
}

// ---- K-C2: Et_o = E_o Linv^T and the calibration border, fused (thread = observation, camera order) -------------
// Levenberg-Marquardt never reads E after the linearisation, so writing its 144 bytes per observation there and
// reading them back here would be 2 x 72 MB of traffic for ~450 flops per observation: the Jacobians are recomputed
// at the linearisation point (pose / pt / Kc do not change inside the lambda loop) from camera-ordered ids and
// measurements. The same thread holds Et_o and has its landmark's record in reach, so it also forms the
// observation's share of the border, Et_o [gt | EKt^T] (6 x 6 = 36 products), which used to be a second pass over
// Et with the same gathers (ba_border_partials, 36 us on S200). Workgroups are CAMERA-ALIGNED chunks of <= 256
// observations (cam_chunks, built once per problem): a block sum per chunk -> bpart[chunk][36], added per camera in
// chunk order by assemble_border (ba_assemble). Fixed order, no atomics.
__device__ __forceinline__ void eliminate_observations_block(const BaDev& D, double lambda, int block) {
    // The border share of an observation is Et_o (6 x 3) M_o (3 x 6), M_o = [EKt^T | gt]; its sum over the chunk runs
    // on v_mfma_f64_16x16x4_f64 with TWO observations per instruction: rows 0..5 / 8..13 of the A operand hold Et of
    // the even / odd observation of a pair, columns 0..5 / 8..13 of the B operand their M, the K index (3 used of 4)
    // is shared, so the diagonal 6 x 6 blocks of D accumulate Et M of the two observations and the off-diagonal
    // blocks are discarded. A wave stages 16 observations at a time in its own LDS region, pair t as 3 x 16 doubles
    // per operand: the operands of step t are lds[48 t + lane] for the lanes of k = lane >> 4 < 3, zero otherwise.
    __shared__ __attribute__((aligned(16))) double stA[TPB / 64][8 * 48], stB[TPB / 64][8 * 48];  // 2 x 3 KB per wave: 16 observations per round
    __shared__ double wsum[TPB / 64][4][64];
    const int2 ch = D.cam_chunks[block];  // {first position, count}
    const int p = ch.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valid = (int)threadIdx.x < ch.y;
    double et[18], lv[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) et[k] = 0.0, lv[k] = 0.0;
    if (valid) {
        const int lm = D.cam_lm[p];
        const double* x = D.pose + 12 * (size_t)D.pos_cam[p];
        const double* lp = D.pt + 3 * (size_t)lm;
        const double2 uv = *reinterpret_cast<const double2*>(&D.cam_uv[2 * (size_t)p]);
        // Linv, gt, EKt of the observation's landmark: recomputed from its linearisation record (the same 192 bytes the stored
        // result would be, ~100 flops), so that this role does not wait for the landmark role — both run in ONE launch
        const LmElim le = eliminate_landmark(D.lmlin + (size_t)LMLIN * lm, lambda);
        double xr[12], K[5];
#pragma unroll
        for (int k = 0; k < 12; ++k) xr[k] = x[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) K[k] = D.Kc[k];
        const double l[3] = {lp[0], lp[1], lp[2]};
        const double m0 = le.m[0], m1 = le.m[1], m2 = le.m[2], m3 = le.m[3], m4 = le.m[4], m5 = le.m[5];
#pragma unroll
        for (int k = 0; k < 3; ++k) lv[k] = le.gt[k];
#pragma unroll
        for (int k = 0; k < 15; ++k) lv[3 + k] = le.ek[k];
        double Ap[12], Al[6], Ak[10], b[2];
        obs_factor(xr, l, K, uv.x, uv.y, D.nz.pix_sigma, D.nz.pix_huber, Ap, Al, Ak, b);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double e0 = Ap[a] * Al[0] + Ap[6 + a] * Al[3], e1 = Ap[a] * Al[1] + Ap[6 + a] * Al[4], e2 = Ap[a] * Al[2] + Ap[6 + a] * Al[5];
            et[3 * a] = m0 * e0;
            et[3 * a + 1] = m1 * e0 + m2 * e1;
            et[3 * a + 2] = m3 * e0 + m4 * e1 + m5 * e2;
        }
#ifndef EXP_NO_ET_STORE  // (knock-out, timing only: what the 72 MB store costs)
        double2* out = reinterpret_cast<double2*>(D.Et + 18 * (size_t)p);
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = double2{et[2 * k], et[2 * k + 1]};
#endif
    }
    mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
    double *sa = stA[wave], *sb = stB[wave];
#pragma unroll
    for (int round = 0; round < 4; ++round) {  // 16 lanes stage, the whole wave multiplies (small LDS footprint: four workgroups per CU)
        if ((lane >> 4) == round) {
            const int o = lane & 15, t = o >> 1, sub = o & 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // A: Et[a][k] = et[3 a + k], a = 0..5 (+ two zero rows); B: M[k][j] = EKt[j][k] = lv[3 + 3 j + k] (j < 5), gt[k] = lv[k] (j = 5)
                double2* da = reinterpret_cast<double2*>(sa + 48 * t + 16 * k + 8 * sub);
                double2* db = reinterpret_cast<double2*>(sb + 48 * t + 16 * k + 8 * sub);
                da[0] = double2{et[k], et[3 + k]};
                da[1] = double2{et[6 + k], et[9 + k]};
                da[2] = double2{et[12 + k], et[15 + k]};
                da[3] = double2{0.0, 0.0};
                db[0] = double2{lv[3 + k], lv[6 + k]};
                db[1] = double2{lv[9 + k], lv[12 + k]};
                db[2] = double2{lv[15 + k], lv[k]};
                db[3] = double2{0.0, 0.0};
            }
        }
        wave_lds_sync();
        {
            const bool live = lane < 48;  // k = lane >> 4 < 3
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double a = live ? sa[48 * t + lane] : 0.0, b = live ? sb[48 * t + lane] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
        }
        wave_lds_sync();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) wsum[wave][r][lane] = acc[r];
    __syncthreads();
    if (threadIdx.x < 36) {
        // bpart entry k: k < 30 -> (a, bb) = (k / 5, k % 5), else (k - 30, 5); D[a][b] + D[8 + a][8 + b] with
        // D[row][col] in register row >> 2 of lane (col + 16 (row & 3)); the waves are added in order
        const int k = threadIdx.x, a = k < 30 ? k / 5 : k - 30, b = k < 30 ? k % 5 : 5;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w)
            v += wsum[w][a >> 2][b + 16 * (a & 3)] + wsum[w][2 + (a >> 2)][8 + b + 16 * (a & 3)];
        D.bpart[(size_t)36 * block + k] = v;
    }
}

// K-C + K-C2 of the pair-list form in ONE launch (round 5; two launches before: 11.6 + 10.4 us on the chain of a local window's
// try): workgroups [0, n_cam_chunks) take the observation role (the longer one), the next n_lm_blocks the landmark role (lmtry for
// the step's back-substitution, the K-corner partials, the clearing of S), the last n_finish the linearisation's second stage.
__global__ __launch_bounds__(TPB) void ba_eliminate(BaDev D, double lambda, const double* __restrict__ clpart, int n_lm_blocks, int n_finish) {
    const int b = (int)blockIdx.x;  // (every branch below is workgroup-uniform)
    if (b < D.n_cam_chunks) eliminate_observations_block(D, lambda, b);
    else if (b < D.n_cam_chunks + n_lm_blocks) eliminate_landmarks_block(D, lambda, b - D.n_cam_chunks, n_lm_blocks);
    else if (threadIdx.x < 64) finish_linearize_block(D, clpart, b - D.n_cam_chunks - n_lm_blocks, (int)threadIdx.x);
    (void)n_finish;
}

// ---- K-D1: Schur pair products (wave = chunk of <= 256 entries of one camera block's pair list) ------
// lane e computes the 6x6 product Et[o_e] Et[o'_e]^T of its entry. The two 144-byte rows of an entry are NOT gathered
// by the lane that multiplies them (64 lanes x 16 bytes out of 64 different rows per instruction: 64 memory sectors
// per wave-instruction, nine instructions per row): the 128 rows of a group of 64 entries are copied into the wave's
// LDS region by LDS-DMA with per-lane source addresses, 16-byte piece c = 64 it + lane of the row-major image coming
// from row c / 9 — a wave-instruction then covers seven rows end to end (21 sectors) — and every lane reads its two
// rows back from LDS. The 36 sums over the wave's entries go through a wave-private LDS transpose (the same
// region): lane e writes its 36 products as column e, lane v < 36 then adds row v in entry order — 36 writes +
// 64 reads + 64 adds per wave instead of the 36 x 6 x (2 bpermute + add) of a shuffle tree. Fixed order, no atomics.
__global__ __launch_bounds__(TPB) void ba_schur_pairs(BaDev D) {
    constexpr int ROW = 65;  // odd stride in doubles: the row reads of the 36 summing lanes spread over the banks
    constexpr int WBUF = 36 * ROW;  // 2340 doubles per wave: >= the 128 x 18 doubles of a staged group
    static_assert(WBUF >= 128 * 18, "the staging image fits the reduction image");
    __shared__ __attribute__((aligned(16))) double tr[TPB / 64][WBUF];
    // Et is stored in CAMERA order: the two sides of a block's entries walk two cameras' contiguous regions in
    // ascending order (landmark-ordered records put every gather in a different page of 72 MB).
    // Workgroups go to the 8 XCDs round-robin, each XCD has its own 4 MB L2, and the chunk list is ordered by
    // camera block (ci, cj): every XCD takes one contiguous eighth of it, so that the rows Et of the few cameras
    // ci it is working on stay in ITS L2 (dealt round-robin every XCD sees every camera and the 2 x 144-byte
    // gathers of an entry nearly all miss: 430 MB of L2 fills per launch for 72 MB of Et).
    const int per_xcd = gridDim.x / 8;  // the grid is a multiple of 8
    const int chunk = ((blockIdx.x % 8) * per_xcd + blockIdx.x / 8) * (TPB / 64) + (threadIdx.x >> 6);
    if (chunk >= D.n_chunks) return;
    const int lane = threadIdx.x & 63;
    double* buf = tr[threadIdx.x >> 6];
    const int4 ch = D.pair_chunks[chunk];
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
    // groups of 64 entries, in order (a fixed summation order per lane); the copy of group g + 1 is issued as soon as
    // the rows of group g sit in registers and flies while they are multiplied. Measured on S200 (per launch): the
    // per-lane gathers 85 us, this staging 79 us, with the copy in flight under the products 76 us; the same image
    // staged through registers (16-byte loads + ds_write_b128) 81 us, and with the next group's pieces held in
    // registers across the products it spills.
    auto stage = [&](int g) {
        const int e = g + lane;
        const int2 pr = D.pair_entries[ch.y + (e < ch.z ? e : 0)];  // (a dead lane stages rows nobody multiplies)
#pragma unroll
        for (int it = 0; it < 18; ++it) {
            const int c = 64 * it + lane, r = c / 9, piece = c - 9 * r;  // piece of row r = side (r & 1) of entry r >> 1
            const int rx = __shfl(pr.x, r >> 1), ry = __shfl(pr.y, r >> 1);
#ifdef EXP_PAIRS_LOCAL  // (knock-out, timing only: every row comes from a 74 KB window — the gathers as if Et sat in LDS / L1)
            const double* src = D.Et + 18 * (size_t)(((r & 1) ? ry : rx) & 511) + 2 * piece;
#else
            const double* src = D.Et + 18 * (size_t)((r & 1) ? ry : rx) + 2 * piece;
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(buf + 128 * it), 16, 0, 0);
        }
    };
    stage(0);
    for (int g = 0; g < ch.z; g += 64) {
        __builtin_amdgcn_s_waitcnt(0);  // the DMA pieces have landed (vmcnt) before any lane reads them
        wave_lds_sync();
        const double2* xp = reinterpret_cast<const double2*>(buf + 36 * lane);
        double x[18], y[18];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double2 a = xp[k], b = xp[9 + k];
            x[2 * k] = a.x, x[2 * k + 1] = a.y, y[2 * k] = b.x, y[2 * k + 1] = b.y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are in registers: LDS-DMA is not ordered behind ds_read
        wave_lds_sync();
        if (g + 64 < ch.z) stage(g + 64);
        if (g + lane < ch.z) {
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) acc[6 * a + b] += x[3 * a] * y[3 * b] + x[3 * a + 1] * y[3 * b + 1] + x[3 * a + 2] * y[3 * b + 2];
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 36; ++k) buf[k * ROW + lane] = acc[k];
    wave_lds_sync();
    if (lane < 36) {
        const double* rowp = buf + lane * ROW;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four interleaved partial sums, combined in a fixed order
#pragma unroll
        for (int e = 0; e < 64; e += 4) {
            s0 += rowp[e];
            s1 += rowp[e + 1];
            s2 += rowp[e + 2];
            s3 += rowp[e + 3];
        }
        D.partial[(size_t)36 * chunk + lane] = (s0 + s1) + (s2 + s3);
    }
}

// ---- K-D2: assemble the camera blocks of S (thread = block element) ---------------------------------
// Element (r, q) of the permuted, padded matrix (ba_plan.hpp) goes to its tile; only the lower triangle of tiles is
// stored (a call that names an upper tile is dropped: the caller stores both (r, q) and (q, r)), a diagonal tile holds
// both of its triangles.
__device__ __forceinline__ void sp_store(const BaDev& D, int r, int q, double v) {
    const int I = r >> 6, J = q >> 6;
    if (I < J) return;
    D.T[(size_t)D.sp_tile_map[I * (I + 1) / 2 + J] * (PB * PB) + (r & (PB - 1)) * PB + (q & (PB - 1))] = v;
}
__device__ __forceinline__ void assemble_blocks(const BaDev& D, double lambda, unsigned block) {
    const long long idx = (long long)block * TPB + threadIdx.x;
    const int blk = (int)(idx / 36), el = (int)(idx % 36);
    if (blk >= D.n_blocks) return;
    const int4 B = D.blocks[blk];
    const int a = el / 6, b = el % 6;
    double s = 0.0;
    for (int k = 0; k < B.w; ++k) s += D.partial[(size_t)36 * (B.z + k) + el];
    double v = -s;
    if (B.x == B.y) {
        const double h = D.camlin[(size_t)CAMLIN * B.x + 6 * a + b];
        v += h;
        if (a == b) v += lambda * clampd(h, 1e-6, 1e32);
    }
    const int r = D.sp_pos[B.x] + a, q = D.sp_pos[B.y] + b;
    sp_store(D, r, q, v);
    if (B.x != B.y) sp_store(D, q, r, v);
}

// ---- K-D3: the calibration border and the right-hand side -------------------------------------------
// thread = border entry: adds the chunk partials of its camera (written by K-C2) in chunk order and writes S;
// the last block reduces the (K,K) corner.
__device__ __forceinline__ void assemble_border(const BaDev& D, double lambda, int c) {
    if (c < D.nc) {
        if (threadIdx.x < 36) {
            const int k = threadIdx.x;
            double s = 0.0;
            for (int ch = D.cam_chunk_ptr[c]; ch < D.cam_chunk_ptr[c + 1]; ++ch) s += D.bpart[(size_t)36 * ch + k];
            const double* cl = D.camlin + (size_t)CAMLIN * c;
            if (k < 30) {
                const int a = k / 5, bb = k % 5;
                const double v = cl[36 + k] - s;
                sp_store(D, D.sp_pos[c] + a, D.sp_posK + bb, v);
                sp_store(D, D.sp_posK + bb, D.sp_pos[c] + a, v);
            } else {
                sp_store(D, D.sp_rhs_row, D.sp_pos[c] + (k - 30), cl[66 + (k - 30)] - s);  // rhs row
            }
        }
    } else if (c > D.nc) {
        // identity on the padding columns of the panels; the pivot of the right-hand-side row: far above anything
        // y^T y can reach, so the last pivot of the root only ever fails through a NaN from before
        for (int k = threadIdx.x; k < D.sp_n_pad; k += TPB) {
            const int q = D.sp_pad_cols[k];
            D.T[(size_t)D.sp_diag_tile[q >> 6] * (PB * PB) + (q & (PB - 1)) * (PB + 1)] = 1.0;
        }
        if (threadIdx.x == 0) sp_store(D, D.sp_rhs_row, D.sp_rhs_row, 1e100);
    } else if (threadIdx.x < 240) {
        // K corner: 8 lanes per entry take every 8th partial (independent loads), then a fixed 3-step
        // shuffle tree
        const int i = threadIdx.x >> 3, part = threadIdx.x & 7;
        double s = 0.0;
        for (int k = part; k < D.n_lm_blocks; k += 8) s += D.kk_part[(size_t)30 * k + i];
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 1);
        if (part != 0) return;
        if (i < 25) {
            const int a = i / 5, bb = i % 5;
            double v = D.klin[i] - s;
            if (a == bb) v += lambda * clampd(D.klin[i], 1e-6, 1e32);
            sp_store(D, D.sp_posK + a, D.sp_posK + bb, v);
        } else {
            sp_store(D, D.sp_rhs_row, D.sp_posK + (i - 25), D.klin[i] - s);
        }
    }
}

// K-D2 and K-D3 write disjoint parts of S from the same inputs: one launch, the block range decides the role
// (two launches of a few microseconds each cost the chain a launch gap more).
__global__ __launch_bounds__(TPB) void ba_assemble(BaDev D, double lambda, unsigned n_block_groups) {
    if (blockIdx.x < n_block_groups) assemble_blocks(D, lambda, blockIdx.x);
    else assemble_border(D, lambda, (int)(blockIdx.x - n_block_groups));
}

// ---- The Schur stage, landmark-major (round 5): ONE launch for K-C, K-C2 and K-D1 ------------------------------------------
// A workgroup owns a group of landmarks (ba_groups.hpp): consecutive landmarks of an order that keeps their camera sets close,
// <= g_rows rows. Phase 0 (thread = landmark): the 3x3 elimination (eliminate_landmark) — Linv, gt, EKt -> lmtry (the
// step's back-substitution reads it) and, in LDS, Linv, the point and the landmark's own row Y = [EKt; gt]. Phase A (thread =
// observation row): Jacobians recomputed at the linearisation point, Et = E Linv^T -> the row's 144 bytes IN LDS (rounds 1-4
// wrote them to HBM: 72 MB per try, and gathered two rows per entry from there). Phase B (wave = chunk of 64 slices, lane =
// one slice = <= GRP_SLICE entries of ONE block): the 6x6 products X Y^T out of LDS, summed in the lane in entry order -> the
// lane's partial, 288 bytes. No cross-lane step at all (a transposed segment sum over lanes was tried first: it cost 1.5 x the
// products whatever its form). Camera blocks, the calibration border, the right-hand side and the K corner are all the same
// sum (the landmark rows act as the observations of a pseudo-camera). Fixed order everywhere, no atomics.
// The per-group arrays are padded to the group bounds, so a thread's first loads need nothing but its block index; the launch
// also clears the tiles of S and carries the linearisation's second stage, as ba_eliminate does.
__host__ __device__ inline size_t schur_groups_lds_bytes(int rows) {
    return sizeof(double) * ((size_t)18 * (rows + 1) + (size_t)9 * (rows / 4));
}
template <int NR>  // rows per thread in phase A: g_rows <= NR * TPB
__global__ __launch_bounds__(TPB, 4 - NR) void ba_schur_groups(BaDev D, double lambda, const double* __restrict__ clpart, int n_finish) {
    extern __shared__ __attribute__((aligned(16))) double g_lds[];
    const int n_grp_blocks = (int)gridDim.x - n_finish;
    if ((int)blockIdx.x >= n_grp_blocks) {  // workgroup-uniform
        if (threadIdx.x < 64) finish_linearize_block(D, clpart, (int)blockIdx.x - n_grp_blocks, (int)threadIdx.x);
        return;
    }
    const int g = blockIdx.x, R = D.g_rows, LMAX = R / 4, tid = threadIdx.x;
    const bool live = g < D.g_ngroups;
    // everything below that comes from memory and needs only the block index is requested first
    int2 ari[NR];
    double2 auv[NR];
    int lmj = -1, lmr = 0;
    BaGroup G = BaGroup{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < NR; ++it) ari[it] = make_int2(-1, 0), auv[it] = double2{0.0, 0.0};
    if (live) {
        G = D.g_groups[g];
#pragma unroll
        for (int it = 0; it < NR; ++it)
            if (tid + it * TPB < R) {
                ari[it] = D.g_rowinfo[(size_t)g * R + tid + it * TPB];
                auv[it] = reinterpret_cast<const double2*>(D.g_uv)[(size_t)g * R + tid + it * TPB];
            }
        if (tid < LMAX) {
            lmj = D.g_lmid[(size_t)g * LMAX + tid];
            lmr = D.g_lmrow[(size_t)g * LMAX + tid];
        }
    }
    {
        double2* S2 = reinterpret_cast<double2*>(D.T);
        const size_t total = (size_t)D.sp_ntiles * (PB * PB / 2);
        for (size_t e = (size_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (size_t)n_grp_blocks * TPB) S2[e] = make_double2(0.0, 0.0);
    }
    if (!live) return;
    double* rows = g_lds;
    double* linv = rows + 18 * (R + 1);
    double* ptl = linv + 6 * LMAX;
    const int lane = tid & 63, wave = tid >> 6;
    bool arow[NR];  // an observation row (a landmark's own row is written in phase 0)
    double axr[NR][12];
#pragma unroll
    for (int it = 0; it < NR; ++it) {
        arow[it] = ari[it].x >= 0 && ari[it].x < D.nc;
        const double* x = D.pose + 12 * (size_t)(arow[it] ? ari[it].x : 0);
#pragma unroll
        for (int k = 0; k < 12; ++k) axr[it][k] = x[k];
    }
    // ---- phase 0: the group's landmarks (one per thread: a group has at most g_rows / 4 <= TPB) ----
    if (lmj >= 0) {
        const int t = tid, j = lmj;
        double* lrw = rows + 18 * lmr;  // the landmark's own row
        const double* in = D.lmlin + (size_t)LMLIN * j;
        double* out = D.lmtry + (size_t)LMLIN * j;
        double h0 = in[0], h1 = in[1], h2 = in[2], h3 = in[3], h4 = in[4], h5 = in[5];
        h0 += lambda * clampd(h0, 1e-6, 1e32);
        h3 += lambda * clampd(h3, 1e-6, 1e32);
        h5 += lambda * clampd(h5, 1e-6, 1e32);
        bool ok = h0 > 0.0;
        const double l00 = sqrt(ok ? h0 : 1.0);
        const double l10 = h1 / l00, l20 = h2 / l00;
        const double d1 = h3 - l10 * l10;
        ok = ok && d1 > 0.0;
        const double l11 = sqrt(d1 > 0.0 ? d1 : 1.0);
        const double l21 = (h4 - l20 * l10) / l11;
        const double d2 = h5 - l20 * l20 - l21 * l21;
        ok = ok && d2 > 0.0;
        const double l22 = sqrt(d2 > 0.0 ? d2 : 1.0);
        if (!ok) atomicOr(D.flags, 1);
        const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;
        const double m10 = -l10 * m00 * m11;
        const double m21 = -l21 * m11 * m22;
        const double m20 = -(l20 * m00 + l21 * m10) * m22;
        out[0] = m00; out[1] = m10; out[2] = m11; out[3] = m20; out[4] = m21; out[5] = m22;
        double* lv = linv + 6 * t;
        lv[0] = m00; lv[1] = m10; lv[2] = m11; lv[3] = m20; lv[4] = m21; lv[5] = m22;
        const double g0 = in[6], g1 = in[7], g2 = in[8];
        const double t0 = m00 * g0, t1 = m10 * g0 + m11 * g1, t2 = m20 * g0 + m21 * g1 + m22 * g2;
        out[6] = t0; out[7] = t1; out[8] = t2;
        lrw[15] = t0; lrw[16] = t1; lrw[17] = t2;
#pragma unroll
        for (int a = 0; a < 5; ++a) {  // EKt[a] = Linv * ElK[a]
            const double e0 = in[9 + 3 * a], e1 = in[10 + 3 * a], e2 = in[11 + 3 * a];
            const double k0 = m00 * e0, k1 = m10 * e0 + m11 * e1, k2 = m20 * e0 + m21 * e1 + m22 * e2;
            out[9 + 3 * a] = k0; out[10 + 3 * a] = k1; out[11 + 3 * a] = k2;
            lrw[3 * a] = k0; lrw[3 * a + 1] = k1; lrw[3 * a + 2] = k2;
        }
        ptl[3 * t] = D.pt[3 * (size_t)j]; ptl[3 * t + 1] = D.pt[3 * (size_t)j + 1]; ptl[3 * t + 2] = D.pt[3 * (size_t)j + 2];
    }
    if (tid >= TPB - 18) rows[18 * R + (tid - (TPB - 18))] = 0.0;  // the null row of the padding entries
    __syncthreads();
    // ---- phase A: Et of the group's observation rows (NR per thread) ----
#pragma unroll
    for (int it = 0; it < NR; ++it)
    if (arow[it]) {
        double K[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) K[k] = D.Kc[k];
        const int lt = ari[it].y;
        const double l[3] = {ptl[3 * lt], ptl[3 * lt + 1], ptl[3 * lt + 2]};
        const double* m = linv + 6 * lt;
        const double m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5];
        double Ap[12], Al[6], Ak[10], b[2], et[18];
        obs_factor(axr[it], l, K, auv[it].x, auv[it].y, D.nz.pix_sigma, D.nz.pix_huber, Ap, Al, Ak, b);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double e0 = Ap[a] * Al[0] + Ap[6 + a] * Al[3], e1 = Ap[a] * Al[1] + Ap[6 + a] * Al[4], e2 = Ap[a] * Al[2] + Ap[6 + a] * Al[5];
            et[3 * a] = m0 * e0;
            et[3 * a + 1] = m1 * e0 + m2 * e1;
            et[3 * a + 2] = m3 * e0 + m4 * e1 + m5 * e2;
        }
        double2* out = reinterpret_cast<double2*>(rows + 18 * (tid + it * TPB));
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = double2{et[2 * k], et[2 * k + 1]};
    }
    __syncthreads();
    // ---- phase B: the slices, one per lane ----
#ifdef EXP_GRP_NO_B  // (knock-out, timing only)
    if (lambda > -1.0) return;
#endif
    for (int c = wave; c < G.nchunks; c += TPB / 64) {
        // (requesting the first chunk's record, lane words and entries before phase 0 — they need only the group record — was
        // measured: 75.4 against 72.6 us, and four spilled registers in the 256-row form)
        const BaChunk ch = D.g_chunks[G.chunk0 + c];
        const uint32_t info = D.g_laneinfo[(size_t)(G.chunk0 + c) * 64 + lane];
        const uint4* ep = reinterpret_cast<const uint4*>(D.g_ent) + (size_t)ch.ent0 * 64 + lane;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0.0;
        uint4 e4 = ep[0];
        for (int i4 = 0; i4 < ch.n4; ++i4) {  // (wave-uniform trip count: shorter slices carry null entries)
            const uint32_t ent[4] = {e4.x, e4.y, e4.z, e4.w};
            if (i4 + 1 < ch.n4) e4 = ep[(size_t)(i4 + 1) * 64];  // the next step's entries fly under this step's products
            // (holding the rows of two entries at once — the second's loads under the first's products, 235 registers, which the two
            // waves per SIMD of the 480-row form could afford — measured no gain: 75 against 72.7 us)
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                uint32_t e = ent[0];  // (selected, not indexed: a runtime index would put the four words in scratch)
#pragma unroll
                for (int w = 1; w < 4; ++w) e = i == w ? ent[w] : e;
                const double2* xp = reinterpret_cast<const double2*>(rows + 18 * (e & 0xffffu));
                const double2* yp = reinterpret_cast<const double2*>(rows + 18 * (e >> 16));
                double x[18], y[18];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const double2 a = xp[k], b = yp[k];
                    x[2 * k] = a.x, x[2 * k + 1] = a.y, y[2 * k] = b.x, y[2 * k + 1] = b.y;
                }
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = 0; b < 6; ++b) acc[6 * a + b] += x[3 * a] * y[3 * b] + x[3 * a + 1] * y[3 * b + 1] + x[3 * a + 2] * y[3 * b + 2];
            }
        }
        // the lanes of a segment (adjacent, at most GRP_SEG): lane i takes [i, i + 2 d) in step d — after three steps the
        // segment's first lane holds its sum, in a fixed tree order
        const int after = (int)(info >> 28);
        static_assert(GRP_SEG == 8, "three steps");
        // (a segment never leaves its row of 16 lanes, so "the value d lanes up" is a DPP row shift — two v_mov_dpp per double — and not
        // two ds_bpermute through the LDS pipe the other waves' row reads are using)
        auto fold_step = [&](auto DC) {
            constexpr int d = decltype(DC)::value;
            const bool take = after >= d;
#ifdef EXP_GRP_NO_FOLD  // (knock-out, timing only)
            if (lambda > -1.0) return;
#endif
            if (!__ballot(take)) return;  // (wave-uniform: nobody in this chunk reaches that far — chunks of one-lane segments skip all three)
#pragma unroll
            for (int k = 0; k < 36; ++k) {
                const int lo = __double2loint(acc[k]), hi = __double2hiint(acc[k]);
                const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + d, 0xf, 0xf, true);   // row_shl:d, out-of-row sources read 0
                const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + d, 0xf, 0xf, true);
                const double o = __hiloint2double(shi, slo);
                acc[k] += take ? o : 0.0;
            }
        };
        fold_step(std::integral_constant<int, 1>{});
        fold_step(std::integral_constant<int, 2>{});
        fold_step(std::integral_constant<int, 4>{});
        const int slot = (int)(info & 0x0fffffffu) - 1;
        if (slot >= 0) {
            double2* dst = reinterpret_cast<double2*>(D.partial + (size_t)36 * slot);
#pragma unroll
            for (int k = 0; k < 18; ++k) dst[k] = double2{acc[2 * k], acc[2 * k + 1]};
        }
    }
}

// One WAVE per block: lane l adds the block's partials l, l + 64, ... (288 contiguous bytes each; a block's partials are contiguous)
// in that order, the 64 lane sums are folded by a fixed xor butterfly; lanes 0..35 then place one element each. Blocks with more
// than GRP_LONG partials — the pseudo-camera's and the diagonal ones collect a partial from nearly every group a camera appears
// in — get a whole WORKGROUP (thread t adds partials t, t + 256, ...; the four waves' sums are added in wave order): walked by one
// thread per element such a list alone took 0.67 ms on S200, by one wave 90 us.
__device__ __forceinline__ void place_group_block(const BaDev& D, double lambda, const int4 B, int el, double s) {
    const int a = el / 6, b = el % 6;
    if (B.y < D.nc) {  // camera block
        double v = -s;
        if (B.x == B.y) {
            const double h = D.camlin[(size_t)CAMLIN * B.x + 6 * a + b];
            v += h;
            if (a == b) v += lambda * clampd(h, 1e-6, 1e32);
        }
        const int r = D.sp_pos[B.x] + a, q = D.sp_pos[B.y] + b;
        sp_store(D, r, q, v);
        if (B.x != B.y) sp_store(D, q, r, v);
    } else if (B.x < D.nc) {  // (c, K): border | right-hand side of camera c
        const double* cl = D.camlin + (size_t)CAMLIN * B.x;
        if (b < 5) {
            const double v = cl[36 + 5 * a + b] - s;
            sp_store(D, D.sp_pos[B.x] + a, D.sp_posK + b, v);
            sp_store(D, D.sp_posK + b, D.sp_pos[B.x] + a, v);
        } else {
            sp_store(D, D.sp_rhs_row, D.sp_pos[B.x] + a, cl[66 + a] - s);
        }
    } else if (a < 5) {  // (K, K): K corner | right-hand side of K
        if (b < 5) {
            double v = D.klin[5 * a + b] - s;
            if (a == b) v += lambda * clampd(D.klin[5 * a + b], 1e-6, 1e32);
            sp_store(D, D.sp_posK + a, D.sp_posK + b, v);
        } else {
            sp_store(D, D.sp_rhs_row, D.sp_posK + a, D.klin[25 + a] - s);
        }
    }
}
// A block's partials are contiguous: element el of partial k sits at partial[36 (first + k) + el], so 36 adjacent threads read 288
// contiguous bytes per k and every thread adds ITS element in k order (eight independent loads in flight per thread).
constexpr int ASM_TPB = 1024;  // workgroup of ba_assemble_groups: 16 blocks of the table (a wave each), or ONE long block
__device__ __forceinline__ double sum_partials(const double* __restrict__ pp, int first, int n, int stride) {
    double s = 0.0;
    int k = first;
    for (; k + 7 * stride < n; k += 8 * stride) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = pp[(size_t)36 * (k + i * stride)];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; k < n; k += stride) s += pp[(size_t)36 * k];
    return s;
}
__device__ __forceinline__ void assemble_group_blocks(const BaDev& D, double lambda, unsigned block) {
    const int blk = (int)block * (ASM_TPB / 64) + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blk >= D.g_nblk || lane >= 36) return;
    const int4 B = D.g_blk[blk];
    if (B.w > GRP_LONG) return;  // assemble_long_block's
    place_group_block(D, lambda, B, lane, sum_partials(D.partial + (size_t)36 * B.z + lane, 0, B.w, 1));
}
// A block with more than GRP_LONG partials — the pseudo-camera's blocks and the diagonal ones collect a partial from nearly
// every group a camera appears in — gets a WORKGROUP: thread (sub, el) adds partials sub, sub + 28, ... of element el, the 28
// sub-sums are added in order. (Walked by one thread per element such a list alone took 0.67 ms on S200.)
__device__ __forceinline__ void assemble_long_block(const BaDev& D, double lambda, int which) {
    constexpr int NSUB = ASM_TPB / 36;  // 28
    __shared__ double wsum[NSUB][36];
    const int4 B = D.g_blk[D.g_longblk[which]];
    const int sub = threadIdx.x / 36, el = threadIdx.x % 36;
    if (sub < NSUB) wsum[sub][el] = sum_partials(D.partial + (size_t)36 * B.z + el, sub, B.w, NSUB);
    __syncthreads();
    if (threadIdx.x < 36) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NSUB; ++w) s += wsum[w][threadIdx.x];
        place_group_block(D, lambda, B, (int)threadIdx.x, s);
    }
}
__global__ __launch_bounds__(ASM_TPB) void ba_assemble_groups(BaDev D, double lambda, unsigned n_block_groups) {
    // the long blocks first (they are the launch's critical path), then sixteen blocks per workgroup, then the padding columns'
    // identity and the pivot of the right-hand-side row
    if ((int)blockIdx.x < D.g_nlong) assemble_long_block(D, lambda, (int)blockIdx.x);
    else if (blockIdx.x - D.g_nlong < n_block_groups) assemble_group_blocks(D, lambda, blockIdx.x - D.g_nlong);
    else if (threadIdx.x < TPB) assemble_border(D, lambda, D.nc + 1);
}


// sum of v over aligned groups of W lanes inside a 16-lane DPP row; valid in each group's last lane
template <int W>
__device__ __forceinline__ double dpp_row_sum(double v) {
    static_assert(W == 8 || W == 16, "row_shr steps 1, 2, 4 (, 8)");
#define EACHAM_DPP_STEP(ctrl)                                                                       \
    {                                                                                               \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);     \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);     \
        v += __hiloint2double(hi, lo);                                                              \
    }
    EACHAM_DPP_STEP(0x111)  // row_shr:1
    EACHAM_DPP_STEP(0x112)  // row_shr:2
    EACHAM_DPP_STEP(0x114)  // row_shr:4
    if (W == 16) EACHAM_DPP_STEP(0x118)  // row_shr:8
#undef EACHAM_DPP_STEP
    return v;
}

// v of lane (lane ^ 1), (lane ^ 2) inside a quad
__device__ __forceinline__ double dpp_quad_xor1(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_quad_xor2(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x4E, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x4E, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int SRC>  // v of lane SRC of the quad
__device__ __forceinline__ double dpp_quad_bcast(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), SRC * 0x55, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), SRC * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ---- The Schur stage, DENSE form (round 5; ba_window.hpp): a local window's try up to the partial sums of the reduced system in ONE
// launch, from the VALUES — no linearisation launch, no pair lists, no entry lists. Per-frame RefineBA (apps/sfm/main.cpp:207) solves
// a NEW window every frame: what a launch saves per try has to be paid for by the structure built per window, and this form's
// structure is the landmark-ordered rows alone. A workgroup owns a group of landmarks (<= w_rows rows, <= 64 landmarks: their
// observations in camera order, then a row of their own) and forms its share of EVERY 6x6 block of the system ((nc + 1)(nc + 2) / 2
// with the calibration + right-hand-side pseudo-camera nc):
//   A1 (thread = row)        Jacobians at the current values; Al | Ak | b -> the row (for the landmark's thread), Ap^T and Q^T =
//                            [Ak; b]^T (6x2 each) -> the row (the Hessian terms of phase B); the row's landmark bit -> its camera's
//                            mask, its index -> rowtab[camera][landmark]
//   0  (thread = landmark)   Hll, gl, ElK summed over its rows in row order (+ its prior) -> lmlin; the 3x3 elimination
//                            (eliminate_landmark) -> lmtry, Linv in LDS, its own row Y = [EKt; gt]
//   A2 (thread = row)        Et = Ap^T Al Linv^T -> the row; the calibration Hessian HKK, gK by a block sum
//   B  (nine lanes = a block, 28 blocks at a time, a 2x2 piece per lane) the landmarks that see both cameras of block (c1, c2) are
//      mask[c1] & mask[c2]; in ascending landmark order  acc -= Et_r1 Et_r2^T  (r = rowtab[c][t]) in REGISTERS, and the Hessian
//      terms where they belong: (c, c) += Ap^T Ap, (c, K) += Ap^T Q; the piece goes straight to the group's partial. The heavy
//      blocks ((K, K): every landmark; (c, K), (c, c): every landmark camera c sees) come first, together.
// No atomics on data, every sum in one order. -> w_part[group] (win_stride(nc) doubles), summed over the groups by ba_assemble_dense.
#ifdef EXP_WD_STAMPS  // (diagnostic build: wall-clock stamps of workgroup 0 at the phase boundaries, 10 ns units; printed by ba_run)
__device__ unsigned long long g_wd_stamp[16];
#define WD_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); if (blockIdx.x == 0) g_wd_stamp[i] = t_; \
    if ((i) == 0) atomicMin(&g_wd_stamp[10], t_); if ((i) == 8) atomicMax(&g_wd_stamp[11], t_); if ((i) == 7) atomicMax(&g_wd_stamp[12], t_); if ((i) == 4) atomicMax(&g_wd_stamp[13], t_); } } while (0)
#else
#define WD_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(WIN_TPB) void ba_schur_dense(BaDev D, double lambda) {
    extern __shared__ __attribute__((aligned(16))) double w_lds[];
    WD_STAMP(0);
    const int g = blockIdx.x, R = D.w_rows, LMAX = R / 4, tid = threadIdx.x, nc = D.nc;
    const int lane = tid & 63, wave = tid >> 6;
    int2 ri = make_int2(-1, 0);
    double2 uv = double2{0.0, 0.0};
    int lmj = -1, lmr = 0, lmprev = -1;
    if (tid < R) {
        ri = D.w_rowinfo[(size_t)g * R + tid];
        uv = reinterpret_cast<const double2*>(D.w_uv)[(size_t)g * R + tid];
    }
    const int lt = tid >> 2, lsub = tid & 3;  // four lanes per landmark in phase 0 (tid < 4 LMAX = R)
    if (tid < R) {
        lmj = D.w_lmid[(size_t)g * LMAX + lt];
        lmr = D.w_lmrow[(size_t)g * LMAX + lt];
        if (lt > 0) lmprev = D.w_lmrow[(size_t)g * LMAX + lt - 1];
    }
    {   // S is rebuilt for every lambda (the factorisation works in place): cleared here, filled by ba_assemble_dense
        double2* S2 = reinterpret_cast<double2*>(D.T);
        const size_t total = (size_t)D.sp_ntiles * (PB * PB / 2);
        for (size_t e = (size_t)blockIdx.x * WIN_TPB + threadIdx.x; e < total; e += (size_t)gridDim.x * WIN_TPB) S2[e] = make_double2(0.0, 0.0);
    }
    const int nblk = win_nblk(nc);
    double* rows = w_lds;
    double* linv = rows + WIN_ROW * R;
    double* ptl = linv + 6 * LMAX;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(ptl + 3 * LMAX);   // [nc + 1] landmarks of the group a camera sees
    unsigned char* rowtab = reinterpret_cast<unsigned char*>(mask + nc + 1);             // [nc + 1][LMAX] its row of that landmark
    uint32_t* vis = reinterpret_cast<uint32_t*>(reinterpret_cast<double*>(mask + nc + 1) + ((nc + 1) * LMAX + 7) / 8);  // the visit list of phase B
    unsigned short* gstart = reinterpret_cast<unsigned short*>(reinterpret_cast<double*>(vis) + (win_visits_max(nc, R) + 1) / 2);  // [WIN_LANE_GROUPS + 1] first visit of a lane group
    unsigned char* npiece = reinterpret_cast<unsigned char*>(gstart + WIN_LANE_GROUPS + 2);                                        // [2 nc] pieces of a long block
    double* piece = reinterpret_cast<double*>(gstart) + (2 * (WIN_LANE_GROUPS + 2) + 2 * nc + 7) / 8;                              // [2 nc][WIN_PIECES][42] their sums
    double* bsum = piece + 42 * WIN_PIECES * 2 * nc;  // [4 (WIN_TPB / 64)][WIN_KK] (+ 8 for the scan below)
    if (tid <= nc) mask[tid] = 0ull;
    const bool arow = ri.x >= 0 && ri.x < nc;
    double xr[12], K[5];
    {
        const double* x = D.pose + 12 * (size_t)(arow ? ri.x : 0);
#pragma unroll
        for (int k = 0; k < 12; ++k) xr[k] = x[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) K[k] = D.Kc[k];
    }
    double lp[3] = {0.0, 0.0, 0.0};
    if (lmj >= 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) lp[k] = D.pt[3 * (size_t)lmj + k];
        if (lsub == 0) ptl[3 * lt] = lp[0], ptl[3 * lt + 1] = lp[1], ptl[3 * lt + 2] = lp[2];
    }
    __syncthreads();
#if EXP_WD_STOP == 1  // (knock-outs, timing only: the kernel up to here)
    return;
#endif
    WD_STAMP(1);
    // ---- A1 ----
    if (ri.x >= 0) {
        atomicOr(&mask[ri.x], 1ull << ri.y);  // (bits: any order gives the same word)
        rowtab[ri.x * LMAX + ri.y] = (unsigned char)tid;
    }
    double Ap[12], Al[6];
    double* rw = rows + WIN_ROW * tid;
#pragma unroll
    for (int k = 0; k < 12; ++k) Ap[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) Al[k] = 0.0;
    double kk[WIN_KK];
#pragma unroll
    for (int k = 0; k < WIN_KK; ++k) kk[k] = 0.0;
    if (arow) {
        const double l[3] = {ptl[3 * ri.y], ptl[3 * ri.y + 1], ptl[3 * ri.y + 2]};
        double Ak[10], b[2];
        obs_factor(xr, l, K, uv.x, uv.y, D.nz.pix_sigma, D.nz.pix_huber, Ap, Al, Ak, b);
        double2* r2 = reinterpret_cast<double2*>(rw);
        r2[0] = double2{Al[0], Al[1]}; r2[1] = double2{Al[2], Al[3]}; r2[2] = double2{Al[4], Al[5]};
#pragma unroll
        for (int k = 0; k < 5; ++k) r2[3 + k] = double2{Ak[2 * k], Ak[2 * k + 1]};
        r2[8] = double2{b[0], b[1]};
#pragma unroll
        for (int a = 0; a < 6; ++a) r2[9 + a] = double2{Ap[a], Ap[6 + a]};            // Ap^T[a][h]
#pragma unroll
        for (int a = 0; a < 5; ++a) r2[15 + a] = double2{Ak[a], Ak[5 + a]};           // Q^T[a][h], a < 5
        r2[20] = double2{b[0], b[1]};                                                 // Q^T[5][h]
        // the calibration Hessian: upper triangle (15) and gradient (5)
        int q = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int bb = a; bb < 5; ++bb) kk[q++] = Ak[a] * Ak[bb] + Ak[5 + a] * Ak[5 + bb];
#pragma unroll
        for (int a = 0; a < 5; ++a) kk[15 + a] = Ak[a] * b[0] + Ak[5 + a] * b[1];
    }
    __syncthreads();
#if EXP_WD_STOP == 2
    return;
#endif
    WD_STAMP(2);
    // ---- 0 ----
    if (lmj >= 0) {
        double rec[LMLIN];
#pragma unroll
        for (int k = 0; k < LMLIN; ++k) rec[k] = 0.0;
        for (int r = lmprev + 1 + lsub; r < lmr; r += 4) {
            const double2* s2 = reinterpret_cast<const double2*>(rows + WIN_ROW * r);
            double v[18];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const double2 t = s2[k];
                v[2 * k] = t.x, v[2 * k + 1] = t.y;
            }
            const double* al = v;        // Al[h][c] = al[3 h + c]
            const double* ak = v + 6;    // Ak[h][a] = ak[5 h + a]
            const double* bv = v + 16;
            rec[0] += al[0] * al[0] + al[3] * al[3];
            rec[1] += al[0] * al[1] + al[3] * al[4];
            rec[2] += al[0] * al[2] + al[3] * al[5];
            rec[3] += al[1] * al[1] + al[4] * al[4];
            rec[4] += al[1] * al[2] + al[4] * al[5];
            rec[5] += al[2] * al[2] + al[5] * al[5];
#pragma unroll
            for (int a = 0; a < 3; ++a) rec[6 + a] += al[a] * bv[0] + al[3 + a] * bv[1];
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) rec[9 + 3 * a + c] += ak[a] * al[c] + ak[5 + a] * al[3 + c];
        }
#pragma unroll
        for (int k = 0; k < LMLIN; ++k) {  // the four lanes' shares, the same total in all of them (fixed butterfly)
            rec[k] += dpp_quad_xor1(rec[k]);
            rec[k] += dpp_quad_xor2(rec[k]);
        }
        {   // PriorFactor<Point3>, Robust(Huber(3/obs), Isotropic(1/obs)) — as linearize_landmark
            const double sg = D.lmprior[2 * (size_t)lmj], kh = D.lmprior[2 * (size_t)lmj + 1];
            double e[3], n2 = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                e[a] = (lp[a] - D.pt0[3 * (size_t)lmj + a]) / sg;
                n2 += e[a] * e[a];
            }
            const double sw = sqrt(huber_weight(sqrt(n2), kh)), w = sw / sg;
            rec[0] += w * w; rec[3] += w * w; rec[5] += w * w;
#pragma unroll
            for (int a = 0; a < 3; ++a) rec[6 + a] += w * (-sw * e[a]);
        }
        const LmElim le = eliminate_landmark(rec, lambda);
        if (lsub == 0) {
        double* lout = D.lmlin + (size_t)LMLIN * lmj;
#pragma unroll
        for (int k = 0; k < LMLIN; ++k) lout[k] = rec[k];
        if (!le.ok) atomicOr(D.flags, 1);
        double* out = D.lmtry + (size_t)LMLIN * lmj;
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = le.m[k], linv[6 * lt + k] = le.m[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) out[6 + k] = le.gt[k];
#pragma unroll
        for (int k = 0; k < 15; ++k) out[9 + k] = le.ek[k];
        double* lrw = rows + WIN_ROW * lmr;  // Y = [EKt; gt]
#pragma unroll
        for (int k = 0; k < 15; ++k) lrw[k] = le.ek[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) lrw[15 + k] = le.gt[k];
        // the landmark's share of block (K, K): EKt EKt^T and EKt gt (the longest sum of the system — every landmark — goes through the
        // block sum below, not through one lane group of phase B)
#pragma unroll
        for (int a = 0; a < 5; ++a) {
#pragma unroll
            for (int bb = 0; bb < 5; ++bb)
                kk[20 + 5 * a + bb] = le.ek[3 * a] * le.ek[3 * bb] + le.ek[3 * a + 1] * le.ek[3 * bb + 1] + le.ek[3 * a + 2] * le.ek[3 * bb + 2];
            kk[45 + a] = le.ek[3 * a] * le.gt[0] + le.ek[3 * a + 1] * le.gt[1] + le.ek[3 * a + 2] * le.gt[2];
        }
        }
    }
    __syncthreads();
#if EXP_WD_STOP == 3
    return;
#endif
    WD_STAMP(3);
    // ---- A2 ----
    if (arow) {
        const double* m = linv + 6 * ri.y;
        const double m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5];
        double et[18];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double e0 = Ap[a] * Al[0] + Ap[6 + a] * Al[3], e1 = Ap[a] * Al[1] + Ap[6 + a] * Al[4], e2 = Ap[a] * Al[2] + Ap[6 + a] * Al[5];
            et[3 * a] = m0 * e0;
            et[3 * a + 1] = m1 * e0 + m2 * e1;
            et[3 * a + 2] = m3 * e0 + m4 * e1 + m5 * e2;
        }
        double2* r2 = reinterpret_cast<double2*>(rw);
#pragma unroll
        for (int k = 0; k < 9; ++k) r2[k] = double2{et[2 * k], et[2 * k + 1]};
    }
    double* part = D.w_part + (size_t)D.w_stride * g;
    {   // the WIN_KK sums over the workgroup: rows of 16 lanes by DPP shifts, the 16 row sums of a value added by ONE thread in order
        // (a shuffle tree over the wave for every value — 600 trips through the LDS pipe — was 6.8 us of this kernel)
#pragma unroll
        for (int k = 0; k < WIN_KK; ++k) kk[k] = dpp_row_sum<16>(kk[k]);
        if ((lane & 15) == 15) {
            double* dst = bsum + WIN_KK * (4 * wave + (lane >> 4));
#pragma unroll
            for (int k = 0; k < WIN_KK; ++k) dst[k] = kk[k];
        }
        __syncthreads();  // (also puts every Et row in place before phase B)
        if (tid < WIN_KK) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 4 * (WIN_TPB / 64); ++r) t += bsum[WIN_KK * r + tid];
            part[nblk * 36 + nc * 12 + tid] = t;
        }
    }
#if EXP_WD_STOP == 4
    return;
#endif
    WD_STAMP(4);
    // ---- B ----
    // The visits (block, landmark) of the group, block by block — (c, K) | (c, c) | (c1 < c2 < nc): the long sums first —, each
    // block's landmarks in ascending order: thread = block counts mask[c1] & mask[c2], a scan gives the block's first visit, the
    // thread writes its visits: rows r1 | r2 << 8, where the sum goes << 16, bit 30 on the last visit of a sum, bit 31 on the first.
    // The 2 nc blocks of the first two kinds — every landmark that sees camera c: up to 64 visits, and they take the Hessian terms
    // too — are cut into <= WIN_PIECES pieces of ~WIN_PIECE visits whose sums meet in LDS and are added in piece order after the
    // loop (as one sum each they WERE the loop: 30 visits in one lane group while the others had six). Lane group l (three lanes:
    // rows 2p, 2p + 1 of c1 against all six rows of c2, twelve sums per lane) takes the sums whose first visit lies in
    // [l L, (l + 1) L), L = visits / 168: whole sums, every pass of the loop the same straight code for every lane — a visit's
    // products, and stores where a sum ends. (Dealt out 28 blocks at a time the longest block of a round was the round's time;
    // handed out one by one from a counter, the hand-out's dependent LDS trips were: 28 us of 40 either way.)
    {
        int* scan = reinterpret_cast<int*>(bsum + 4 * (WIN_TPB / 64) * WIN_KK);  // [WIN_TPB / 64]
        int c1 = 0, c2 = nc, cnt = 0;
        unsigned long long bm = 0ull;
        const int idx = tid;  // (nblk - 1 <= 324 < WIN_TPB)
        static_assert((WIN_NC_MAX + 1) * (WIN_NC_MAX + 2) / 2 - 1 <= WIN_TPB, "a thread per block");
        if (idx < nblk - 1) {
            if (idx < nc) c1 = idx;
            else if (idx < 2 * nc) c1 = c2 = idx - nc;
            else {
                const int q = idx - 2 * nc;  // c2 (c2 - 1) / 2 + c1
                c2 = (int)((1.0f + sqrtf(8.0f * (float)q + 1.0f)) * 0.5f);
                while (c2 * (c2 - 1) / 2 > q) --c2;
                while ((c2 + 1) * c2 / 2 <= q) ++c2;
                c1 = q - c2 * (c2 - 1) / 2;
            }
            bm = mask[c1] & mask[c2];
            cnt = __popcll(bm);
        }
        // exclusive scan of the counts in block order
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) scan[wave] = incl;
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < WIN_TPB / 64; ++w) {
            const int sw = scan[w];
            if (w < wave) base += sw;
            total += sw;
        }
        const int start_a = base + incl - cnt;
        const int L = (total + WIN_LANE_GROUPS - 1) / WIN_LANE_GROUPS > 0 ? (total + WIN_LANE_GROUPS - 1) / WIN_LANE_GROUPS : 1;
        WD_STAMP(5);
        if (idx < nblk - 1) {
            const int sb = c2 * (c2 + 1) / 2 + c1;
            const bool heavy = idx < 2 * nc;  // (c, K), (c, c): Hessian terms, pieces
            const unsigned char *u1 = rowtab + c1 * LMAX, *u2 = rowtab + c2 * LMAX;
            int np = 1;
            if (heavy) {
                np = (cnt + WIN_PIECE - 1) / WIN_PIECE;
                np = np > WIN_PIECES ? WIN_PIECES : np;
                npiece[idx] = (unsigned char)np;  // (0: nobody contributes)
            }
            unsigned long long mm = bm;
            int e = 0;
            for (int q = 0; q < np; ++q) {
                const int len = cnt / np + (q < cnt % np ? 1 : 0);
                const uint32_t hi = heavy ? (uint32_t)(WIN_PIECES * idx + q) << 16 | 1u << 29 : (uint32_t)sb << 16;
                for (int i = 0; i < len; ++i, ++e) {
                    const int t = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    vis[start_a + e] = (uint32_t)u1[t] | (uint32_t)u2[t] << 8 | hi | (i == len - 1 ? 1u << 30 : 0u) | (i == 0 ? 1u << 31 : 0u);
                }
            }
            if (cnt == 0) {  // nobody contributes: the block is zeros
                double2* A = reinterpret_cast<double2*>(part + 36 * sb);
                for (int i = 0; i < 18; ++i) A[i] = double2{0.0, 0.0};
                if (heavy)
                    for (int i = 0; i < 6; ++i) part[36 * nblk + 12 * c1 + (c1 == c2 ? 0 : 6) + i] = 0.0;
            }
        }
        __syncthreads();
        WD_STAMP(6);
        // first visit of every lane group: the first sum that starts at or behind l L
        if (tid <= WIN_LANE_GROUPS) {
            int at = tid * L;
            at = at < total ? at : total;
            while (at < total && !(vis[at] >> 31)) ++at;
            gstart[tid] = (unsigned short)at;
        }
        __syncthreads();
        WD_STAMP(7);
        const int v = lane / 3, p = lane - 3 * v, grp = 21 * wave + v;  // three lanes per block: rows 2p, 2p + 1 of c1 (lane 63 idles)
        int pos = 0, end = 0;
        if (v < 21) pos = gstart[grp], end = gstart[grp + 1];
        double acc[12], h0 = 0.0, h1 = 0.0;
#pragma unroll
        for (int e = 0; e < 12; ++e) acc[e] = 0.0;
        uint32_t w = pos < end ? vis[pos] : 0u;
        while (__ballot(pos < end)) {
            if (pos < end) {
                const int r1 = (int)(w & 0xffu), r2 = (int)((w >> 8) & 0xffu);
                const bool hess = (w >> 29) & 1u, diag = r1 == r2, last = (w >> 30) & 1u;
                const int sb = (int)((w >> 16) & 0x1ffu);  // block of the partial, or (hess) piece in LDS
                const double2* X2 = reinterpret_cast<const double2*>(rows + WIN_ROW * r1 + 6 * p);
                const double2* Y2 = reinterpret_cast<const double2*>(rows + WIN_ROW * r2);
                const double2 x0 = X2[0], x1 = X2[1], x2 = X2[2];
                double y[18];
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    const double2 t = Y2[e];
                    y[2 * e] = t.x, y[2 * e + 1] = t.y;
                }
                double2 pa = double2{0.0, 0.0}, pb = pa;
                double z[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) z[e] = 0.0;
                if (hess) {
                    const double2* P2 = reinterpret_cast<const double2*>(rows + WIN_ROW * r1 + 18 + 4 * p);
                    const double2* Z2 = reinterpret_cast<const double2*>(rows + WIN_ROW * r1 + (diag ? 18 : 30));
                    pa = P2[0], pb = P2[1];
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        const double2 t = Z2[e];
                        z[2 * e] = t.x, z[2 * e + 1] = t.y;
                    }
                }
                ++pos;
                if (pos < end) w = vis[pos];  // the next visit is looked up under these loads
                // rows 2p: (x0.x x0.y x1.x), 2p + 1: (x1.y x2.x x2.y) against column b: y[3 b .. 3 b + 2]
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    acc[b] = __builtin_fma(-x1.x, y[3 * b + 2], __builtin_fma(-x0.y, y[3 * b + 1], __builtin_fma(-x0.x, y[3 * b], acc[b])));
                    acc[6 + b] = __builtin_fma(-x2.y, y[3 * b + 2], __builtin_fma(-x2.x, y[3 * b + 1], __builtin_fma(-x1.y, y[3 * b], acc[6 + b])));
                }
                if (hess) {
#pragma unroll
                    for (int b = 0; b < 6; ++b) {
                        acc[b] = __builtin_fma(pa.y, z[2 * b + 1], __builtin_fma(pa.x, z[2 * b], acc[b]));
                        acc[6 + b] = __builtin_fma(pb.y, z[2 * b + 1], __builtin_fma(pb.x, z[2 * b], acc[6 + b]));
                    }
                    // the Hessian's own diagonal (damping, linearised cost change: elements (2p, 2p), (2p + 1, 2p + 1) of a diagonal
                    // block) | its gradient (column 5 of (c, K))
                    const int b0 = diag ? 2 * p : 5, b1 = diag ? 2 * p + 1 : 5;
                    const double z00 = b0 == 0 ? z[0] : (b0 == 2 ? z[4] : (b0 == 4 ? z[8] : z[10]));
                    const double z01 = b0 == 0 ? z[1] : (b0 == 2 ? z[5] : (b0 == 4 ? z[9] : z[11]));
                    const double z10 = b1 == 1 ? z[2] : (b1 == 3 ? z[6] : z[10]);
                    const double z11 = b1 == 1 ? z[3] : (b1 == 3 ? z[7] : z[11]);
                    h0 = __builtin_fma(pa.y, z01, __builtin_fma(pa.x, z00, h0));
                    h1 = __builtin_fma(pb.y, z11, __builtin_fma(pb.x, z10, h1));
                }
                if (last) {  // the sum is complete: its twelve values of this lane — to the partial, or to its piece —, and start the next one
                    double2* A = reinterpret_cast<double2*>(hess ? piece + 42 * sb + 12 * p : part + 36 * sb + 12 * p);
#pragma unroll
                    for (int e = 0; e < 6; ++e) A[e] = double2{acc[2 * e], acc[2 * e + 1]};
                    if (hess) *reinterpret_cast<double2*>(piece + 42 * sb + 36 + 2 * p) = double2{h0, h1};
#pragma unroll
                    for (int e = 0; e < 12; ++e) acc[e] = 0.0;
                    h0 = h1 = 0.0;
                }
            }
        }
        WD_STAMP(8);
#ifdef EXP_WD_STAMPS
        if (lane == 0 && blockIdx.x == 0) atomicMax(&g_wd_stamp[14], wall_clock64());  // the last wave of workgroup 0 out of the loop
#endif
        __syncthreads();
        // the pieces of the long blocks, in piece order -> the partial
        for (int e = tid; e < 2 * nc * 42; e += WIN_TPB) {
            const int hb = e / 42, el = e - 42 * hb, np = npiece[hb];
            if (np == 0) continue;
            double t = 0.0;
            for (int q = 0; q < np; ++q) t += piece[42 * (WIN_PIECES * hb + q) + el];
            const int c = hb < nc ? hb : hb - nc, sb = hb < nc ? nc * (nc + 1) / 2 + c : c * (c + 1) / 2 + c;
            if (el < 36) part[36 * sb + el] = t;
            else part[36 * nblk + 12 * c + (hb < nc ? 6 : 0) + (el - 36)] = t;
        }
        __syncthreads();
        WD_STAMP(9);
    }
}

// Sum of the groups' partials of ba_schur_dense, the priors and the damping -> the tiles of S; SIXTEEN lanes per element of a block
// (each adds every sixteenth group, the sums are folded by a fixed butterfly: a thread that walked all ~64 groups alone was the
// launch's time). Also leaves what the step's tail reads of the linearisation: the cameras' Hessian diagonal and gradient (camlin),
// the calibration's (klin) — with their priors, as finish_linearize_block leaves them in the other forms.
constexpr int WIN_ASM_LANES = 16;
__global__ __launch_bounds__(TPB) void ba_assemble_dense(BaDev D, double lambda, unsigned n_el_blocks) {
    if (blockIdx.x >= n_el_blocks) {
        assemble_border(D, lambda, D.nc + 1);  // identity on the padding columns, the pivot of the right-hand-side row
        return;
    }
    const int nc = D.nc, nblk = win_nblk(nc), gid = (int)blockIdx.x * TPB + (int)threadIdx.x;
    const int sub = gid % WIN_ASM_LANES;
    const bool valid = gid / WIN_ASM_LANES < nblk * 36;
    const int e = valid ? gid / WIN_ASM_LANES : 0;
    const int blk = e / 36, el = e - 36 * blk, a = el / 6, b = el - 6 * a;
    int c2 = (int)((sqrtf(8.0f * (float)blk + 1.0f) - 1.0f) * 0.5f);
    while ((c2 + 1) * (c2 + 2) / 2 <= blk) ++c2;
    while (c2 * (c2 + 1) / 2 > blk) --c2;
    const int c1 = blk - c2 * (c2 + 1) / 2;
    // the element's second sum: the Hessian's own part where the damping / the step's linearised cost need it apart from the Schur term
    const int hbase = 36 * nblk, kbase = hbase + 12 * nc;
    int off1 = e, off2 = e;
    if (c2 < nc) {
        if (c1 == c2 && a == b) off2 = hbase + 12 * c1 + a;
    } else if (c1 < nc) {
        if (b == 5) off2 = hbase + 12 * c1 + 6 + a;
    } else if (a < 5) {
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        off2 = b < 5 ? kbase + 5 * lo - lo * (lo - 1) / 2 + (hi - lo) : kbase + 15 + a;
        off1 = b < 5 ? kbase + 20 + 5 * a + b : kbase + 45 + a;  // the block's Schur sums came through the groups' block sums (positive)
    }
    // everything the placement needs is requested before the sums: column positions, and the camera's pose where its prior enters
    const bool writer = valid && sub == 0;
    const bool prior = writer && c1 < nc && ((c1 == c2 && a == b) || (c2 == nc && b == 5));
    const int pos1 = D.sp_pos[c1 < nc ? c1 : 0], pos2 = D.sp_pos[c2 < nc ? c2 : 0];
    double x[12], x0[12];
    int fixedc = 0;
    if (prior) {
#pragma unroll
        for (int k = 0; k < 12; ++k) x[k] = D.pose[12 * (size_t)c1 + k], x0[k] = D.pose0[12 * (size_t)c1 + k];
        fixedc = D.fixed[c1];
    }
    const size_t stride = (size_t)D.w_stride;
    const int ng = D.w_ngroups;
    double s = 0.0, s2 = 0.0;
    {
        const double* p1 = D.w_part + off1;
        const double* p2 = D.w_part + off2;
        for (int k = sub; k < ng; k += 4 * WIN_ASM_LANES) {  // eight loads in flight
            double v[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = k + i * WIN_ASM_LANES;
                v[i] = kk < ng ? p1[stride * (size_t)kk] : 0.0;
                w[i] = kk < ng ? p2[stride * (size_t)kk] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s += v[i], s2 += w[i];
        }
#pragma unroll
        for (int d = 1; d < WIN_ASM_LANES; d <<= 1) s += __shfl_xor(s, d), s2 += __shfl_xor(s2, d);
    }
    if (!writer) return;
    if (c1 == nc) s = -s;
    double pd = 0.0, pr = 0.0;
    if (prior) {
        // PriorFactor<Pose3>: e = -Local(x, prior), H = I; Robust(Huber 2.5) unless the node is fixed (finish_linearize_block)
        double xi[6], n2 = 0.0;
        pose_local(x, x0, xi);
        const bool fx = fixedc != 0;
        const double* sg = fx ? D.nz.fixed_sigma : D.nz.pose_sigma;
#pragma unroll
        for (int k = 0; k < 6; ++k) n2 += (xi[k] / sg[k]) * (xi[k] / sg[k]);
        const double sw = fx ? 1.0 : sqrt(huber_weight(sqrt(n2), D.nz.pose_huber));
        pd = (sw / sg[a]) * (sw / sg[a]);
        pr = (sw / sg[a]) * (-sw * (-xi[a] / sg[a]));
    }
    if (c2 < nc) {  // camera block
        double v = s;
        if (c1 == c2 && a == b) {
            const double hd = s2 + pd;
            v += pd + lambda * clampd(hd, 1e-6, 1e32);
            D.camlin[(size_t)CAMLIN * c1 + 7 * a] = hd;
        }
        const int r = pos1 + a, q = pos2 + b;
        sp_store(D, r, q, v);
        if (c1 != c2) sp_store(D, q, r, v);
    } else if (c1 < nc) {  // (c, K): border | right-hand side of camera c
        if (b < 5) {
            sp_store(D, pos1 + a, D.sp_posK + b, s);
            sp_store(D, D.sp_posK + b, pos1 + a, s);
        } else {
            sp_store(D, D.sp_rhs_row, pos1 + a, s + pr);
            D.camlin[(size_t)CAMLIN * c1 + 66 + a] = s2 + pr;
        }
    } else if (a < 5) {  // (K, K): K corner | right-hand side of K
        if (b < 5) {
            double h = s2;
            if (a == b) h += 1.0 / (D.nz.k_sigma[a] * D.nz.k_sigma[a]);
            double v = s + h;
            if (a == b) {
                v += lambda * clampd(h, 1e-6, 1e32);
                D.klin[6 * a] = h;
            }
            sp_store(D, D.sp_posK + a, D.sp_posK + b, v);
        } else {
            const double gk = s2 + (1.0 / D.nz.k_sigma[a]) * (-(D.Kc[a] - D.K0[a]) / D.nz.k_sigma[a]);
            sp_store(D, D.sp_rhs_row, D.sp_posK + a, s + gk);
            D.klin[25 + a] = gk;
        }
    }
}

// ---- K-E: the building blocks of the factorisation of a 64x64 diagonal tile -------------------------------------------
// (factor_32: one wave factorises a 32x32 block in MFMA accumulators; invert_behind_factor: a second wave inverts the
// factor right behind it.) The sparse, level-scheduled factorisation that uses them follows further down (sp_diag /
// sp_level / sp_backsolve); its serial chain is, per level of the elimination tree: the factorising workgroup's loads,
// its panel product, the look-ahead update of the first quadrant, the two diagonal-block factors (ONE wave each, the
// block in MFMA accumulators) and the few hundred cycles the inverting wave trails each factor by.

// first wave only; Dn complete. Publishes L (strictly lower part) and 1/diag through the FactorImage.
// The 32x32 block lives in the accumulators of v_mfma_f64_16x16x4_f64 (three 16x16 blocks of the lower
// triangle; C/D layout col = lane & 15, row = (lane >> 4) + 4 reg) and is factorised four columns at a time:
//   1. the 16-column slab holding the four pivot columns is dumped to LDS (P) — the only cross-lane step;
//   2. every lane reads the 4x4 pivot block (same addresses: LDS broadcast) and the four entries of ITS two
//      rows (l & 15 and 16 + (l & 15)), factorises the 4x4 block and solves its rows against it — all of it
//      in-lane, redundantly in the four lanes that share a row: no readlane, no broadcast;
//   3. lane l keeps component k = l >> 4 of its rows' solution: that IS the A (and B) operand layout of the
//      16x16x4 MFMA, so the rank-4 update of the trailing blocks is one MFMA per 16x16 block.
// The former version (lane = row, pivots and multipliers broadcast with 992 v_readlane into SGPR pairs feeding
// 496 v_fma_f64) spent 13.6k cycles per block, 47 % of the chain of a Cholesky step.
// The non-positive-pivot test stays off the critical path: a bad pivot is only recorded (flags bit 1, the
// solve then counts as failed and its NaNs are never used), it is not replaced. Rows and columns that are
// already factorised keep dead values in the accumulators; they only ever feed other dead elements.
constexpr int PLD = 18;  // row stride of the slab image P[32][PLD]: 16 columns + padding, 16-byte aligned rows

__device__ __forceinline__ double rsqrt_newton(double d) {
    // 1/sqrt(d): hardware estimate (~2^-26 relative, like v_rcp_f64: the compiler's own fp64 division refines it
    // with two Newton steps only to round correctly) + ONE Newton step y <- y (1.5 - (d/2) y^2): the error after
    // the step is ~1.5 e0^2 = a few 1e-16, one or two units in the last place — far inside what a Cholesky factor
    // carries anyway (tests/test_ba_gpu.py holds the step against the oracle at 1e-8). The second step cost three
    // dependent fp64 operations per pivot on the one-wave chain of the diagonal-block factor.
    const double h = 0.5 * d;
    double y = __builtin_amdgcn_rsq(d);
    const double p = h * y, q = __builtin_fma(-p, y, 1.5);
    return y * q;
}

// Fault injection for the time-out branches of the two in-kernel hand-offs (diagnostic build -DEXP_BA_FAULT,
// `make fault`; tests/test_ba_gpu.py). The product build compiles the hooks away.
#ifdef EXP_BA_FAULT
__device__ int g_ba_fault;  // 1: the last super-block of the back-substitution never raises its flag
                            // 2: the first diagonal-block factor drops its last progress store
#define BA_FAULT(k) (g_ba_fault == (k))
#else
#define BA_FAULT(k) false
#endif

typedef int __attribute__((address_space(3))) LdsInt;
constexpr int LFS = NB + 2;  // row stride of the factor image in LDS (16-byte aligned rows)
struct FactorImage {          // what the factorising wave hands to the inverting wave, four columns at a time
    double L[NB][LFS];        // only the strictly lower triangle is meaningful
    double dinv[NB];          // reciprocals of the diagonal
    int progress;             // block columns of four finished so far
};

__device__ __forceinline__ void factor_32(double (*Dn)[NB + 1], double (&P)[NB * PLD] /* LDS slab image */,
                                          FactorImage& F, int* __restrict__ flags, bool inject_fault = false) {
    static_assert(NB == 32, "two 16-row halves");
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    mfma_d4 acc00, acc10, acc11;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        acc00[reg] = Dn[g + 4 * reg][c];
        acc10[reg] = Dn[16 + g + 4 * reg][c];
        acc11[reg] = Dn[16 + g + 4 * reg][16 + c];
    }
    bool bad = false;
#pragma unroll
    for (int s = 0; s < NB / 4; ++s) {
        const int j0 = 4 * s, jb = s >> 2, c0 = j0 & 15;
        // 1. slab -> LDS
        asm volatile("" ::: "memory");
        if (jb == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                P[(g + 4 * reg) * PLD + c] = acc00[reg];
                P[(16 + g + 4 * reg) * PLD + c] = acc10[reg];
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) P[(16 + g + 4 * reg) * PLD + c] = acc11[reg];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // 2. pivot block (lower triangle) and this lane's rows
        double dd[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double2* src = reinterpret_cast<const double2*>(&P[(j0 + a) * PLD + c0]);
            const double2 lo = src[0], hi = src[1];
            dd[a][0] = lo.x, dd[a][1] = lo.y, dd[a][2] = hi.x, dd[a][3] = hi.y;
        }
        double xr[2][4];
#pragma unroll
        for (int h = jb; h < 2; ++h) {
            const double2* src = reinterpret_cast<const double2*>(&P[(16 * h + c) * PLD + c0]);
            const double2 lo = src[0], hi = src[1];
            xr[h][0] = lo.x, xr[h][1] = lo.y, xr[h][2] = hi.x, xr[h][3] = hi.y;
        }
        // 4x4 Cholesky of the pivot block, column by column, the rows of this lane solved alongside
        double y[4], l[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double d = dd[k][k];
#pragma unroll
            for (int m = 0; m < k; ++m) d = __builtin_fma(-l[k][m], l[k][m], d);
            // a non-positive pivot turns into NaN (rsq of a negative, 0 x inf) and stays NaN through every
            // later pivot — rows of identity padding included, 0 x NaN = NaN — so only the last one is tested
            if (s == NB / 4 - 1 && k == 3) bad = !(d > 0.0);
            y[k] = rsqrt_newton(d);
#pragma unroll
            for (int a = k + 1; a < 4; ++a) {
                double v = dd[a][k];
#pragma unroll
                for (int m = 0; m < k; ++m) v = __builtin_fma(-l[a][m], l[k][m], v);
                l[a][k] = v * y[k];
            }
#pragma unroll
            for (int h = jb; h < 2; ++h) {
                double v = xr[h][k];
#pragma unroll
                for (int m = 0; m < k; ++m) v = __builtin_fma(-xr[h][m], l[k][m], v);
                xr[h][k] = v * y[k];
            }
        }
        // 3. rank-4 update of the trailing blocks
        if (s < NB / 4 - 1) {
            // the blocks the next step dumps go first
            const double b1 = g == 0 ? xr[1][0] : (g == 1 ? xr[1][1] : (g == 2 ? xr[1][2] : xr[1][3]));
            if (s < 3) {
                const double b0 = g == 0 ? xr[0][0] : (g == 1 ? xr[0][1] : (g == 2 ? xr[0][2] : xr[0][3]));
                acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-b0, b0, acc00, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-b1, b0, acc10, 0, 0, 0);
            }
            acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-b1, b1, acc11, 0, 0, 0);
        }
        // L and 1/diag of these four columns go to the inverting wave through LDS (lanes 0..15 hold rows c and
        // 16 + c), behind the MFMAs: the stores ride in their shadow. Only the STRICTLY LOWER triangle of the
        // image is meaningful — its one consumer reads nothing else — so the diagonal and what lands above
        // it are left as computed.
        if (g == 0) {  // one exec region: the sixteen lanes store the same 1/diag values and the same flag
#pragma unroll
            for (int h = jb; h < 2; ++h) {
                double2* dst = reinterpret_cast<double2*>(&F.L[16 * h + c][j0]);
                dst[0] = double2{xr[h][0], xr[h][1]}, dst[1] = double2{xr[h][2], xr[h][3]};
            }
            double2* dd2 = reinterpret_cast<double2*>(&F.dinv[j0]);
            dd2[0] = double2{y[0], y[1]}, dd2[1] = double2{y[2], y[3]};
            asm volatile("" ::: "memory");  // LDS executes a wave's operations in order: only the compiler must not reorder
            if (!(inject_fault && BA_FAULT(2) && s == NB / 4 - 1))  // the LAST store: an earlier one is made up for by the next
                *(volatile LdsInt*)&F.progress = s + 1;  // a DS store like the data before it (a flat store is not ordered with them)
        }
    }
    if (bad && threadIdx.x == 0) atomicOr(flags, 2);
}

// One wave, beside the factorising wave: W = L^-1 (lower), produced four rows at a time as soon as the four
// columns of L they end in are published (FactorImage::progress), so that it is complete a few hundred cycles
// after the factor — the next launch turns every panel solve into a product with W (MFMA) instead of a
// 32-step substitution, and the back-substitution wants W anyway. Lanes c and 32 + c own column c of W.
// Right-looking: rhs[ii] = [ii == c] - sum_{m done} L[ii][m] w_m is kept for every row; when the block columns
// 4s..4s+3 arrive, rows 4s..4s+3 are finished by a 4x4 substitution (all that is left behind the factor's
// last step) and their four w's are then folded into the rows still to come. The two halves of the wave share
// that folding: row blocks of four alternate between them (block rb belongs to half rb & 1), the finished w's
// cross with v_permlane32_swap. A lone wave issues one v_fma_f64 per ~8 cycles: on one half-wave the folding
// alone outlasts the factor.
// A lower-triangular 32x32 inverse W as the B operand of the panel product L = A W^T: lane (j = lane & 15, kk = lane >> 4)
// of k-step t holds W[j][4t + kk] (rows 0..15, t < 4: slots 0..3) or W[16 + j][4t + kk] (t < 8: slots 4..11). Stored in
// that order — two slots per 16-byte lane element, op_index — a consumer wave fetches a slot pair with ONE coalesced 1 KB load; read from the
// row-major image, 64 lanes x 8 bytes out of 16 different rows, the operand fetches of a 64-column panel product (40 per wave) took
// most of 19k cycles per launch. X (a full 32x32 block) uses slots 0..7 for its rows 0..15 and 8..15 for rows 16..31.
constexpr int WOP = 12 * 64, XOP = 16 * 64, TILE_OPS = 2 * WOP + XOP;  // per 64x64 diagonal tile: W_a | W_b | X
__device__ __forceinline__ int op_index(int slot, int lane) { return (slot >> 1) * 128 + 2 * lane + (slot & 1); }  // slot pairs: 16-byte loads
__device__ __forceinline__ int wop_index(int row, int col) {
    return row < 16 ? op_index(col >> 2, (col & 3) * 16 + row) : op_index(4 + (col >> 2), (col & 3) * 16 + row - 16);
}
__device__ __forceinline__ int xop_index(int row, int col) {
    return row < 16 ? op_index(col >> 2, (col & 3) * 16 + row) : op_index(8 + (col >> 2), (col & 3) * 16 + row - 16);
}
// W (LDS image, row stride wls) -> global memory, row-major (back-substitution) and in operand order, by `nthreads` threads
template <int NT>
__device__ __forceinline__ void publish_w(const double* Wl, int wls, double* __restrict__ Wout, double* __restrict__ Wop, int thread) {
    double v[NB * NB / NT];
#pragma unroll
    for (int m = 0; m < NB * NB / NT; ++m) v[m] = Wl[((thread + NT * m) / NB) * wls + (thread + NT * m) % NB];  // every read first
#pragma unroll
    for (int m = 0; m < NB * NB / NT; ++m) {
        const int e = thread + NT * m, row = e / NB, col = e % NB;
        Wout[e] = v[m];
        if (row >= 16 || col < 16) Wop[wop_index(row, col)] = v[m];
    }
}
__device__ __forceinline__ void invert_behind_factor(const FactorImage& F, double* __restrict__ Wout, int* __restrict__ flags,
                                                     double* Wlds = nullptr, int wls = 0 /* LDS copy, row stride; Wout may then be null */) {
    const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
    double rhs[NB / 2];  // rhs[4p + a] = row 4 (2p + hh) + a
#pragma unroll
    for (int e = 0; e < NB / 2; ++e) rhs[e] = (4 * (2 * (e >> 2) + hh) + (e & 3)) == c ? 1.0 : 0.0;
    const double* lmine = &F.L[4 * hh][0];  // row 4 (2p + hh) + a starts at lmine + (8p + a) LFS
#pragma unroll
    for (int s = 0; s < NB / 4; ++s) {
        // bounded wait (the factorising wave never waits for anybody, so this cannot expire while it runs);
        // an expired wait marks the solve failed like a hand-off time-out of the back-substitution
        int spins = 0;
        while (*(const volatile LdsInt*)&F.progress <= s && spins < (1 << 24)) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
        }
        if (spins >= (1 << 24) && lane == 0) atomicOr(flags, 4);
        asm volatile("" ::: "memory");
        const double2 dv01 = *reinterpret_cast<const double2*>(&F.dinv[4 * s]), dv23 = *reinterpret_cast<const double2*>(&F.dinv[4 * s + 2]);
        const double l10 = F.L[4 * s + 1][4 * s];
        const double2 l2x = *reinterpret_cast<const double2*>(&F.L[4 * s + 2][4 * s]);   // l20 l21
        const double2 l3x = *reinterpret_cast<const double2*>(&F.L[4 * s + 3][4 * s]);   // l30 l31
        const double l32 = F.L[4 * s + 3][4 * s + 2];
        // the substitution, meaningful in the half that owns row block s
        double w[4];
        w[0] = rhs[4 * (s >> 1)] * dv01.x;
        w[1] = __builtin_fma(-l10, w[0], rhs[4 * (s >> 1) + 1]) * dv01.y;
        w[2] = __builtin_fma(-l2x.y, w[1], __builtin_fma(-l2x.x, w[0], rhs[4 * (s >> 1) + 2])) * dv23.x;
        w[3] = __builtin_fma(-l32, w[2], __builtin_fma(-l3x.y, w[1], __builtin_fma(-l3x.x, w[0], rhs[4 * (s >> 1) + 3]))) * dv23.y;
        if (hh == (s & 1)) {
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2) {
                if (Wout) Wout[(4 * s + a2) * NB + c] = w[a2];
                if (Wlds) Wlds[(4 * s + a2) * wls + c] = w[a2];
            }
        }
        if (s == NB / 4 - 1) break;
#pragma unroll
        for (int a2 = 0; a2 < 4; ++a2) {  // the owner's w's to both halves
            const auto wl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(w[a2]), (unsigned)__double2loint(w[a2]), false, false);
            const auto wh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(w[a2]), (unsigned)__double2hiint(w[a2]), false, false);
            w[a2] = __hiloint2double((int)wh[s & 1], (int)wl[s & 1]);
        }
        // fold them into the rows still to come: row blocks 2p + hh, p >= (s + 1) / 2 (for even s the lower
        // half's first block is the finished one itself: dead values, harmless), two blocks = eight rows at a
        // time with every read of a group issued before its arithmetic
#pragma unroll
        for (int p0 = (s + 1) >> 1; p0 < NB / 8; p0 += 2) {
            double2 la[8], lb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + (u >> 2) < NB / 8) {
                    const double* row = lmine + (8 * (p0 + (u >> 2)) + (u & 3)) * LFS + 4 * s;
                    la[u] = *reinterpret_cast<const double2*>(row);
                    lb[u] = *reinterpret_cast<const double2*>(row + 2);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + (u >> 2) < NB / 8) {
                    double& r = rhs[4 * (p0 + (u >> 2)) + (u & 3)];
                    r = __builtin_fma(-lb[u].y, w[3], __builtin_fma(-lb[u].x, w[2], __builtin_fma(-la[u].y, w[1], __builtin_fma(-la[u].x, w[0], r))));
                }
            // the next group's reads stay behind this group's arithmetic (hoisted together they spill): a memory
            // clobber alone lets the compiler sink the FMAs below it, so the results are pinned as operands
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + (u >> 2) < NB / 8) asm volatile("" : "+v"(rhs[4 * (p0 + (u >> 2)) + (u & 3)]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- K-E in panels of 64 columns ----------------------------------------------------------------------------------
// A 64x64 diagonal tile holds two 32x32 diagonal blocks a, b, and everything the second one needs — L_ba = A'_ba W_a^T
// and D_b = A'_bb - L_ba L_ba^T — comes from the tile itself, so ONE workgroup factorises both without a launch in
// between (a third of a 32-column step was paid per LAUNCH: kernel start + dependent global loads):
//   * the panel of a tile row is ONE product with the inverse of the 64x64 factor, [L_ia L_ib] = [A_ia A_ib] W64^T,
//     W64 = [[W_a, 0], [X, W_b]], X = -W_b L_ba W_a: 40 MFMAs per 16 rows. The producer stores W_a, W_b, X in MFMA
//     operand order (wop_index), so the B operands are coalesced 16-byte loads; a wave fetches its own 16 rows of each
//     raw strip as whole rows and is their only reader in LDS: no workgroup barrier before the product;
//   * the update of a target tile is rank 64 per source panel;
//   * a factorising workgroup updates the first quadrant of its tile first (waves 0-2), then wave 0 factorises it, wave 1
//     inverts behind it and waves 2-3 update the other quadrants (kept in registers); after a barrier waves 2-3 form
//     L_ba from the rows they hold (waves 0-1 publish W_a meanwhile), waves 0-2 D_b, then wave 0 factorises D_b, wave 1
//     inverts, waves 2-3 form L_ba W_a, and all four waves finish X and publish W_b.
// LDS: two 64 x 66 strips (67.6 KB, two workgroups per CU). A diagonal target never uses the second strip, and its
// first strip is dead once the quadrants are updated: the factor images live in the second strip's region, the four
// 32 x 33 images of the second half in the first's.
constexpr int LS2 = PB + 2;  // row stride of the strips
constexpr int TS = NB + 1;   // row stride of the 32x32 images of the second half (8-byte accesses only)
constexpr int WLS = NB + 2;  // row stride of the LDS copy of W_a
constexpr int TILE = PB * PB;  // doubles per tile of the sparse storage (row-major, row stride PB)

// L = A W^T for 16 rows whose A operand a[0..7] (columns 0..31) sits in registers, W (lower triangular, row stride ws)
// in LDS or global memory: columns 0..15 -> x0, 16..31 -> x1 + x2
__device__ __forceinline__ void panel32(const double (&a)[8], const double* __restrict__ W, int ws, int mc, int mg, mfma_d4& lo,
                                        mfma_d4& hi) {
    mfma_d4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], W[mc * ws + 4 * t + mg], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], W[(16 + mc) * ws + 4 * t + mg], x1, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t + 4], W[(16 + mc) * ws + 4 * t + 16 + mg], x2, 0, 0, 0);
    }
    lo = x0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) hi[reg] = x1[reg] + x2[reg];
}

struct DiagImages {  // LDS views of a 64x64 diagonal tile being factorised
    double (*Dn)[NB + 1];
    double* Pslab;    // [NB * PLD], 16-byte aligned
    FactorImage* F;
    double* Wa;       // [NB][WLS] copy of W_a (written by the inverting wave of the first half)
    double* T10;      // [NB][TS] A'_ba, then L_ba
    double* T11;      // [NB][TS] A'_bb
    double* P1;       // [NB][TS] L_ba W_a
    double* Wb;       // [NB][TS] copy of W_b
};

// Second half of a diagonal tile: every thread of the workgroup, after a barrier behind which W_a (LDS copy), T10 and
// T11 are complete. Stores W_b -> Wout_b (row-major) and Wop_b, X -> Xop (operand order, for the panel products of the
// launches that use this panel as a source) and Xrow (row-major, for the back-substitution).
__device__ __forceinline__ void diag_second_half(const DiagImages& I, double* __restrict__ Wout_b, double* __restrict__ Wop_b,
                                                 double* __restrict__ Xop, double* __restrict__ Xrow, int* __restrict__ flags, int w0 = 0) {
    const int tid = threadIdx.x, mc = tid & 15, mg = (tid >> 4) & 3;
    int wv = (tid >> 6) - w0;  // waves w0, w0 + 1 form L_ba (in sp_level the two that just wrote those rows of T10)
    if (wv >= 0 && wv < 2) {  // L_ba = A'_ba W_a^T, 16 rows per wave, in place (the wave's own rows: reads before writes, in LDS order)
        double a[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = I.T10[(16 * wv + mc) * TS + 4 * t + mg];
        mfma_d4 lo, hi;
        panel32(a, I.Wa, WLS, mc, mg, lo, hi);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * wv + mg + 4 * reg;
            I.T10[row * TS + mc] = lo[reg];
            I.T10[row * TS + 16 + mc] = hi[reg];
        }
    }
    if (tid == 0) I.F->progress = 0;
    wv = tid >> 6;
    __syncthreads();
    if (wv < 3) {  // D_b = A'_bb - L_ba L_ba^T, one 16x16 block of the lower triangle per wave
        const int mbi = wv == 0 ? 0 : 16, mbj = wv == 2 ? 16 : 0;
        mfma_d4 m0, m1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) m0[reg] = I.T11[(mbi + mg + 4 * reg) * TS + mbj + mc];
#pragma unroll
        for (int t = 0; t < NB / 4; t += 2) {
            const double a0 = I.T10[(mbi + mc) * TS + 4 * t + mg], b0 = I.T10[(mbj + mc) * TS + 4 * t + mg];
            const double a1 = I.T10[(mbi + mc) * TS + 4 * t + 4 + mg], b1 = I.T10[(mbj + mc) * TS + 4 * t + 4 + mg];
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1, m1, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) I.Dn[mbi + mg + 4 * reg][mbj + mc] = m0[reg] + m1[reg];
    }
    __syncthreads();
    if (wv == 0) {
        factor_32(I.Dn, *reinterpret_cast<double (*)[NB * PLD]>(I.Pslab), *I.F, flags);
    } else if (wv == 1) {
        invert_behind_factor(*I.F, nullptr, flags, I.Wb, TS);
    } else {  // P1 = L_ba W_a beside the factor: rows 16 (wv - 2) .., W_a[k][j] = 0 for k < j
        const int bi = wv - 2;
        mfma_d4 p0 = {0.0, 0.0, 0.0, 0.0}, p1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const double a = I.T10[(16 * bi + mc) * TS + 4 * t + mg];
            p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, I.Wa[(4 * t + mg) * WLS + mc], p0, 0, 0, 0);
            if (t >= 4) p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, I.Wa[(4 * t + mg) * WLS + 16 + mc], p1, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            I.P1[(16 * bi + mg + 4 * reg) * TS + mc] = p0[reg];
            I.P1[(16 * bi + mg + 4 * reg) * TS + 16 + mc] = p1[reg];
        }
    }
    __syncthreads();
    {   // X = -W_b P1: one 16x16 block per wave, W_b[i][k] = 0 for k > i; W_b is published in the shadow of the chain
        const int bi = wv >> 1, bj = wv & 1;
        mfma_d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = x;  // two chains: a dependent f64 MFMA waits out the full 64-cycle pass
#pragma unroll
        for (int t = 0; t < 8; t += 2)
            if (t < 4 * (bi + 1)) {
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(-I.Wb[(16 * bi + mc) * TS + 4 * t + mg], I.P1[(4 * t + mg) * TS + 16 * bj + mc], x, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-I.Wb[(16 * bi + mc) * TS + 4 * t + 4 + mg], I.P1[(4 * t + 4 + mg) * TS + 16 * bj + mc], x2, 0, 0, 0);
            }
        publish_w<TPB>(I.Wb, TS, Wout_b, Wop_b, tid);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * bi + mg + 4 * reg, col = 16 * bj + mc;
            const double v = x[reg] + x2[reg];
            Xop[xop_index(row, col)] = v;
            Xrow[row * NB + col] = v;
        }
    }
}

// ---- the sparse, level-scheduled factorisation (plan: ba_plan.hpp) ------------------------------------------------
// S lives as 64x64 tiles of the symbolic pattern of its Cholesky factor (cameras permuted and laid out in panels by the
// plan; padding columns carry an identity diagonal, the right-hand side is row 63 of the root panel). Per panel P the
// factorisation leaves W64_P = [[W_a, 0], [X, W_b]] = (factor of the fully updated diagonal tile)^-1 in Winv (W_a, W_b
// row-major), Xrow and Wops (operand order); the off-diagonal tiles end as the RAW updated strips A'[I][P] — the factor
// strips L[I][P] = A'[I][P] W64_P^T are formed on the fly by every consumer and never stored: the back-substitution
// works on the raw tiles too (sp_backsolve).

// Launch 0: the diagonal tiles of the leaves of the elimination tree (nothing updates them), one workgroup each.
__global__ __launch_bounds__(TPB) void sp_diag(const double* __restrict__ T, const int* __restrict__ leaves, const int* __restrict__ diag_tile,
                                               double* __restrict__ Winv, double* __restrict__ Wops, double* __restrict__ Xrow,
                                               int* __restrict__ flags) {
    __shared__ double Dn[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Pslab[NB * PLD];
    __shared__ __attribute__((aligned(16))) FactorImage Fimg;
    __shared__ double Wa[NB * WLS], T10[NB * TS], T11[NB * TS], P1[NB * TS], Wb[NB * TS];
    const int tid = threadIdx.x;
    const int P = leaves[blockIdx.x];
    const double* __restrict__ A = T + (size_t)diag_tile[P] * TILE;
    double raw[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) {  // the three quadrants of the lower triangle, all loads in flight together
        const int q = m >> 2, idx = tid + TPB * (m & 3), i = idx / NB + (q > 0 ? NB : 0), j = idx % NB + (q == 2 ? NB : 0);
        raw[m] = A[i * PB + j];
    }
#pragma unroll
    for (int m = 0; m < 12; ++m) {
        const int q = m >> 2, idx = tid + TPB * (m & 3), i = idx / NB, j = idx % NB;
        if (q == 0) Dn[i][j] = raw[m];
        else if (q == 1) T10[i * TS + j] = raw[m];
        else T11[i * TS + j] = raw[m];
    }
    if (tid == 0) Fimg.progress = 0;
    __syncthreads();
    if (tid < 64) factor_32(Dn, Pslab, Fimg, flags, blockIdx.x == 0);
    else if (tid < 128) invert_behind_factor(Fimg, nullptr, flags, Wa, WLS);
    __syncthreads();
    if (tid >= 128) publish_w<128>(Wa, WLS, Winv + (size_t)(2 * P) * NB * NB, Wops + (size_t)P * TILE_OPS, tid - 128);  // (waves 0-1 form L_ba next)
    DiagImages I{Dn, Pslab, &Fimg, Wa, T10, T11, P1, Wb};
    diag_second_half(I, Winv + (size_t)(2 * P + 1) * NB * NB, Wops + (size_t)P * TILE_OPS + WOP, Wops + (size_t)P * TILE_OPS + 2 * WOP,
                     Xrow + (size_t)P * NB * NB, flags);
}

// [L_a L_b] = [A_a A_b] W64^T for the wave's 16 rows of a strip, in place: every operand read of the strip precedes the
// writes in the wave's LDS order. wa / wb: W_a / W_b in operand order (wop_index), xp: X (xop_index).
__device__ __forceinline__ void panel64(double (*strip)[LS2], int wv, int mc, int mg, const double (&wa)[12], const double (&wb)[12],
                                        const double (&xp)[16]) {
    double a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = strip[16 * wv + mc][4 * q + mg];
    mfma_d4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = x0, x2 = x0, y0 = x0, y1 = x0, y2 = x0, y3 = x0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], wa[q], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], wa[4 + q], x1, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 + q], wa[8 + q], x2, 0, 0, 0);
        y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[8 + q], wb[q], y1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], xp[q], y0, 0, 0, 0);
        y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], xp[8 + q], y2, 0, 0, 0);
        y3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[8 + q], wb[4 + q], y3, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        double* row = strip[16 * wv + mg + 4 * reg];
        row[mc] = x0[reg];
        row[16 + mc] = x1[reg] + x2[reg];
        row[32 + mc] = y0[reg] + y1[reg];
        row[48 + mc] = y2[reg] + y3[reg];
    }
}

// Launch l + 1: the updates of every source panel J of level l, one workgroup per TARGET tile (I1, I2):
//   A[I1][I2] -= sum_J L[I1][J] L[I2][J]^T,   L[I][J] = A'[I][J] W64_J^T formed from the raw strip and J's operands,
// the sources in the plan's order (ascending J: a fixed summation order, no atomics; a tile is written by exactly one
// workgroup of a launch and read as a strip only in the launch of its own column's level). A diagonal target (P, P)
// whose panel P has level l + 1 is complete after its last source: that workgroup goes on and factorises it ->
// W_a, W_b, X of P (the chain of the launch: these items are dispatched first). Its tile is not written back: nothing
// reads a diagonal tile after its factorisation.
template <bool FIRST>
__device__ __forceinline__ void sp_level_item(double* lds, double* __restrict__ T, const BaPlanItem* __restrict__ items,
                                              const BaPlanSrc* __restrict__ srcs, double* __restrict__ Winv, double* __restrict__ Wops,
                                              double* __restrict__ Xrow, int* __restrict__ flags) {
    double (*Li)[LS2] = reinterpret_cast<double (*)[LS2]>(lds);
    double (*Lj)[LS2] = reinterpret_cast<double (*)[LS2]>(lds + 64 * LS2);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, mc = tid & 15, mg = (tid >> 4) & 3;
    const BaPlanItem it = items[blockIdx.x];
    const bool first = FIRST;                          // a diagonal target that this workgroup factorises after its last source
    const bool same = FIRST || (it.flags & 1) != 0;    // a diagonal target: both strips are the same rows, one is formed
    double* __restrict__ A = T + (size_t)it.tgt * TILE;
    const int mbi = wv == 0 ? 0 : 16, mbj = wv == 2 ? 16 : 0;  // first quadrant of a factorising item: blocks (0,0), (1,0), (1,1)
    const bool block_thread = !first || wv >= 2;  // a wave owns 16 rows of the tile and its four 16-column blocks
    mfma_d4 old[4];
    if (block_thread) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                old[cb][reg] = (same && cb > wv) ? 0.0 : A[(16 * wv + mg + 4 * reg) * PB + 16 * cb + mc];
    }
    mfma_d4 mold = {0.0, 0.0, 0.0, 0.0};
    if (first && wv < 3) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) mold[reg] = A[(mbi + mg + 4 * reg) * PB + mbj + mc];
    }
    // a target that takes more sources than its window allows has shadow accumulators (ba_plan.hpp): other workgroups summed
    // part of its updates into them in earlier launches; they join the target here, in the launch of its deadline
    for (int k = 0; k < it.nshadow; ++k) {
        const double* __restrict__ Sh = T + (size_t)(it.shadow0 + k) * TILE;
        if (block_thread) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (!(same && cb > wv)) old[cb][reg] += Sh[(16 * wv + mg + 4 * reg) * PB + 16 * cb + mc];
        }
        if (first && wv < 3) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) mold[reg] += Sh[(mbi + mg + 4 * reg) * PB + mbj + mc];
        }
    }
#pragma unroll 1
    for (int s = 0; s < it.nsrc; ++s) {
        const BaPlanSrc sr = srcs[it.src0 + s];
        // Every global read of a source is issued up front, none is predicated. A wave fetches ITS 16 rows of each raw
        // strip, two whole rows (2 x 512 bytes) per instruction, and is the only reader of their LDS image: no
        // workgroup barrier before the panel product.
        const double* __restrict__ Si = T + (size_t)sr.tile_i * TILE;
        const double* __restrict__ Sj = T + (size_t)sr.tile_j * TILE;
        double2 pi[8], pj[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) pi[m] = *reinterpret_cast<const double2*>(&Si[(16 * wv + 2 * m + (lane >> 5)) * PB + 2 * (lane & 31)]);
        if (!FIRST) {  // (a diagonal target fetches the same rows twice: a conditionally initialised array would live in scratch)
#pragma unroll
            for (int m = 0; m < 8; ++m) pj[m] = *reinterpret_cast<const double2*>(&Sj[(16 * wv + 2 * m + (lane >> 5)) * PB + 2 * (lane & 31)]);
        }
        // the B operands of the panel product, stored in operand order by the producer: one coalesced load per k-step
        const double* __restrict__ ops = Wops + (size_t)sr.J * TILE_OPS;
        double wa[12], wb[12], xp[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < 6) {
                const double2 u = *reinterpret_cast<const double2*>(&ops[op_index(2 * q, lane)]), v = *reinterpret_cast<const double2*>(&ops[WOP + op_index(2 * q, lane)]);
                wa[2 * q] = u.x, wa[2 * q + 1] = u.y, wb[2 * q] = v.x, wb[2 * q + 1] = v.y;
            }
            const double2 u = *reinterpret_cast<const double2*>(&ops[2 * WOP + op_index(2 * q, lane)]);
            xp[2 * q] = u.x, xp[2 * q + 1] = u.y;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) *reinterpret_cast<double2*>(&Li[16 * wv + 2 * m + (lane >> 5)][2 * (lane & 31)]) = pi[m];
        if (!FIRST) {
#pragma unroll
            for (int m = 0; m < 8; ++m) *reinterpret_cast<double2*>(&Lj[16 * wv + 2 * m + (lane >> 5)][2 * (lane & 31)]) = pj[m];
        }
        wave_lds_sync();
        panel64(Li, wv, mc, mg, wa, wb, xp);
        if (!same) panel64(Lj, wv, mc, mg, wa, wb, xp);
        __syncthreads();
        if (first) {
            if (s + 1 == it.nsrc) break;  // the last source: its updates are interleaved with the factorisation below
            if (wv < 3) {  // first quadrant, kept in registers
                mfma_d4 m0 = mold, m1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < PB / 4; q += 2) {
                    const double a0 = Li[mbi + mc][4 * q + mg], b0 = Li[mbj + mc][4 * q + mg];
                    const double a1 = Li[mbi + mc][4 * q + 4 + mg], b1 = Li[mbj + mc][4 * q + 4 + mg];
                    m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0, m0, 0, 0, 0);
                    m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1, m1, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) mold[reg] = m0[reg] + m1[reg];
            }
            if (wv >= 2) {  // rows 32..63
#pragma unroll
                for (int q = 0; q < PB / 4; ++q) {
                    const double av = -Li[16 * wv + mc][4 * q + mg];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        if (cb <= wv) old[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Li[16 * cb + mc][4 * q + mg], old[cb], 0, 0, 0);
                }
            }
            __syncthreads();  // every read of this source's strips is complete (no array of loads is live across this barrier)
            continue;
        }
        // A_ij -= L_i L_j^T over the 64 panel columns: the four 16-column blocks of the wave's rows as four chains
        double (*LjE)[LS2] = same ? Li : Lj;
#pragma unroll
        for (int q = 0; q < PB / 4; ++q) {
            const double av = -Li[16 * wv + mc][4 * q + mg];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                if (!(same && cb > wv)) old[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, LjE[16 * cb + mc][4 * q + mg], old[cb], 0, 0, 0);
        }
        if (s + 1 < it.nsrc) __syncthreads();  // every read of this source's strips is complete before the next one's land
    }
    if (!first) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (!(same && cb > wv)) A[(16 * wv + mg + 4 * reg) * PB + 16 * cb + mc] = old[cb][reg];
        return;
    }
    // ---- the factorising item, behind the barrier of its last source's panel product ----
    // images of the factorisation: second strip's region (a diagonal target never writes it)
    double* base = lds + 64 * LS2;
    double (*Dn)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(base);
    double* Pslab = base + NB * (NB + 1);  // 1056 doubles in: 16-byte aligned
    FactorImage* F = reinterpret_cast<FactorImage*>(Pslab + NB * PLD);
    double* Wal = reinterpret_cast<double*>(F) + (sizeof(FactorImage) + 7) / 8 + 1;
    static_assert(NB * (NB + 1) % 2 == 0 && (NB * PLD) % 2 == 0, "Pslab and the factor image stay 16-byte aligned");
    static_assert(NB * (NB + 1) + NB * PLD + (sizeof(FactorImage) + 7) / 8 + 1 + NB * WLS <= 64 * LS2, "the images fit the second strip");
    static_assert(4 * NB * TS <= 64 * LS2, "the images of the second half fit the first strip");
    DiagImages I{Dn, Pslab, F, Wal, lds, lds + NB * TS, lds + 2 * NB * TS, lds + 3 * NB * TS};
    const int P = it.panel;
    if (tid == 0) F->progress = 0;
    if (wv < 3) {  // first quadrant: A_00 - L_0 L_0^T over the 64 panel columns
        mfma_d4 m0 = mold, m1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < PB / 4; q += 2) {
            const double a0 = Li[mbi + mc][4 * q + mg], b0 = Li[mbj + mc][4 * q + mg];
            const double a1 = Li[mbi + mc][4 * q + 4 + mg], b1 = Li[mbj + mc][4 * q + 4 + mg];
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1, m1, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Dn[mbi + mg + 4 * reg][mbj + mc] = m0[reg] + m1[reg];
    }
    __syncthreads();
    if (wv == 0) {
        factor_32(Dn, *reinterpret_cast<double (*)[NB * PLD]>(Pslab), *F, flags);
    } else if (wv == 1) {
        invert_behind_factor(*F, nullptr, flags, Wal, WLS);
    } else {  // rows 32..63 of the tile beside the factor; the results stay in registers until the strip is dead
#pragma unroll
        for (int q = 0; q < PB / 4; ++q) {
            const double av = -Li[16 * wv + mc][4 * q + mg];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                if (cb <= wv) old[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Li[16 * cb + mc][4 * q + mg], old[cb], 0, 0, 0);
        }
    }
    __syncthreads();  // the strip is dead: W_a's copy, the factor of block a and every read of Li are complete
    if (wv >= 2) {  // the wave's 16 rows of A'_ba (own rows of T10: wave-local) and of A'_bb
        const int r0 = 16 * (wv - 2);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = r0 + mg + 4 * reg, col = 16 * (cb & 1) + mc;
                if (cb < 2) I.T10[row * TS + col] = old[cb][reg];
                else I.T11[row * TS + col] = old[cb][reg];  // (block (2, 3) of wave 2 is the zero it was loaded as: never read)
            }
        wave_lds_sync();
    } else {  // waves 0-1 are idle until D_b: they publish W_a
        publish_w<128>(Wal, WLS, Winv + (size_t)(2 * P) * NB * NB, Wops + (size_t)P * TILE_OPS, tid);
    }
    diag_second_half(I, Winv + (size_t)(2 * P + 1) * NB * NB, Wops + (size_t)P * TILE_OPS + WOP, Wops + (size_t)P * TILE_OPS + 2 * WOP,
                     Xrow + (size_t)P * NB * NB, flags, 2);
}

// The factorising items of a launch come first in its item list (n_first of them): the two roles are separate code paths
// of one kernel (as two kernels the second would wait for the first: a launch more on the chain of every level).
__global__ __launch_bounds__(TPB, 2) void sp_level(double* __restrict__ T, const BaPlanItem* __restrict__ items, const BaPlanSrc* __restrict__ srcs,
                                                   int n_first, double* __restrict__ Winv, double* __restrict__ Wops,
                                                   double* __restrict__ Xrow, int* __restrict__ flags) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 64 * LS2];
    if ((int)blockIdx.x < n_first) sp_level_item<true>(lds, T, items, srcs, Winv, Wops, Xrow, flags);
    else sp_level_item<false>(lds, T, items, srcs, Winv, Wops, Xrow, flags);
}

// Back-substitution on the elimination tree, ONE launch: workgroup b solves panel J = order[b] (root first, a panel
// after its parent) and the panels hand their solutions down through global memory with a flag per panel instead of
// kernel boundaries. With the right-hand side carried as row 63 of the root, z = [x; -1] solves L_aug^T z = -delta e_63:
//   root:   z = -W[63][:] / W[63][63]                    (W = W64 of the root, delta = its last pivot)
//   J:      t = -sum_{I in struct(J)} A'[I][J]^T z_I     (the RAW strips: L[I][J] = A'[I][J] W_J^T is never stored; the
//                                                          root's row 63 times z_root[63] = -1 brings in the rhs)
//           z_J = W_J^T (W_J t)                          (= D_J^-1 t, D_J the fully updated diagonal tile)
// The chain is one hand-off per level of the tree, and only the PARENT's strip sits on it: a workgroup consumes its
// strips from the far end (root side) as their z appear, so what is left when the parent publishes is one poll, eight
// loads, one strip of multiply-adds (its values prefetched), a cross-wave sum, two 64x64 triangular products and the
// publication. Protocol as in cdna_hip_programming.md section 6 (write-through form): z_J is stored by ONE wave with
// agent-scope atomic stores (sc1), that wave drains (vmcnt(0)) and its lane 0 sets the flag; every consuming wave polls
// the flag itself with an agent-scope atomic and reads z with agent-scope atomic loads (they bypass its CU's L1) — no
// fences, no workgroup barrier per strip. A wave whose wait expires raises flags[0] bit 2 (the solve counts as failed,
// EACHAM_ERR_HIP) and goes on: no wave can spin forever. The low workgroup ids go to the producers (the hardware
// dispatches in ascending order), so a consumer never occupies a CU its producer still needs.
constexpr int BS_THREADS = 512;  // eight waves: wave g owns rows 8 g .. 8 g + 7 of every strip
constexpr int BS_PF = 4;         // strips whose values are in flight ahead of the one being consumed
__global__ __launch_bounds__(BS_THREADS) void sp_backsolve(const double* __restrict__ T, const int* __restrict__ order, const int* __restrict__ ptr,
                                                           const int2* __restrict__ ent, const double* __restrict__ Winv,
                                                           const double* __restrict__ Xrow, double* __restrict__ z,
                                                           const int* __restrict__ col_dest, double* __restrict__ delta_c,
                                                           int* __restrict__ flags) {
    __shared__ double Wl[PB][PB + 1];
    __shared__ double part[BS_THREADS / PB][PB];
    __shared__ double tl[PB], ul[PB], zl[PB];
    const int tid = threadIdx.x, j = tid & (PB - 1), g = tid >> 6;
    constexpr int RPT = PB / (BS_THREADS / PB);  // rows of a strip per thread: 8
    const int b = blockIdx.x, J = order[b], e0 = ptr[b], e1 = ptr[b + 1];
    int* handoff = flags + N_STATUS;
    {   // W64_J = [[W_a, 0], [X, W_b]]
        const double* __restrict__ Wa = Winv + (size_t)(2 * J) * NB * NB;
        const double* __restrict__ Wb = Wa + NB * NB;
        const double* __restrict__ X = Xrow + (size_t)J * NB * NB;
        for (int idx = tid; idx < PB * PB; idx += BS_THREADS) {
            const int r = idx >> 6, c = idx & 63;
            Wl[r][c] = r < NB ? (c < NB ? Wa[r * NB + c] : 0.0) : (c < NB ? X[(r - NB) * NB + c] : Wb[(r - NB) * NB + c - NB]);
        }
    }
    if (e0 == e1) {  // the root: it waits for nobody
        __syncthreads();
        if (tid < PB) zl[tid] = -Wl[PB - 1][tid] / Wl[PB - 1][PB - 1];
    } else {
        // The strips of column J are listed nearest ancestor first; they are consumed from the FAR end: those ancestors
        // published long ago, so everything but the parent's strip is summed in the shadow of the chain. A wave polls
        // the hand-off flags itself (no workgroup barrier per strip) and reads the eight z it needs straight from
        // global memory; the strip values (independent of any z) run BS_PF strips ahead.
        double pre[BS_PF][RPT];
        auto fetch = [&](int e, double (&v)[RPT]) {
            const double* __restrict__ A = T + (size_t)ent[e].x * TILE;
#pragma unroll
            for (int r = 0; r < RPT; ++r) v[r] = A[(RPT * g + r) * PB + j];
        };
#pragma unroll
        for (int p = 0; p < BS_PF; ++p)
            if (e1 - 1 - p >= e0) fetch(e1 - 1 - p, pre[p]);
        double acc = 0.0;
        bool late = false;
        for (int base = e1 - 1; base >= e0; base -= BS_PF) {
#pragma unroll
            for (int p = 0; p < BS_PF; ++p) {
                const int e = base - p;
                if (e < e0) break;
                const int I = ent[e].y;
                int spins = 0;
                while (__hip_atomic_load(&handoff[I], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    if (++spins > (1 << 22)) {
                        late = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the loads below the poll
                double zr[RPT];
#pragma unroll
                for (int r = 0; r < RPT; ++r) zr[r] = __hip_atomic_load(&z[(size_t)I * PB + RPT * g + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int r = 0; r < RPT; ++r) acc += pre[p][r] * zr[r];
                if (e - BS_PF >= e0) fetch(e - BS_PF, pre[p]);
            }
        }
        if (late && (tid & 63) == 0) atomicOr(flags, 4);
        part[g][j] = acc;
        __syncthreads();
        if (tid < PB) {
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < BS_THREADS / PB; ++q) tot += part[q][tid];
            tl[tid] = -tot;
        }
        __syncthreads();
        {   // u = W t: output i on 8 lanes, eight terms per lane (W is lower triangular: the rest are stored zeros)
            const int i = tid >> 3, q = tid & 7;
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += Wl[i][8 * q + k] * tl[8 * q + k];
            v = dpp_row_sum<8>(v);
            if (q == 7) ul[i] = v;
        }
        __syncthreads();
        {   // z = W^T u
            const int c = tid >> 3, q = tid & 7;
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += Wl[8 * q + k][c] * ul[8 * q + k];
            v = dpp_row_sum<8>(v);
            if (q == 7) zl[c] = v;
        }
    }
    __syncthreads();
    // publication by ONE wave: its stores drain (vmcnt(0)) before its lane 0 raises the flag — no workgroup barrier
    if (tid < PB) {
        __hip_atomic_store(&z[(size_t)J * PB + tid], zl[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int dst = col_dest[J * PB + tid];
        if (dst >= 0) delta_c[dst] = zl[tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && !(BA_FAULT(1) && e0 == e1)) __hip_atomic_store(&handoff[J], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- K-F / K-G: the tail of a tryLambda() in ONE launch ----------------------------------------------------------
// landmark back-substitution + tentative points + linearised-cost terms, the retraction of the cameras and K with their
// prior errors, the nonlinear error at the tentative values, and the final sums:
//   delta_l = Linv^T (gt - sum_o Et_o^T dc[c_o] - EKt^T dK);  lin += 1/2 (dl.gl + lambda dl.D dl)
//   error   = sum_o rho(|r_o(x (+) dc, l + dl, K + dK)|) + priors
// Until round 3 these were three dependent launches (back-substitution + retraction, error pass over the retracted poses,
// one-block final sums): 34 us of 0.48 ms on S200, 16 of 118 us on a local window. The error pass only waited for
// pose_new — so a landmark's thread retracts the pose of each of its observations' cameras itself (the same instructions on
// the same inputs as the retraction workgroups, which still write pose_new / K_new for the next linearisation) — and the
// final sums are made by whichever workgroup of the launch finishes LAST. The hand-over is fence-free like the
// back-substitution's: partial sums are stored write-through (agent-scope atomic stores), the storing wave drains,
// one lane takes a ticket from an agent-scope counter; the workgroup that draws the last ticket reads every partial with
// agent-scope atomic loads and adds them in index order — the same order whichever workgroup that is: the sums stay
// deterministic. The counter resets itself.
__device__ __forceinline__ void retract_pose(const double* x, const double* d, double* y) {  // x (+) [omega, v], first-order chart (Cayley)
    double C[9];
    cayley(d, C);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) y[3 * i + j] = x[3 * i] * C[j] + x[3 * i + 1] * C[3 + j] + x[3 * i + 2] * C[6 + j];
    for (int i = 0; i < 3; ++i) y[9 + i] = x[9 + i] + x[3 * i] * d[3] + x[3 * i + 1] * d[4] + x[3 * i + 2] * d[5];
}

__device__ __forceinline__ void store_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_wt(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Every thread of the workgroup calls it after the workgroup's write-through stores; true in the workgroup that arrives last.
__device__ __forceinline__ bool last_workgroup(int* counter) {
    __shared__ int last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partials have left for memory
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = t == (int)gridDim.x - 1;
        if (last_flag) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return last_flag != 0;
}

// ---- K-G1: retract cameras + K, pose/K prior errors and their linearised-cost terms (thread = camera) --
// err_cam[c] = prior error at the NEW pose; lin_cam[c] = 1/2 (dc.gc + lambda dc.D dc); slot nc = K.
__device__ __forceinline__ void retract_camera(const BaDev& D, int c, double lambda, const double* pose_in, double* pose_out,
                                               const double* K_in, double* K_out, int apply_delta, double* err_cam,
                                               double* lin_cam) {
    if (c < D.nc) {
        const double* x = pose_in + 12 * (size_t)c;
        double y[12];
        double lin = 0.0;
        if (apply_delta) {
            const double* d = D.delta_c + 6 * (size_t)c;
            retract_pose(x, d, y);
            const double* cl = D.camlin + (size_t)CAMLIN * c;
            for (int a = 0; a < 6; ++a) lin += 0.5 * d[a] * cl[66 + a] + 0.5 * lambda * clampd(cl[7 * a], 1e-6, 1e32) * d[a] * d[a];
        } else {
            for (int k = 0; k < 12; ++k) y[k] = x[k];
        }
        for (int k = 0; k < 12; ++k) pose_out[12 * (size_t)c + k] = y[k];
        double xi[6], n2 = 0.0;
        pose_local(y, D.pose0 + 12 * (size_t)c, xi);
        const bool fx = D.fixed[c] != 0;
        const double* sg = fx ? D.nz.fixed_sigma : D.nz.pose_sigma;
        for (int k = 0; k < 6; ++k) n2 += (xi[k] / sg[k]) * (xi[k] / sg[k]);
        store_wt(&err_cam[c], fx ? 0.5 * n2 : huber_loss(sqrt(n2), D.nz.pose_huber));
        store_wt(&lin_cam[c], lin);
    } else if (c == D.nc) {
        double e = 0.0, lin = 0.0;
        for (int a = 0; a < 5; ++a) {
            const double d = apply_delta ? D.delta_c[6 * D.nc + a] : 0.0;
            const double kn = K_in[a] + d;
            K_out[a] = kn;
            const double w = (kn - D.K0[a]) / D.nz.k_sigma[a];
            e += 0.5 * w * w;
            if (apply_delta) lin += 0.5 * d * D.klin[25 + a] + 0.5 * lambda * clampd(D.klin[6 * a], 1e-6, 1e32) * d * d;
        }
        store_wt(&err_cam[D.nc], e);
        store_wt(&lin_cam[D.nc], lin);
    }
}
__global__ void ba_retract_cameras(BaDev D, double lambda, const double* pose_in, double* pose_out,
                                   const double* K_in, double* K_out, int apply_delta, double* err_cam,
                                   double* lin_cam) {
    retract_camera(D, blockIdx.x * blockDim.x + threadIdx.x, lambda, pose_in, pose_out, K_in, K_out, apply_delta, err_cam, lin_cam);
}

// ---- K-G3: fixed-order final sums -> scal[0] = error, scal[1] = linearised cost change; one workgroup, all its threads ----
// n_err / n_lin = workgroups that left a partial in err_part / lin_part (n_lin = 0: no linearised-cost term in this pass)
__device__ __forceinline__ void final_sums(const BaDev& D, const double* err_cam, const double* lin_cam, int n_err, int n_lin,
                                           double ticket, double* sm /* [(TPB / 64) * 2] */) {
    double v[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < n_err; i += TPB) v[0] += load_wt(&D.err_part[i]);
    for (int i = threadIdx.x; i < n_lin; i += TPB) v[1] += load_wt(&D.lin_part[i]);
    for (int i = threadIdx.x; i <= D.nc; i += TPB) {
        v[0] += load_wt(&err_cam[i]);
        if (n_lin) v[1] += load_wt(&lin_cam[i]);
    }
    block_sum<2>(v, sm);
    // The status word and the hand-off flags of the back-substitution are consumed here, so they are also cleared here
    // for the next tryLambda(), and the three scalars go to pinned host memory as well: a memset node and a
    // device-to-host copy less on the chain of every LM inner iteration.
    if (threadIdx.x == 0) {
        const double st = (double)D.flags[0];
        D.scal[0] = v[0];
        D.scal[1] = v[1];
        D.scal[2] = st;
        if (D.scal_pinned) {  // the host polls slot 3 for this launch's ticket (read_scal): the values first, fenced
            D.scal_pinned[0] = v[0];
            D.scal_pinned[1] = v[1];
            D.scal_pinned[2] = st;
            __threadfence_system();
            *(volatile double*)&D.scal_pinned[3] = ticket;
        }
    }
    for (int k = threadIdx.x; k < N_STATUS + D.sp_npan; k += TPB) D.flags[k] = 0;  // (thread 0 read flags[0] above, in program order)
}

// nonlinear error of a landmark's reprojection factors (this lane's share of them) + its prior (lane sub == 0).
// RETRACT: the poses are D.pose (+) delta_c, retracted here; else `pose` as given.
template <int LPL, bool RETRACT>
__device__ __forceinline__ double landmark_error(const BaDev& D, int j, int sub, const double* __restrict__ pose, const double (&l)[3],
                                                 const double (&K)[5]) {
    const int o0 = D.lm_ptr[j], o1 = D.lm_ptr[j + 1];
    double e = 0.0;
    for (int o = o0 + sub; o < o1; o += LPL) {
        const int cam = (int)D.obs_cam[o];
        const double* x = pose + 12 * (size_t)cam;
        double xr[12], r[2];
#pragma unroll
        for (int k = 0; k < 12; ++k) xr[k] = x[k];
        if (RETRACT) {
            double y[12];
            retract_pose(xr, D.delta_c + 6 * (size_t)cam, y);
#pragma unroll
            for (int k = 0; k < 12; ++k) xr[k] = y[k];
        }
        reproj<false>(xr, l, K, D.obs_uv[2 * (size_t)o], D.obs_uv[2 * (size_t)o + 1], r, nullptr, nullptr, nullptr);
        e += huber_loss(sqrt(r[0] * r[0] + r[1] * r[1]) / D.nz.pix_sigma, D.nz.pix_huber);
    }
    if (sub == 0) {  // the landmark's prior, once
        const double sg = D.lmprior[2 * j], kh = D.lmprior[2 * j + 1];
        double n2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double w = (l[a] - D.pt0[3 * (size_t)j + a]) / sg;
            n2 += w * w;
        }
        e += huber_loss(sqrt(n2), kh);
    }
    return e;
}

// The tail of a tryLambda(). Grid = the landmark workgroups, then the workgroups that retract the cameras and K.
template <int LPL>
__global__ __launch_bounds__(TPB) void ba_step_landmarks(BaDev D, double lambda, double* err_cam, double* lin_cam, double ticket) {
    __shared__ double sm[(TPB / 64) * 2];
    if ((int)blockIdx.x >= D.n_step_blocks) {
        retract_camera(D, ((int)blockIdx.x - D.n_step_blocks) * TPB + threadIdx.x, lambda, D.pose, D.pose_new, D.Kc, D.K_new, 1, err_cam, lin_cam);
    } else {
        const int gid = blockIdx.x * TPB + threadIdx.x, j = gid / LPL, sub = gid % LPL;
        double lin[1] = {0.0}, dl[3] = {0.0, 0.0, 0.0};
        if (j < D.nl) {
            const int o0 = D.lm_ptr[j], o1 = D.lm_ptr[j + 1];
            double d0 = 0.0, d1 = 0.0, d2 = 0.0;
            if (o1 > o0) {
                const double* lt = D.lmtry + (size_t)LMLIN * j;
                double t0 = lt[6], t1 = lt[7], t2 = lt[8];
                const double* dK = D.delta_c + 6 * D.nc;
    #pragma unroll
                for (int a = 0; a < 5; ++a) {
                    t0 -= lt[9 + 3 * a] * dK[a];
                    t1 -= lt[10 + 3 * a] * dK[a];
                    t2 -= lt[11 + 3 * a] * dK[a];
                }
                // sum_o Et_o^T dc = Linv sum_o E_o^T dc and E_o^T dc = Al^T (Ap dc): the Jacobians are recomputed at the
                // linearisation point (pose / pt / Kc do not change inside the lambda loop) from this landmark's own
                // observation records instead of gathering its ten 144-byte Et rows out of the camera-ordered array
                // (137 MB of HBM traffic per try, profiles/r02_pmc_ba_traffic.json of the first round-2 build)
                const double l[3] = {D.pt[3 * (size_t)j], D.pt[3 * (size_t)j + 1], D.pt[3 * (size_t)j + 2]};
                double K[5];
    #pragma unroll
                for (int k = 0; k < 5; ++k) K[k] = D.Kc[k];
                double w0 = 0.0, w1 = 0.0, w2 = 0.0;
                for (int o = o0 + sub; o < o1; o += LPL) {
                    const int cam = (int)D.obs_cam[o];
                    const double* x = D.pose + 12 * (size_t)cam;
                    const double* dc = D.delta_c + 6 * (size_t)cam;
                    double xr[12];
    #pragma unroll
                    for (int k = 0; k < 12; ++k) xr[k] = x[k];
                    double Ap[12], Al[6], Ak[10], b[2];
                    obs_factor(xr, l, K, D.obs_uv[2 * (size_t)o], D.obs_uv[2 * (size_t)o + 1], D.nz.pix_sigma, D.nz.pix_huber, Ap, Al, Ak, b);
                    double u0 = 0.0, u1 = 0.0;
    #pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        u0 += Ap[a] * dc[a];
                        u1 += Ap[6 + a] * dc[a];
                    }
                    w0 += Al[0] * u0 + Al[3] * u1;
                    w1 += Al[1] * u0 + Al[4] * u1;
                    w2 += Al[2] * u0 + Al[5] * u1;
                }
                if (LPL > 1) w0 = group_sum<LPL>(w0), w1 = group_sum<LPL>(w1), w2 = group_sum<LPL>(w2);  // (a landmark's lanes take this branch together)
                const double m00 = lt[0], m10 = lt[1], m11 = lt[2], m20 = lt[3], m21 = lt[4], m22 = lt[5];
                t0 -= m00 * w0;
                t1 -= m10 * w0 + m11 * w1;
                t2 -= m20 * w0 + m21 * w1 + m22 * w2;
                d0 = m00 * t0 + m10 * t1 + m20 * t2;
                d1 = m11 * t1 + m21 * t2;
                d2 = m22 * t2;
                const double* in = D.lmlin + (size_t)LMLIN * j;
                if (sub == 0)
                    lin[0] = 0.5 * (d0 * in[6] + d1 * in[7] + d2 * in[8]) +
                             0.5 * lambda * (clampd(in[0], 1e-6, 1e32) * d0 * d0 + clampd(in[3], 1e-6, 1e32) * d1 * d1 +
                                             clampd(in[5], 1e-6, 1e32) * d2 * d2);
            }
            dl[0] = d0, dl[1] = d1, dl[2] = d2;
            if (sub == 0) {
            D.delta_l[3 * (size_t)j] = d0;
            D.delta_l[3 * (size_t)j + 1] = d1;
            D.delta_l[3 * (size_t)j + 2] = d2;
            D.pt_new[3 * (size_t)j] = D.pt[3 * (size_t)j] + d0;
            D.pt_new[3 * (size_t)j + 1] = D.pt[3 * (size_t)j + 1] + d1;
            D.pt_new[3 * (size_t)j + 2] = D.pt[3 * (size_t)j + 2] + d2;
            }
        }
        double v[2] = {0.0, lin[0]};
        if (j < D.nl && D.lm_ptr[j + 1] > D.lm_ptr[j]) {
            const double lnew[3] = {D.pt[3 * (size_t)j] + dl[0], D.pt[3 * (size_t)j + 1] + dl[1], D.pt[3 * (size_t)j + 2] + dl[2]};
            double Kn[5];
#pragma unroll
            for (int a = 0; a < 5; ++a) Kn[a] = D.Kc[a] + D.delta_c[6 * D.nc + a];
            v[0] = landmark_error<LPL, true>(D, j, sub, D.pose, lnew, Kn);
        }
        block_sum<2>(v, sm);
        if (threadIdx.x == 0) {
            store_wt(&D.err_part[blockIdx.x], v[0]);
            store_wt(&D.lin_part[blockIdx.x], v[1]);
        }
    }
    if (last_workgroup(D.sync_counter)) final_sums(D, err_cam, lin_cam, D.n_step_blocks, D.n_step_blocks, ticket, sm);
}

// ---- K-G2: nonlinear error at given values (the first error pass of a solve, DogLeg's candidates, the PCG path): the
// reprojection + landmark-prior factors (thread = landmark); the camera / K prior errors come from ba_retract_cameras,
// launched before it; the last workgroup makes the final sums.
template <int LPL>
__global__ __launch_bounds__(TPB) void ba_error_landmarks(BaDev D, const double* __restrict__ pose, const double* __restrict__ pt,
                                                          const double* __restrict__ Kc, const double* err_cam, const double* lin_cam,
                                                          int n_lin, double ticket) {
    __shared__ double sm[(TPB / 64) * 2];
    const int gid = blockIdx.x * TPB + threadIdx.x, j = gid / LPL, sub = gid % LPL;
    double e[1] = {0.0};
    if (j < D.nl && D.lm_ptr[j + 1] > D.lm_ptr[j]) {
        const double l[3] = {pt[3 * (size_t)j], pt[3 * (size_t)j + 1], pt[3 * (size_t)j + 2]};
        double K[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) K[k] = Kc[k];
        e[0] = landmark_error<LPL, false>(D, j, sub, pose, l, K);
    }
    block_sum<1>(e, sm);
    if (threadIdx.x == 0) store_wt(&D.err_part[blockIdx.x], e[0]);
    if (last_workgroup(D.sync_counter)) final_sums(D, err_cam, lin_cam, D.n_step_blocks, n_lin, ticket, sm);
}

// ---- the iterative solve the reference can select: PCG + block-Jacobi (BundleAdjuster.cpp:192-200) -----------------
// params.linearSolverType = Iterative, PCGSolverParameters{BlockJacobi, epsilon_abs = epsilon_rel = 1e-10}: GTSAM 4.1.1
// runs preconditioned conjugate gradients on the DAMPED system over ALL variables (poses, calibration, landmarks),
// preconditioned by the Cholesky factors of its diagonal blocks, from x0 = 0, until gamma = r^T M^-1 r <=
// max(epsilon_abs, epsilon_rel^2 gamma0) or 500 iterations (gtsam/linear/PCGSolver.cpp, iterative-inl.h; restated from
// memory like SURVEY.md Appendix A). Here the operator is applied in block form on the linearisation that is already
// resident — y_c = (Hcc + lambda Dc) x_c + HcK x_K + sum_o E_o x_l(o), y_l = (Hll + lambda Dl) x_l + ElK^T x_K +
// sum_o E_o^T x_c(o), y_K likewise — with E kept by the linearisation (store_E). Five launches per iteration, every sum
// in a fixed order (no atomics), the scalars (gamma, alpha, beta, the stop flag) stay on the device and the host looks
// at the flag once per batch of iterations: after convergence the remaining launches of a batch return at once.
constexpr int PCG_MAX_IT = 500;
constexpr int PCG_BATCH = 25;

template <int M>
__device__ __forceinline__ bool spd_inverse(const double* A /* M x M, lower used */, double* inv /* M x M full */) {
    double L[M][M], W[M][M];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        double d = A[j * M + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        ok = ok && d > 0.0;
        const double sd = sqrt(d > 0.0 ? d : 1.0);
        L[j][j] = sd;
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            double v = A[i * M + j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
            L[i][j] = v / sd;
        }
    }
#pragma unroll
    for (int c = 0; c < M; ++c)  // W = L^-1, column by column
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i < c) { W[i][c] = 0.0; continue; }
            double v = i == c ? 1.0 : 0.0;
#pragma unroll
            for (int k = c; k < i; ++k) v -= L[i][k] * W[k][c];
            W[i][c] = v / L[i][i];
        }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {  // inv = W^T W
            double v = 0.0;
#pragma unroll
            for (int k = (i > j ? i : j); k < M; ++k) v += W[k][i] * W[k][j];
            inv[i * M + j] = v;
        }
    return ok;
}

// x = 0, r = g, z = M^-1 r, p = z; partial gamma. thread = landmark
__global__ __launch_bounds__(TPB) void pcg_setup_landmarks(BaDev D, double lambda) {
    __shared__ double sm[(TPB / 64) * 1];
    const int j = blockIdx.x * TPB + threadIdx.x;
    double g[1] = {0.0};
    if (j < D.nl) {
        double r[3] = {0, 0, 0}, z[3] = {0, 0, 0}, Mi[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, dd[3] = {0, 0, 0};
        if (D.lm_ptr[j + 1] > D.lm_ptr[j]) {
            const double* in = D.lmlin + (size_t)LMLIN * j;
            dd[0] = clampd(in[0], 1e-6, 1e32), dd[1] = clampd(in[3], 1e-6, 1e32), dd[2] = clampd(in[5], 1e-6, 1e32);
            const double A[9] = {in[0] + lambda * dd[0], in[1], in[2], in[1], in[3] + lambda * dd[1], in[4], in[2], in[4], in[5] + lambda * dd[2]};
            if (!spd_inverse<3>(A, Mi)) atomicOr(D.flags, 1);
#pragma unroll
            for (int a = 0; a < 3; ++a) r[a] = in[6 + a];
#pragma unroll
            for (int a = 0; a < 3; ++a) z[a] = Mi[3 * a] * r[0] + Mi[3 * a + 1] * r[1] + Mi[3 * a + 2] * r[2];
            g[0] = r[0] * z[0] + r[1] * z[1] + r[2] * z[2];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            D.pcg_Dl[3 * (size_t)j + a] = dd[a];
            D.pcg_rl[3 * (size_t)j + a] = r[a];
            D.pcg_zl[3 * (size_t)j + a] = z[a];
            D.pcg_pl[3 * (size_t)j + a] = z[a];
            D.delta_l[3 * (size_t)j + a] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) D.pcg_Ml[9 * (size_t)j + k] = Mi[k];
    }
    block_sum<1>(g, sm);
    if (threadIdx.x == 0) D.pcg_p3[blockIdx.x] = g[0];
}

// the same for the cameras (thread = camera) and the calibration (thread nc); one block: also the scalars
__global__ __launch_bounds__(1024) void pcg_setup_cameras(BaDev D, double lambda) {
    __shared__ double red[1024 / 64];
    double g = 0.0;
    for (int c = threadIdx.x; c <= D.nc; c += blockDim.x) {
        if (c < D.nc) {
            const double* cl = D.camlin + (size_t)CAMLIN * c;
            double A[36], Mi[36], r[6], z[6];
#pragma unroll
            for (int k = 0; k < 36; ++k) A[k] = cl[k];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const double dd = clampd(cl[7 * a], 1e-6, 1e32);
                D.pcg_Dc[6 * (size_t)c + a] = dd;
                A[7 * a] += lambda * dd;
                r[a] = cl[66 + a];
            }
            if (!spd_inverse<6>(A, Mi)) atomicOr(D.flags, 2);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < 6; ++b) v += Mi[6 * a + b] * r[b];
                z[a] = v;
                g += r[a] * v;
            }
#pragma unroll
            for (int k = 0; k < 36; ++k) D.pcg_Mc[36 * (size_t)c + k] = Mi[k];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                D.pcg_rc[6 * (size_t)c + a] = r[a];
                D.pcg_zc[6 * (size_t)c + a] = z[a];
                D.pcg_pc[6 * (size_t)c + a] = z[a];
                D.delta_c[6 * (size_t)c + a] = 0.0;
            }
        } else {
            double A[25], Mi[25], r[5];
#pragma unroll
            for (int k = 0; k < 25; ++k) A[k] = D.klin[k];
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const double dd = clampd(D.klin[6 * a], 1e-6, 1e32);
                D.pcg_Dc[6 * (size_t)D.nc + a] = dd;
                A[6 * a] += lambda * dd;
                r[a] = D.klin[25 + a];
            }
            if (!spd_inverse<5>(A, Mi)) atomicOr(D.flags, 2);
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < 5; ++b) v += Mi[5 * a + b] * r[b];
                g += r[a] * v;
                D.pcg_rc[6 * (size_t)D.nc + a] = r[a];
                D.pcg_zc[6 * (size_t)D.nc + a] = v;
                D.pcg_pc[6 * (size_t)D.nc + a] = v;
                D.delta_c[6 * (size_t)D.nc + a] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < 25; ++k) D.pcg_MK[k] = Mi[k];
        }
    }
    for (int i = threadIdx.x; i < D.n_lm_blocks; i += blockDim.x) g += D.pcg_p3[i];  // the landmarks' partial gammas
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) g += __shfl_down(g, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 1024 / 64; ++w) tot += red[w];
        D.pcg_s[0] = tot;                                   // gamma
        D.pcg_s[1] = fmax(1e-10, 1e-10 * 1e-10 * tot);      // max(epsilon_abs, epsilon_rel^2 gamma0)
        D.pcg_s[2] = D.pcg_s[3] = 0.0;
        D.pcg_s[4] = 0.0;                                   // done
        D.pcg_s[5] = 0.0;                                   // iterations
    }
}

// 1/5: p_l <- z_l + beta p_l (not in the first iteration); q_l = A_ll p_l + ElK^T p_K + sum_o E_o^T p_c(o); partials of p_l.q_l and ElK p_l
__global__ __launch_bounds__(TPB) void pcg_apply_landmarks(BaDev D, double lambda) {
    __shared__ double sm[(TPB / 64) * 6];
    if (D.pcg_s[4] != 0.0) return;
    const int j = blockIdx.x * TPB + threadIdx.x;
    const double beta = D.pcg_s[3];
    const bool first = D.pcg_s[5] == 0.0;
    double part[6] = {0, 0, 0, 0, 0, 0};
    if (j < D.nl) {
        const int o0 = D.lm_ptr[j], o1 = D.lm_ptr[j + 1];
        double pl[3], q[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double pv = D.pcg_pl[3 * (size_t)j + a];
            pl[a] = first ? pv : D.pcg_zl[3 * (size_t)j + a] + beta * pv;
            D.pcg_pl[3 * (size_t)j + a] = pl[a];
        }
        if (o1 > o0) {
            const double* in = D.lmlin + (size_t)LMLIN * j;
            const double* pK = D.pcg_pc + 6 * (size_t)D.nc;
            const double* dd = D.pcg_Dl + 3 * (size_t)j;
            q[0] = (in[0] + lambda * dd[0]) * pl[0] + in[1] * pl[1] + in[2] * pl[2];
            q[1] = in[1] * pl[0] + (in[3] + lambda * dd[1]) * pl[1] + in[4] * pl[2];
            q[2] = in[2] * pl[0] + in[4] * pl[1] + (in[5] + lambda * dd[2]) * pl[2];
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const double e0 = in[9 + 3 * a], e1 = in[10 + 3 * a], e2 = in[11 + 3 * a];
                q[0] += e0 * pK[a], q[1] += e1 * pK[a], q[2] += e2 * pK[a];
                part[1 + a] = e0 * pl[0] + e1 * pl[1] + e2 * pl[2];  // (ElK p_l)[a]
            }
            for (int o = o0; o < o1; ++o) {
                const double* E = D.E + 18 * (size_t)D.obs_pos[o];
                const double* pc = D.pcg_pc + 6 * (size_t)D.obs_cam[o];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    q[0] += E[3 * a] * pc[a];
                    q[1] += E[3 * a + 1] * pc[a];
                    q[2] += E[3 * a + 2] * pc[a];
                }
            }
            part[0] = pl[0] * q[0] + pl[1] * q[1] + pl[2] * q[2];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) D.pcg_ql[3 * (size_t)j + a] = q[a];
    }
    block_sum<6>(part, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) D.pcg_p1[6 * (size_t)blockIdx.x + k] = part[k];
}

// 2/5: per camera-aligned chunk: sum_o E_o p_l(o) (6 values)
__global__ __launch_bounds__(TPB) void pcg_apply_cam_chunks(BaDev D) {
    __shared__ double sm[(TPB / 64) * 6];
    if (D.pcg_s[4] != 0.0) return;
    const int2 ch = D.cam_chunks[blockIdx.x];
    double v[6] = {0, 0, 0, 0, 0, 0};
    if ((int)threadIdx.x < ch.y) {
        const int p = ch.x + threadIdx.x;
        const double* E = D.E + 18 * (size_t)p;
        const double* pl = D.pcg_pl + 3 * (size_t)D.cam_lm[p];
        const double p0 = pl[0], p1 = pl[1], p2 = pl[2];
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] = E[3 * a] * p0 + E[3 * a + 1] * p1 + E[3 * a + 2] * p2;
    }
    block_sum<6>(v, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) D.pcg_p2[6 * (size_t)blockIdx.x + k] = v[k];
}

// 3/5 (one block): q_c, q_K, p.q and alpha = gamma / p.q
__global__ __launch_bounds__(1024) void pcg_apply_finish(BaDev D, double lambda) {
    __shared__ double red[1024 / 64][6];
    if (D.pcg_s[4] != 0.0) return;
    const double* pK = D.pcg_pc + 6 * (size_t)D.nc;
    double acc[6] = {0, 0, 0, 0, 0, 0};  // p.q | HcK^T p_c + ElK p_l partials (5)
    for (int c = threadIdx.x; c < D.nc; c += blockDim.x) {
        const double* cl = D.camlin + (size_t)CAMLIN * c;
        const double* pc = D.pcg_pc + 6 * (size_t)c;
        double q[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double v = lambda * D.pcg_Dc[6 * (size_t)c + a] * pc[a];
#pragma unroll
            for (int b = 0; b < 6; ++b) v += cl[6 * a + b] * pc[b];
#pragma unroll
            for (int b = 0; b < 5; ++b) v += cl[36 + 5 * a + b] * pK[b];
            q[a] = v;
        }
        for (int chn = D.cam_chunk_ptr[c]; chn < D.cam_chunk_ptr[c + 1]; ++chn)
#pragma unroll
            for (int a = 0; a < 6; ++a) q[a] += D.pcg_p2[6 * (size_t)chn + a];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            D.pcg_qc[6 * (size_t)c + a] = q[a];
            acc[0] += pc[a] * q[a];
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[1 + b] += cl[36 + 5 * a + b] * pc[a];
        }
    }
    for (int i = threadIdx.x; i < D.n_lm_blocks; i += blockDim.x)
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += D.pcg_p1[6 * (size_t)i + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double x = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 1024 / 64; ++w)
            for (int k = 0; k < 6; ++k) tot[k] += red[w][k];
        double pq = tot[0];
        for (int a = 0; a < 5; ++a) {
            double v = lambda * D.pcg_Dc[6 * (size_t)D.nc + a] * pK[a] + tot[1 + a];
            for (int b = 0; b < 5; ++b) v += D.klin[5 * a + b] * pK[b];
            D.pcg_qc[6 * (size_t)D.nc + a] = v;
            pq += pK[a] * v;
        }
        D.pcg_s[6] = pq;
        D.pcg_s[2] = D.pcg_s[0] / pq;  // alpha
    }
}

// 4/5: x += alpha p, r -= alpha q, z = M^-1 r, partial r.z; threads [0, nl) landmarks, then cameras + K
__global__ __launch_bounds__(TPB) void pcg_update(BaDev D) {
    __shared__ double sm[(TPB / 64) * 1];
    if (D.pcg_s[4] != 0.0) return;
    const double alpha = D.pcg_s[2];
    const int idx = blockIdx.x * TPB + threadIdx.x;
    double g[1] = {0.0};
    if (idx < D.nl) {
        const size_t b = 3 * (size_t)idx;
        double r[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            D.delta_l[b + a] += alpha * D.pcg_pl[b + a];
            r[a] = D.pcg_rl[b + a] - alpha * D.pcg_ql[b + a];
            D.pcg_rl[b + a] = r[a];
        }
        const double* Mi = D.pcg_Ml + 9 * (size_t)idx;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double z = Mi[3 * a] * r[0] + Mi[3 * a + 1] * r[1] + Mi[3 * a + 2] * r[2];
            D.pcg_zl[b + a] = z;
            g[0] += r[a] * z;
        }
    } else if (idx - D.nl <= D.nc) {
        const int c = idx - D.nl, m = c < D.nc ? 6 : 5;
        const size_t b = 6 * (size_t)c;
        const double* Mi = c < D.nc ? D.pcg_Mc + 36 * (size_t)c : D.pcg_MK;
        double r[6] = {0, 0, 0, 0, 0, 0};
        for (int a = 0; a < m; ++a) {
            D.delta_c[b + a] += alpha * D.pcg_pc[b + a];
            r[a] = D.pcg_rc[b + a] - alpha * D.pcg_qc[b + a];
            D.pcg_rc[b + a] = r[a];
        }
        for (int a = 0; a < m; ++a) {
            double z = 0.0;
            for (int k = 0; k < m; ++k) z += Mi[m * a + k] * r[k];
            D.pcg_zc[b + a] = z;
            g[0] += r[a] * z;
        }
    }
    block_sum<1>(g, sm);
    if (threadIdx.x == 0) D.pcg_p3[blockIdx.x] = g[0];
}

// 5/5 (one block): gamma', beta, the stop test, p_c <- z_c + beta p_c (p_l follows in 1/5 of the next iteration)
__global__ __launch_bounds__(1024) void pcg_update_finish(BaDev D, int n_update_blocks) {
    __shared__ double red[1024 / 64];
    __shared__ double s_beta;
    if (D.pcg_s[4] != 0.0) return;
    double g = 0.0;
    for (int i = threadIdx.x; i < n_update_blocks; i += blockDim.x) g += D.pcg_p3[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) g += __shfl_down(g, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 1024 / 64; ++w) tot += red[w];
        const double beta = tot / D.pcg_s[0];
        const double it = D.pcg_s[5] + 1.0;
        D.pcg_s[0] = tot;
        D.pcg_s[3] = beta;
        D.pcg_s[5] = it;
        // for (k = 1; k <= maxIterations && (gamma > threshold || k <= minIterations); ++k), minIterations = 1
        if (!(tot > D.pcg_s[1]) || it >= (double)PCG_MAX_IT || !(tot == tot)) D.pcg_s[4] = 1.0;
        s_beta = beta;
    }
    __syncthreads();
    const double beta = s_beta;
    for (int i = threadIdx.x; i < D.n; i += blockDim.x) D.pcg_pc[i] = D.pcg_zc[i] + beta * D.pcg_pc[i];
}

// tentative points + the landmarks' linearised-cost terms from the PCG step (the role K-F has for the direct solve)
__global__ __launch_bounds__(TPB) void pcg_landmark_step(BaDev D, double lambda) {
    __shared__ double sm[(TPB / 64) * 1];
    const int j = blockIdx.x * TPB + threadIdx.x;
    double lin[1] = {0.0};
    if (j < D.nl) {
        const bool used = D.lm_ptr[j + 1] > D.lm_ptr[j];
        double d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            d[a] = used ? D.delta_l[3 * (size_t)j + a] : 0.0;
            D.delta_l[3 * (size_t)j + a] = d[a];
            D.pt_new[3 * (size_t)j + a] = D.pt[3 * (size_t)j + a] + d[a];
        }
        if (used) {
            const double* in = D.lmlin + (size_t)LMLIN * j;
            lin[0] = 0.5 * (d[0] * in[6] + d[1] * in[7] + d[2] * in[8]) +
                     0.5 * lambda * (clampd(in[0], 1e-6, 1e32) * d[0] * d[0] + clampd(in[3], 1e-6, 1e32) * d[1] * d[1] +
                                     clampd(in[5], 1e-6, 1e32) * d[2] * d[2]);
        }
    }
    block_sum<1>(lin, sm);
    if (threadIdx.x == 0) D.lin_part[blockIdx.x] = lin[0];
}

// ---- DogLeg (GTSAM DoglegOptimizerImpl): scalar forms of the steepest-descent direction g = A^T b and the
// Gauss-Newton step n on the UNDAMPED linearised system H = A^T A (blocks of K-A/K-B), all sums in
// fixed order. forms = {g.g, g.n, n.n, g^T H g, g^T H n, n^T H n}; landmark part here, thread = landmark.
__global__ __launch_bounds__(TPB) void ba_dl_forms_landmarks(BaDev D) {
    __shared__ double sm[(TPB / 64) * 6];
    const int j = blockIdx.x * TPB + threadIdx.x;
    double f[6] = {0, 0, 0, 0, 0, 0};
    if (j < D.nl) {
        const int o0 = D.lm_ptr[j], o1 = D.lm_ptr[j + 1];
        if (o1 > o0) {
            const double* in = D.lmlin + (size_t)LMLIN * j;
            const double gl[3] = {in[6], in[7], in[8]};
            const double nl[3] = {D.dl_nl[3 * (size_t)j], D.dl_nl[3 * (size_t)j + 1], D.dl_nl[3 * (size_t)j + 2]};
            const double* gK = D.klin + 25;
            const double* nK = D.dl_nc + 6 * D.nc;
            // a_x = ElK^T x_K + sum_o E_o^T x_c(o): the part of H x that lands on this landmark from cameras and K
            double ag[3] = {0, 0, 0}, an[3] = {0, 0, 0};
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    ag[c] += in[9 + 3 * a + c] * gK[a];
                    an[c] += in[9 + 3 * a + c] * nK[a];
                }
            for (int o = o0; o < o1; ++o) {
                const double* E = D.E + 18 * (size_t)D.obs_pos[o];
                const int cam = (int)D.obs_cam[o];
                const double* gc = D.camlin + (size_t)CAMLIN * cam + 66;
                const double* nc = D.dl_nc + 6 * (size_t)cam;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        ag[c] += E[3 * a + c] * gc[a];
                        an[c] += E[3 * a + c] * nc[a];
                    }
            }
            const double H00 = in[0], H01 = in[1], H02 = in[2], H11 = in[3], H12 = in[4], H22 = in[5];
            const double hg[3] = {H00 * gl[0] + H01 * gl[1] + H02 * gl[2], H01 * gl[0] + H11 * gl[1] + H12 * gl[2],
                                  H02 * gl[0] + H12 * gl[1] + H22 * gl[2]};
            const double hn[3] = {H00 * nl[0] + H01 * nl[1] + H02 * nl[2], H01 * nl[0] + H11 * nl[1] + H12 * nl[2],
                                  H02 * nl[0] + H12 * nl[1] + H22 * nl[2]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f[0] += gl[c] * gl[c];
                f[1] += gl[c] * nl[c];
                f[2] += nl[c] * nl[c];
                f[3] += gl[c] * hg[c] + 2.0 * ag[c] * gl[c];
                f[4] += gl[c] * hn[c] + ag[c] * nl[c] + an[c] * gl[c];
                f[5] += nl[c] * hn[c] + 2.0 * an[c] * nl[c];
            }
        }
    }
    block_sum<6>(f, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) D.dl_part[6 * (size_t)blockIdx.x + k] = f[k];
}

// camera + K part and the final fixed-order sums -> scal[4..9]
__global__ __launch_bounds__(TPB) void ba_dl_forms_final(BaDev D) {
    __shared__ double sm[(TPB / 64) * 6];
    double f[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < D.n_lm_blocks; i += TPB)
        for (int k = 0; k < 6; ++k) f[k] += D.dl_part[6 * (size_t)i + k];
    const double* gK = D.klin + 25;
    const double* nK = D.dl_nc + 6 * D.nc;
    for (int c = threadIdx.x; c < D.nc; c += TPB) {
        const double* cl = D.camlin + (size_t)CAMLIN * c;
        const double* gc = cl + 66;
        const double* nc = D.dl_nc + 6 * (size_t)c;
        double hg[6], hn[6];  // Hcc x_c + HcK x_K
        for (int a = 0; a < 6; ++a) {
            double sg = 0.0, sn = 0.0;
            for (int b = 0; b < 6; ++b) {
                sg += cl[6 * a + b] * gc[b];
                sn += cl[6 * a + b] * nc[b];
            }
            double kg = 0.0, kn = 0.0;
            for (int b = 0; b < 5; ++b) {
                kg += cl[36 + 5 * a + b] * gK[b];
                kn += cl[36 + 5 * a + b] * nK[b];
            }
            hg[a] = sg + 2.0 * kg;  // x_c^T Hcc y_c + x_c^T HcK y_K + y_c^T HcK x_K, folded per form below
            hn[a] = sn + 2.0 * kn;
            f[0] += gc[a] * gc[a];
            f[1] += gc[a] * nc[a];
            f[2] += nc[a] * nc[a];
            f[3] += gc[a] * hg[a];
            f[4] += gc[a] * (sn + kn) + nc[a] * kg;
            f[5] += nc[a] * hn[a];
        }
    }
    if (threadIdx.x == 0) {
        for (int a = 0; a < 5; ++a) {
            double sg = 0.0, sn = 0.0;
            for (int b = 0; b < 5; ++b) {
                sg += D.klin[5 * a + b] * gK[b];
                sn += D.klin[5 * a + b] * nK[b];
            }
            f[0] += gK[a] * gK[a];
            f[1] += gK[a] * nK[a];
            f[2] += nK[a] * nK[a];
            f[3] += gK[a] * sg;
            f[4] += gK[a] * sn;
            f[5] += nK[a] * sn;
        }
    }
    block_sum<6>(f, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) D.scal[4 + k] = f[k];
}

// x_d = cu * g + cn * n -> delta_c / delta_l (the inputs of the retraction and error kernels) and pt_new
__global__ __launch_bounds__(TPB) void ba_dl_apply(BaDev D, double cu, double cn) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < 3 * D.nl) {
        const int j = i / 3;
        const double g = D.lm_ptr[j + 1] > D.lm_ptr[j] ? D.lmlin[(size_t)LMLIN * j + 6 + i % 3] : 0.0;
        const double d = cu * g + cn * D.dl_nl[i];
        D.delta_l[i] = d;
        D.pt_new[i] = D.pt[i] + d;
    }
    if (i < D.n) {
        const double g = i < 6 * D.nc ? D.camlin[(size_t)CAMLIN * (i / 6) + 66 + i % 6] : D.klin[25 + (i - 6 * D.nc)];
        D.delta_c[i] = cu * g + cn * D.dl_nc[i];
    }
}

}  // namespace eacham

// ====================================================================================================
// host side
// ====================================================================================================
using namespace eacham;

struct eacham_ba_handle {
    BaDev D;
    int block = -1;            // index into ctx->ba_pool: the arena all device arrays of this problem live in
    int block2 = -1;           // device-built problems: a second arena for what is sized by the pair lists and the plan
    char* arena = nullptr;
    size_t arena_off = 0;      // bump pointer (planning pass: the total)
    bool planning = false;     // first pass over the allocation sequence: sizes only
    size_t upload_end = 0;     // end of the last uploaded array in the arena (the uploads come first in the sequence)
    std::vector<char> stage;   // host image of uploaded arrays [stage_base, stage_base + size) of the arena, sent with ONE copy
    size_t stage_base = 0;
    std::vector<int> lm_order;  // landmark-sorted observation index -> caller's observation index
    eacham::BaGroups groups;    // host-built problems: the landmark-major structure of the Schur stage (device-built: sizes only)
    double *kpart = nullptr, *err_cam = nullptr, *lin_cam = nullptr;
    bool finish_pending = false;  // the linearisation's second stage has not run yet: the next try's first launch carries it
    double* scal_host = nullptr;  // pinned: the per-try scalar read-back sits on the LM loop's critical path
    long long ticket = 0;         // number of the last ba_final_sums launch (it stores it behind the scalars)
    double *pose_init = nullptr, *pt_init = nullptr, *K_init = nullptr;
    int n_landmarks_used = 0;
    size_t bytes_linearize = 0, bytes_try = 0;
    // the sparse solve (ba_plan.hpp): the plan stays on the host for the launch sequence and the diagnostic read-back
    eacham::BaPlan plan;
    double prep_us[3] = {0, 0, 0};  // eacham_ba_prepare: host structures | the plan (ordering + symbolic) | arena + upload + sync
    const int* sp_leaves = nullptr;
    const eacham::BaPlanItem* sp_items = nullptr;
    const eacham::BaPlanSrc* sp_srcs = nullptr;
    const int *bs_order = nullptr, *bs_ptr = nullptr, *sp_col_dest = nullptr;
    const int2* bs_ent = nullptr;
};

namespace eacham {

template <class T>
static int dev_alloc(eacham_ctx* ctx, eacham_ba_handle* h, T** p, size_t count) {
    (void)ctx;
    const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
    *p = h->planning ? nullptr : (T*)(h->arena + h->arena_off);
    h->arena_off += bytes;
    return EACHAM_OK;
}
template <class T>
static int dev_upload(eacham_ctx* ctx, eacham_ba_handle* h, const T** p, const std::vector<T>& v) {
    T* q = nullptr;
    int rc = dev_alloc(ctx, h, &q, v.size());
    if (rc) return rc;
    if (h->planning) {
        h->upload_end = h->arena_off;
    } else if (!v.empty()) {
        const size_t off = (size_t)((char*)q - h->arena);
        if (!h->stage.empty() && off >= h->stage_base && off + v.size() * sizeof(T) <= h->stage_base + h->stage.size())
            memcpy(h->stage.data() + (off - h->stage_base), v.data(), v.size() * sizeof(T));
        else EACHAM_HIP_TRY(ctx, hipMemcpyAsync(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    }
    *p = q;
    return EACHAM_OK;
}

static void pose_from_Twc(const double* T, double* x) {  // rigid inverse: camera->world
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) x[3 * i + j] = T[4 * j + i];
    for (int i = 0; i < 3; ++i) x[9 + i] = -(x[3 * i] * T[3] + x[3 * i + 1] * T[7] + x[3 * i + 2] * T[11]);
}
static void pose_to_Twc(const double* x, double* T) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = x[3 * j + i];
        T[4 * i + 3] = -(x[i] * x[9] + x[3 + i] * x[10] + x[6 + i] * x[11]);
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}

// An arena of at least `bytes` from the context's pool: the smallest free one that fits, else a new one (free arenas
// are dropped first once the pool holds eight: a long sequence of growing windows must not keep every size).
static int ba_block_acquire(eacham_ctx* ctx, size_t bytes, int* index) {
    int best = -1, n_free = 0;
    for (int i = 0; i < (int)ctx->ba_pool.size(); ++i) {
        const BaBlock& b = ctx->ba_pool[i];
        if (b.busy || !b.dev) continue;
        ++n_free;
        if (b.bytes >= bytes && (best < 0 || b.bytes < ctx->ba_pool[best].bytes)) best = i;
    }
    if (best < 0) {
        if ((int)ctx->ba_pool.size() >= 8 && n_free > 0) {
            for (BaBlock& b : ctx->ba_pool)
                if (!b.busy && b.dev) {
                    (void)hipFree(b.dev);
                    b.dev = nullptr;
                    b.bytes = 0;
                }
        }
        for (int i = 0; i < (int)ctx->ba_pool.size() && best < 0; ++i)
            if (!ctx->ba_pool[i].busy && !ctx->ba_pool[i].dev) best = i;  // an emptied slot (its pinned block is kept)
        if (best < 0) {
            ctx->ba_pool.emplace_back();
            best = (int)ctx->ba_pool.size() - 1;
        }
        BaBlock& b = ctx->ba_pool[best];
        const size_t want = bytes + bytes / 8;  // some headroom: the next window is rarely the same size
        EACHAM_HIP_TRY(ctx, hipMalloc(&b.dev, want));
        b.bytes = want;
        if (!b.pinned) {
            hipError_t e = hipHostMalloc((void**)&b.pinned, SCAL * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
            if (e != hipSuccess) {
                (void)hipFree(b.dev);
                b.dev = nullptr;
                b.bytes = 0;
                return ctx->fail(EACHAM_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(e));
            }
        }
    }
    ctx->ba_pool[best].busy = true;
    *index = best;
    return EACHAM_OK;
}

// The allocation sequence of a problem's device arrays, in three parts (each runs twice: sizes first, then pointers):
// the plan's tables (uploads), the work arrays sized by (nc, nl, no), and those sized by the pair lists and the plan.
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
static int ba_upload_plan(eacham_ctx* ctx, eacham_ba_handle* h, const BaPlan& plan, const std::vector<int2>& bs_ent) {
    BaDev& D = h->D;
    TRY(dev_upload(ctx, h, &D.sp_pos, plan.pos));
    TRY(dev_upload(ctx, h, &D.sp_tile_map, plan.tile_map));
    TRY(dev_upload(ctx, h, &D.sp_pad_cols, plan.pad_cols));
    TRY(dev_upload(ctx, h, &D.sp_diag_tile, plan.diag_tile));
    TRY(dev_upload(ctx, h, &h->sp_leaves, plan.leaves));
    TRY(dev_upload(ctx, h, &h->sp_items, plan.items));
    TRY(dev_upload(ctx, h, &h->sp_srcs, plan.srcs));
    TRY(dev_upload(ctx, h, &h->bs_order, plan.bs_order));
    TRY(dev_upload(ctx, h, &h->bs_ptr, plan.bs_ptr));
    TRY(dev_upload(ctx, h, &h->bs_ent, bs_ent));
    TRY(dev_upload(ctx, h, &h->sp_col_dest, plan.col_dest));
    return EACHAM_OK;
}
template <class T, class U>
static int dev_upload_as(eacham_ctx* ctx, eacham_ba_handle* h, const T** p, const std::vector<U>& v) {  // same-layout PODs (GrpI2 = int2, ...)
    static_assert(sizeof(T) == sizeof(U), "layout");
    return dev_upload(ctx, h, p, reinterpret_cast<const std::vector<T>&>(v));
}
static int ba_upload_groups(eacham_ctx* ctx, eacham_ba_handle* h, const BaGroups& GR) {
    BaDev& D = h->D;
    D.g_rows = GR.rows; D.g_ngroups = (int)GR.groups.size(); D.g_nblk = GR.n_blk; D.g_nparts = GR.n_parts;
    D.g_nchunks = GR.n_chunks; D.g_nlong = (int)GR.longblk.size(); D.g_nent4 = GR.n_ent4;
    if (GR.rows == 0) return EACHAM_OK;
    TRY(dev_upload(ctx, h, &D.g_groups, GR.groups));
    TRY(dev_upload(ctx, h, &D.g_lmid, GR.lmid));
    TRY(dev_upload(ctx, h, &D.g_lmrow, GR.lmrow));
    TRY(dev_upload_as(ctx, h, &D.g_rowinfo, GR.rowinfo));
    TRY(dev_upload(ctx, h, &D.g_uv, GR.uv));
    TRY(dev_upload(ctx, h, &D.g_chunks, GR.chunks));
    TRY(dev_upload(ctx, h, &D.g_ent, GR.ent));
    TRY(dev_upload(ctx, h, &D.g_laneinfo, GR.laneinfo));
    TRY(dev_upload_as(ctx, h, &D.g_blk, GR.blk));
    TRY(dev_upload(ctx, h, &D.g_longblk, GR.longblk));
    return EACHAM_OK;
}
static int ba_upload_window(eacham_ctx* ctx, eacham_ba_handle* h, const BaWin& WN) {
    BaDev& D = h->D;
    D.w_rows = WN.rows; D.w_ngroups = (int)WN.groups.size(); D.w_stride = WN.rows ? win_stride(D.nc) : 0;
    if (WN.rows == 0) return EACHAM_OK;
    TRY(dev_upload_as(ctx, h, &D.w_groups, WN.groups));
    TRY(dev_upload_as(ctx, h, &D.w_rowinfo, WN.rowinfo));
    TRY(dev_upload(ctx, h, &D.w_uv, WN.uv));
    TRY(dev_upload(ctx, h, &D.w_lmid, WN.lmid));
    TRY(dev_upload(ctx, h, &D.w_lmrow, WN.lmrow));
    return EACHAM_OK;
}
static int ba_alloc_work_a(eacham_ctx* ctx, eacham_ba_handle* h) {
    BaDev& D = h->D;
    const int nc = D.nc, nl = D.nl, no = D.no;
    TRY(dev_alloc(ctx, h, &D.pose, 12 * (size_t)nc));
    TRY(dev_alloc(ctx, h, &D.pose_new, 12 * (size_t)nc));
    TRY(dev_alloc(ctx, h, &D.pt, 3 * (size_t)nl));
    TRY(dev_alloc(ctx, h, &D.pt_new, 3 * (size_t)nl));
    TRY(dev_alloc(ctx, h, &D.Kc, 8));
    TRY(dev_alloc(ctx, h, &D.K_new, 8));
    TRY(dev_alloc(ctx, h, &D.E, 18 * (size_t)no));
    TRY(dev_alloc(ctx, h, &D.Et, 18 * (size_t)no));
    TRY(dev_alloc(ctx, h, &D.lmlin, (size_t)LMLIN * nl));
    TRY(dev_alloc(ctx, h, &D.lmtry, (size_t)LMLIN * nl));
    TRY(dev_alloc(ctx, h, &D.camlin, (size_t)CAMLIN * nc));
    TRY(dev_alloc(ctx, h, &D.klin, (size_t)KLIN));
    TRY(dev_alloc(ctx, h, &h->kpart, (size_t)CLP * LSEG * std::max(nc, 1)));
    TRY(dev_alloc(ctx, h, &D.kk_part, (size_t)30 * D.n_lm_blocks));
    TRY(dev_alloc(ctx, h, &D.delta_c, (size_t)D.n + 8));
    TRY(dev_alloc(ctx, h, &D.delta_l, 3 * (size_t)nl));
    TRY(dev_alloc(ctx, h, &D.err_part, (size_t)std::max(D.n_ll_blocks, D.n_step_blocks)));
    TRY(dev_alloc(ctx, h, &D.lin_part, (size_t)std::max(std::max(D.n_ll_blocks, D.n_step_blocks), D.n_lm_blocks)));
    TRY(dev_alloc(ctx, h, &h->err_cam, (size_t)nc + 1));
    TRY(dev_alloc(ctx, h, &h->lin_cam, (size_t)nc + 1));
    TRY(dev_alloc(ctx, h, &D.scal, (size_t)SCAL));
    TRY(dev_alloc(ctx, h, &D.dl_nc, (size_t)D.n));
    TRY(dev_alloc(ctx, h, &D.dl_nl, (size_t)3 * D.nl));
    TRY(dev_alloc(ctx, h, &D.dl_part, (size_t)6 * D.n_lm_blocks));
    {
        const size_t nn = (size_t)D.n + 8, n3 = 3 * (size_t)std::max(nl, 1);
        TRY(dev_alloc(ctx, h, &D.pcg_rc, nn)); TRY(dev_alloc(ctx, h, &D.pcg_zc, nn)); TRY(dev_alloc(ctx, h, &D.pcg_pc, nn));
        TRY(dev_alloc(ctx, h, &D.pcg_qc, nn)); TRY(dev_alloc(ctx, h, &D.pcg_Dc, nn));
        TRY(dev_alloc(ctx, h, &D.pcg_rl, n3)); TRY(dev_alloc(ctx, h, &D.pcg_zl, n3)); TRY(dev_alloc(ctx, h, &D.pcg_pl, n3));
        TRY(dev_alloc(ctx, h, &D.pcg_ql, n3)); TRY(dev_alloc(ctx, h, &D.pcg_Dl, n3));
        TRY(dev_alloc(ctx, h, &D.pcg_Mc, (size_t)36 * std::max(nc, 1))); TRY(dev_alloc(ctx, h, &D.pcg_MK, 32));
        TRY(dev_alloc(ctx, h, &D.pcg_Ml, 3 * n3));
        TRY(dev_alloc(ctx, h, &D.pcg_p1, (size_t)6 * D.n_lm_blocks));
        TRY(dev_alloc(ctx, h, &D.pcg_p3, (size_t)D.n_lm_blocks + (size_t)(nc + 1 + TPB) / TPB + 2)); TRY(dev_alloc(ctx, h, &D.pcg_s, 16));
    }
    TRY(dev_alloc(ctx, h, &D.sync_counter, 4));
    if (!h->planning) {
        EACHAM_HIP_TRY(ctx, hipMemsetAsync(D.sync_counter, 0, 4 * sizeof(int), ctx->stream));
        h->scal_host = ctx->ba_pool[h->block].pinned;
        EACHAM_HIP_TRY(ctx, hipHostGetDevicePointer((void**)&D.scal_pinned, h->scal_host, 0));
        for (int k = 0; k < SCAL; ++k) h->scal_host[k] = 0.0;  // (tickets start at 1; the block may have served another problem)
    }
    return EACHAM_OK;
}
static int ba_alloc_work_b(eacham_ctx* ctx, eacham_ba_handle* h) {
    BaDev& D = h->D;
    const BaPlan& plan = h->plan;
    TRY(dev_alloc(ctx, h, &D.T, (size_t)plan.ntiles * TILE));
    TRY(dev_alloc(ctx, h, &D.Winv, (size_t)2 * plan.npan * NB * NB));   // W_a, W_b of every panel, row-major
    TRY(dev_alloc(ctx, h, &D.Xrow, (size_t)plan.npan * NB * NB));       // X = -W_b L_ba W_a, row-major
    TRY(dev_alloc(ctx, h, &D.Wops, (size_t)plan.npan * TILE_OPS));      // W_a, W_b, X of every panel in MFMA operand order
    TRY(dev_alloc(ctx, h, &D.zsol, (size_t)plan.npan * PB));
    TRY(dev_alloc(ctx, h, &D.partial, (size_t)36 * std::max(D.g_rows > 0 ? D.g_nparts : D.n_chunks, 1)));
    TRY(dev_alloc(ctx, h, &D.w_part, (size_t)D.w_stride * std::max(D.w_ngroups, 1)));
    if (!h->planning && D.w_rows > 0) {
        EACHAM_HIP_TRY(ctx, hipMemsetAsync(D.lmtry, 0, sizeof(double) * LMLIN * (size_t)D.nl, ctx->stream));  // (landmarks without observations)
        if (!ctx->ba_dense_lds_set) {
            EACHAM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)ba_schur_dense, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->ba_dense_lds_set = true;
        }
    }
    if (!h->planning && D.g_rows > 0) {
        // landmarks without observations are never visited by ba_schur_groups and keep a zero record
        EACHAM_HIP_TRY(ctx, hipMemsetAsync(D.lmtry, 0, sizeof(double) * LMLIN * (size_t)D.nl, ctx->stream));
        if (!ctx->ba_groups_lds_set) {  // (the kernel asks for more than the default dynamic LDS limit)
            EACHAM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)ba_schur_groups<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_groups_lds_bytes(TPB)));
            EACHAM_HIP_TRY(ctx, hipFuncSetAttribute((const void*)ba_schur_groups<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_groups_lds_bytes(2 * TPB)));
            ctx->ba_groups_lds_set = true;
        }
    }
    TRY(dev_alloc(ctx, h, &D.bpart, (size_t)36 * std::max(D.n_cam_chunks, 1)));
    TRY(dev_alloc(ctx, h, &D.pcg_p2, (size_t)6 * std::max(D.n_cam_chunks, 1)));
    TRY(dev_alloc(ctx, h, &D.flags, (size_t)N_STATUS + plan.npan));  // [0..3] status, [4 + P] hand-off flag of panel P
    if (!h->planning) EACHAM_HIP_TRY(ctx, hipMemsetAsync(D.flags, 0, (N_STATUS + plan.npan) * sizeof(int), ctx->stream));  // (final_sums leaves them cleared)
    return EACHAM_OK;
}
#undef TRY

// The structure built by host loops (the round-1..3 form): what a local window of a few thousand observations still uses —
// a dozen dependent launches and three read-backs cost more than these loops on a problem that small — and the reference
// the device-built structure is held against bit for bit (EACHAM_BA_PREPARE=host|device forces either form).
static int ba_prepare_host(eacham_ctx* ctx, const eacham_ba_problem* P, eacham_ba_handle** out, bool allow_dense) {
    if (!P || P->n_cams < 0 || P->n_points < 0 || P->n_obs < 0) return ctx->fail(EACHAM_ERR_INVALID, "bad BA problem");
    const int nc = P->n_cams, nl = P->n_points, no = P->n_obs;
    if (no > 0 && (!P->obs_cam || !P->obs_point || !P->obs_uv)) return ctx->fail(EACHAM_ERR_INVALID, "null observation arrays");
    if ((nc > 0 && (!P->cam_T_wc || !P->cam_fixed)) || (nl > 0 && (!P->points || !P->point_observers)))
        return ctx->fail(EACHAM_ERR_INVALID, "null camera/point arrays");
    for (int o = 0; o < no; ++o)
        if (P->obs_cam[o] >= (uint32_t)nc || P->obs_point[o] >= (uint32_t)nl)
            return ctx->fail(EACHAM_ERR_INVALID, "observation %d references a camera/point out of range", o);
    eacham_ba_handle* h = new eacham_ba_handle();
    const auto t_begin = std::chrono::steady_clock::now();
    auto us_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    BaDev& D = h->D;
    memset(&D, 0, sizeof(D));
    D.nc = nc; D.nl = nl; D.no = no; D.n = 6 * nc + 5;
    D.nz = make_noise();
    D.n_lm_blocks = std::max(1, (nl + TPB - 1) / TPB);
    D.lpl = ctx->ba_lpl_lin > 0 ? ctx->ba_lpl_lin : (nl <= 8192 ? 8 : 2);  // (S200: 45.7 / 40.4 / 41.0 / 45.6 us for 1 / 2 / 4 / 8, tools/ba_lpl_lin.sh)
    D.n_ll_blocks = std::max(1, (int)(((long long)nl * D.lpl + TPB - 1) / TPB));
    D.lpl_step = ctx->ba_lpl_step > 0 ? ctx->ba_lpl_step : (nl <= 8192 ? 8 : 2);
    D.n_step_blocks = std::max(1, (int)(((long long)nl * D.lpl_step + TPB - 1) / TPB));

    // ---- structure: observations grouped by landmark (stable), then by camera ----
    std::vector<int> lm_ptr(nl + 1, 0);
    for (int o = 0; o < no; ++o) lm_ptr[P->obs_point[o] + 1]++;
    for (int j = 0; j < nl; ++j) {
        if (lm_ptr[j + 1] > 0) h->n_landmarks_used++;
        lm_ptr[j + 1] += lm_ptr[j];
    }
    std::vector<int> fill(lm_ptr.begin(), lm_ptr.end() - 1);
    h->lm_order.resize(no);
    for (int o = 0; o < no; ++o) h->lm_order[fill[P->obs_point[o]]++] = o;
    std::vector<unsigned> obs_cam(no), obs_lm(no);
    std::vector<double> obs_uv(2 * (size_t)no);
    for (int p = 0; p < no; ++p) {
        const int o = h->lm_order[p];
        obs_cam[p] = P->obs_cam[o];
        obs_lm[p] = P->obs_point[o];
        obs_uv[2 * (size_t)p] = P->obs_uv[2 * (size_t)o];
        obs_uv[2 * (size_t)p + 1] = P->obs_uv[2 * (size_t)o + 1];
    }
    std::vector<int> cam_ptr(nc + 1, 0), cam_obs(no);
    for (int p = 0; p < no; ++p) cam_ptr[obs_cam[p] + 1]++;
    for (int c = 0; c < nc; ++c) cam_ptr[c + 1] += cam_ptr[c];
    {
        std::vector<int> f2(cam_ptr.begin(), cam_ptr.end() - 1);
        for (int p = 0; p < no; ++p) cam_obs[f2[obs_cam[p]]++] = p;
    }
    std::vector<int> obs_pos(no), pos_cam(no);
    for (int p = 0; p < no; ++p) obs_pos[cam_obs[p]] = p, pos_cam[p] = (int)obs_cam[cam_obs[p]];
    std::vector<int2> cam_chunks;
    std::vector<int> cam_chunk_ptr(nc + 1, 0);
    for (int c = 0; c < nc; ++c) {
        for (int q = cam_ptr[c]; q < cam_ptr[c + 1]; q += TPB) cam_chunks.push_back(make_int2(q, std::min(TPB, cam_ptr[c + 1] - q)));
        cam_chunk_ptr[c + 1] = (int)cam_chunks.size();
    }
    D.n_cam_chunks = (int)cam_chunks.size();
    std::vector<int> cam_lm(no);
    std::vector<double> cam_uv(2 * (size_t)no);
    for (int p = 0; p < no; ++p) {
        cam_lm[p] = (int)obs_lm[cam_obs[p]];
        cam_uv[2 * (size_t)p] = obs_uv[2 * (size_t)cam_obs[p]];
        cam_uv[2 * (size_t)p + 1] = obs_uv[2 * (size_t)cam_obs[p] + 1];
    }
    // ---- the landmark-major structure of the Schur stage (ba_groups.hpp); the pair lists below only when it does not apply ----
    // ---- the dense form for a local window (ba_window.hpp): its structure is these rows and nothing else ----
    BaWin WN;
    const bool use_dense = allow_dense && build_window(nc, nl, lm_ptr.data(), obs_cam.data(), obs_uv.data(), WN, ctx->ba_window_rows);
    BaGroups& GR = h->groups;
    const bool use_groups = !use_dense && ctx->ba_schur_mode == 1 && build_groups(nc, nl, lm_ptr.data(), obs_cam.data(), obs_uv.data(), GR, ctx->ba_group_rows);
    if (!use_groups) GR = BaGroups();
    // ---- camera-pair lists of the Schur complement: block (c <= c') -> (o, o') pairs, landmark order ----
    // (two passes over every observation pair of every landmark — count, then fill: this loop is most of the host time of
    // preparing a local window, hence the flat 32-bit index arithmetic; entries are written as Et positions directly)
    const bool use_pairs = !use_groups && !use_dense;
    const long long nblk_all = use_pairs ? (long long)nc * (nc + 1) / 2 : 0;
    if (nblk_all > 0x7fffffffLL) {
        delete h;
        return ctx->fail(EACHAM_ERR_UNSUPPORTED, "too many cameras (%d) for the camera-block index", nc);
    }
    std::vector<int> rowoff(std::max(nc, 1));  // block (c, c2 >= c) has index rowoff[c] + c2
    for (int c = 0; c < nc; ++c) rowoff[c] = (int)((long long)c * nc - (long long)c * (c - 1) / 2 - c);
    auto bid = [&](int c, int c2) { return rowoff[c] + c2; };
    std::vector<int> bcount((size_t)nblk_all + 1, 0);
    if (use_pairs) {
        const unsigned* oc = obs_cam.data();
        int* bc = bcount.data() + 1;
        for (int j = 0; j < nl; ++j) {
            const int a0 = lm_ptr[j], a1 = lm_ptr[j + 1];
            for (int a = a0; a < a1; ++a) {
                const int ca = (int)oc[a];
                bc[rowoff[ca] + ca] += 1;  // (a, a)
                for (int b = a + 1; b < a1; ++b) {
                    const int cb = (int)oc[b];
                    if (ca < cb) bc[rowoff[ca] + cb] += 1;
                    else if (ca > cb) bc[rowoff[cb] + ca] += 1;
                    else bc[rowoff[ca] + ca] += 2;
                }
            }
        }
    }
    std::vector<long long> bstart((size_t)nblk_all + 1, 0);
    for (long long b = 0; b < nblk_all; ++b) bstart[b + 1] = bstart[b] + bcount[b + 1];
    const long long n_entries = bstart[nblk_all];
    if (n_entries > 0x7fffffffLL) {
        delete h;
        return ctx->fail(EACHAM_ERR_UNSUPPORTED, "Schur pair list too large (%lld entries)", n_entries);
    }
    std::vector<int2> entries((size_t)n_entries);
    if (use_pairs) {
        std::vector<int> pos((size_t)nblk_all);
        for (long long b = 0; b < nblk_all; ++b) pos[b] = (int)bstart[b];
        const unsigned* oc = obs_cam.data();
        const int* op = obs_pos.data();  // Et records live in camera order
        int2* en = entries.data();
        int* ps = pos.data();
        for (int j = 0; j < nl; ++j) {
            const int a0 = lm_ptr[j], a1 = lm_ptr[j + 1];
            for (int a = a0; a < a1; ++a) {
                const int ca = (int)oc[a], pa = op[a];
                en[ps[rowoff[ca] + ca]++] = make_int2(pa, pa);
                for (int b = a + 1; b < a1; ++b) {
                    const int cb = (int)oc[b], pb = op[b];
                    if (ca < cb) en[ps[rowoff[ca] + cb]++] = make_int2(pa, pb);
                    else if (ca > cb) en[ps[rowoff[cb] + ca]++] = make_int2(pb, pa);
                    else {
                        int& q = ps[rowoff[ca] + ca];
                        en[q++] = make_int2(pa, pb);
                        en[q++] = make_int2(pb, pa);
                    }
                }
            }
        }
    }
    std::vector<int4> chunks, blocks;
    for (int c = 0; c < (use_pairs ? nc : 0); ++c)
        for (int c2 = c; c2 < nc; ++c2) {
            const long long b = bid(c, c2);
            const long long cnt = bstart[b + 1] - bstart[b];
            if (cnt == 0 && c != c2) continue;  // absent off-diagonal block stays zero
            const int first_chunk = (int)chunks.size();
            int k = 0;
            for (long long s = 0; s < cnt; s += PAIR_CHUNK, ++k)
                chunks.push_back(make_int4((int)blocks.size(), (int)(bstart[b] + s), (int)std::min<long long>(PAIR_CHUNK, cnt - s), k));
            blocks.push_back(make_int4(c, c2, first_chunk, k));
        }
    D.n_chunks = (int)chunks.size();
    D.n_blocks = (int)blocks.size();
    // ---- the sparse solve: ordering, panels, symbolic factor, level schedule (ba_plan.hpp) ----
    const auto t_plan = std::chrono::steady_clock::now();
    h->prep_us[0] = us_since(t_begin);
    {
        std::vector<std::pair<int, int>> cam_edges;
        cam_edges.reserve(blocks.size() + GR.blk.size());
        for (const int4& b : blocks)
            if (b.x != b.y) cam_edges.emplace_back(b.x, b.y);
        for (const GrpI4& b : GR.blk)
            if (b.x != b.y && b.y < nc) cam_edges.emplace_back(b.x, b.y);
        if (use_dense)  // every block is kept: the cameras of a window all share landmarks with the current frame
            for (int c = 0; c < nc; ++c)
                for (int c2 = c + 1; c2 < nc; ++c2) cam_edges.emplace_back(c, c2);
        int hint = P->ordering;
        if (hint == EACHAM_BA_ORDER_AUTO && ctx->ba_ordering != EACHAM_BA_ORDER_AUTO) hint = ctx->ba_ordering;
        if (hint < EACHAM_BA_ORDER_AUTO || hint > EACHAM_BA_ORDER_ND) {
            delete h;
            return ctx->fail(EACHAM_ERR_INVALID, "unknown BA ordering %d", hint);
        }
        try {
            build_ba_plan(nc, cam_edges, hint, h->plan);
        } catch (...) {  // (no exception crosses the C-ABI)
            delete h;
            return ctx->fail(EACHAM_ERR_HIP, "BA preparation: the analysis of the reduced system ran out of memory or threads");
        }
    }
    h->prep_us[1] = us_since(t_plan);
    const auto t_upload = std::chrono::steady_clock::now();
    const BaPlan& plan = h->plan;
    D.sp_npan = plan.npan; D.sp_ntiles = plan.ntiles; D.sp_posK = plan.posK; D.sp_rhs_row = plan.rhs_row;
    D.sp_n_pad = (int)plan.pad_cols.size();
    std::vector<int2> bs_ent(plan.bs_ent.size());
    for (size_t e = 0; e < bs_ent.size(); ++e) bs_ent[e] = make_int2(plan.bs_ent[e].first, plan.bs_ent[e].second);

    // ---- values ----
    std::vector<double> pose(12 * (size_t)nc), lmprior(2 * (size_t)nl), K5(5);
    for (int c = 0; c < nc; ++c) pose_from_Twc(P->cam_T_wc + 16 * (size_t)c, &pose[12 * (size_t)c]);
    for (int j = 0; j < nl; ++j) {  // BundleAdjuster.cpp:109-113: sigma = 1.0f/obs, k = 3.0f/obs (float)
        const float o = (float)(P->point_observers[j] > 0 ? P->point_observers[j] : 1);
        lmprior[2 * (size_t)j] = (double)(1.0f / o);
        lmprior[2 * (size_t)j + 1] = (double)(3.0f / o);
    }
    K5[0] = P->K[0]; K5[1] = P->K[1]; K5[2] = 0.0; K5[3] = P->K[2]; K5[4] = P->K[3];
    std::vector<double> pts(P->points, P->points + 3 * (size_t)nl);
    std::vector<int> fixed(P->cam_fixed, P->cam_fixed + nc);

    // Every device array of the problem is carved out of one arena: the allocation sequence below runs twice, first
    // to add up the sizes, then — with an arena of that size taken from the context's pool — to hand out the
    // pointers and issue the uploads. An arena that served another problem holds its bytes: nothing here may rely on
    // fresh memory being zero (S is cleared by every tryLambda, Lm and the flags below).
    int rc = EACHAM_OK;
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
    auto layout = [&]() -> int {
        const double *c_pose0, *c_pt0, *c_K0, *c_lmprior, *c_uv;
        TRY(dev_upload(ctx, h, &c_pose0, pose));
        TRY(dev_upload(ctx, h, &c_pt0, pts));
        TRY(dev_upload(ctx, h, &c_K0, K5));
        TRY(dev_upload(ctx, h, &c_lmprior, lmprior));
        TRY(dev_upload(ctx, h, &c_uv, obs_uv));
        D.pose0 = (double*)c_pose0; D.pt0 = (double*)c_pt0; D.K0 = (double*)c_K0; D.lmprior = c_lmprior; D.obs_uv = c_uv;
        h->pose_init = D.pose0; h->pt_init = D.pt0; h->K_init = D.K0;
        TRY(dev_upload(ctx, h, &D.fixed, fixed));
        TRY(dev_upload(ctx, h, &D.lm_ptr, lm_ptr));
        TRY(dev_upload(ctx, h, &D.cam_ptr, cam_ptr));
        TRY(dev_upload(ctx, h, &D.cam_obs, cam_obs));
        TRY(dev_upload(ctx, h, &D.cam_lm, cam_lm));
        TRY(dev_upload(ctx, h, &D.cam_chunks, cam_chunks));
        TRY(dev_upload(ctx, h, &D.cam_chunk_ptr, cam_chunk_ptr));
        TRY(dev_upload(ctx, h, &D.obs_pos, obs_pos));
        TRY(dev_upload(ctx, h, &D.pos_cam, pos_cam));
        TRY(dev_upload(ctx, h, &D.cam_uv, cam_uv));
        TRY(dev_upload(ctx, h, &D.obs_cam, obs_cam));
        TRY(dev_upload(ctx, h, &D.obs_lm, obs_lm));
        TRY(dev_upload(ctx, h, &D.pair_entries, entries));
        TRY(dev_upload(ctx, h, &D.pair_chunks, chunks));
        TRY(dev_upload(ctx, h, &D.blocks, blocks));
        TRY(ba_upload_groups(ctx, h, GR));
        TRY(ba_upload_window(ctx, h, WN));
        TRY(ba_upload_plan(ctx, h, plan, bs_ent));
        TRY(ba_alloc_work_a(ctx, h));
        TRY(ba_alloc_work_b(ctx, h));
        return EACHAM_OK;
    };
#undef TRY
    h->planning = true;
    h->arena_off = 0;
    rc = layout();
    if (!rc) rc = ba_block_acquire(ctx, h->arena_off, &h->block);
    if (!rc) {
        h->arena = (char*)ctx->ba_pool[h->block].dev;
        h->planning = false;
        h->arena_off = 0;
        // a local window uploads ~20 arrays of a few KB: one copy of their image instead of 20 (each is a staged,
        // synchronous-looking call from pageable memory); large problems keep the per-array copies (no second host copy)
        if (h->upload_end <= ((size_t)4 << 20)) h->stage.assign(h->upload_end, 0);
        rc = layout();
        if (!rc && !h->stage.empty())
            rc = hipMemcpyAsync(h->arena, h->stage.data(), h->stage.size(), hipMemcpyHostToDevice, ctx->stream) == hipSuccess
                     ? EACHAM_OK : ctx->fail(EACHAM_ERR_HIP, "BA upload failed");
    }
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);  // (uploads from the host vectors may be in flight)
        if (h->block >= 0) ctx->ba_pool[h->block].busy = false;
        delete h;
        return rc;
    }
    // The per-array copies of a large problem read host vectors that die here: wait for them. A small problem was sent as
    // ONE image out of h->stage, which lives as long as the handle: nothing to wait for (the first kernel of the solve
    // queues behind the copy on the same stream) — 30-40 us of every local-window call.
    hipError_t e = h->stage.empty() ? hipStreamSynchronize(ctx->stream) : hipSuccess;
    if (e != hipSuccess) {
        ctx->ba_pool[h->block].busy = false;
        delete h;
        return ctx->fail(EACHAM_ERR_HIP, "BA upload failed: %s", hipGetErrorString(e));
    }
    h->prep_us[2] = us_since(t_upload);
    // algorithmic HBM bytes; used by the benchmark's roofline line
    h->bytes_linearize = (size_t)no * (24 + 144) + (size_t)no * 24 + (size_t)nl * (24 + LMLIN * 8) + (size_t)nc * (96 + CAMLIN * 8);
    h->bytes_try = (size_t)no * (144 * 2 + 144 + 144) + (size_t)n_entries * 8 + (size_t)nl * (LMLIN * 8 * 3 + 48) +
                   (size_t)no * 24 + (size_t)D.n * D.n * 8;
    *out = h;
    return EACHAM_OK;
}

// ======================================================================================================================
// Device-side construction of the problem structure (round 4). RefineBA's graph build is part of the reference's call
// (modules/sfm/reconstruction/BundleAdjuster.cpp:57-178); as host loops it took 11.6 ms on S200 and 30 ms on config 4 in
// front of 1.8 / 3.4 ms of Levenberg-Marquardt. Here the caller's arrays are uploaded as they are and everything else is
// built by sorts and scans (devprim.hpp): observations grouped by landmark and by camera (stable radix sorts: the order of
// a sequential host loop), the camera-aligned chunk table, the observation pairs of every landmark expanded in (landmark,
// a, b) order and sorted by camera block (stable: a block's entries stay in landmark order, which is the order every
// fp64 sum of the Schur complement runs in), the block and chunk tables by a scan over the block histogram. The host
// keeps what is irregular and small: the elimination ordering + symbolic analysis of the camera graph (ba_plan.hpp), fed by
// the block table read back. Three read-backs of a few bytes / kilobytes; the result is bit-identical with ba_prepare_host.
// ======================================================================================================================

struct PrepCounters {   // device-side scalars of the construction, read back by the host
    long long n_entries;     // expanded observation pairs
    int n_used;              // landmarks with at least one observation
    int bad;                 // an observation names a camera / landmark out of range
    int n_cam_chunks;
    int pad;
    prim::I3 totals;         // {entries, blocks, chunks} after the block scan
};

__device__ __forceinline__ int block_row_start(int c, int nc) {  // index of block (c, c) in the row-major upper triangle
    return (int)((long long)c * nc - (long long)c * (c - 1) / 2);
}

__global__ __launch_bounds__(TPB) void prep_values(int nc, int nl, const double* __restrict__ T_wc, const int* __restrict__ observers,
                                                   double* __restrict__ pose0, double* __restrict__ lmprior, double* __restrict__ K0,
                                                   double fx, double fy, double cx, double cy) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i == 0) {
        K0[0] = fx; K0[1] = fy; K0[2] = 0.0; K0[3] = cx; K0[4] = cy;
    }
    if (i < nc) {  // pose_from_Twc, operation for operation (no contraction: the host form has none)
        const double* T = T_wc + 16 * (size_t)i;
        double* x = pose0 + 12 * (size_t)i;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) x[3 * a + b] = T[4 * b + a];
        {
#pragma clang fp contract(off)  // (the header-defined __dmul_rn / __dadd_rn are plain operators and would be fused)
            for (int a = 0; a < 3; ++a) x[9 + a] = -((T[a] * T[3] + T[4 + a] * T[7]) + T[8 + a] * T[11]);
        }
    } else if (i - nc < nl) {  // BundleAdjuster.cpp:109-113: sigma = 1.0f/obs, k = 3.0f/obs (float)
        const int j = i - nc;
        const float o = (float)(observers[j] > 0 ? observers[j] : 1);
        lmprior[2 * (size_t)j] = (double)(1.0f / o);
        lmprior[2 * (size_t)j + 1] = (double)(3.0f / o);
    }
}

__global__ __launch_bounds__(TPB) void prep_keys_lm(int no, int nc, int nl, const uint32_t* __restrict__ in_cam,
                                                    const uint32_t* __restrict__ in_pt, uint32_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals, PrepCounters* __restrict__ cnt) {
    const int o = blockIdx.x * TPB + threadIdx.x;
    if (o >= no) return;
    uint32_t pt = in_pt[o];
    if (pt >= (uint32_t)nl || in_cam[o] >= (uint32_t)nc) {
        cnt->bad = 1;  // (benign race: every writer stores 1)
        pt = 0;
    }
    keys[o] = pt;
    vals[o] = (uint32_t)o;
}

// fills ptr[k] = first position whose key is >= k for the keys (prev, cur] seen at a boundary; sorted keys
__device__ __forceinline__ void fill_ptr(int* __restrict__ ptr, int pos, int prev_key, int cur_key) {
    for (int k = prev_key + 1; k <= cur_key; ++k) ptr[k] = pos;
}

constexpr int GATHER_CHUNKS = 4;   // chunks of TPB positions per workgroup of prep_gather_lm
__global__ __launch_bounds__(TPB) void prep_gather_lm(int no, int nl, int nc, const uint32_t* __restrict__ lm_sorted,
                                                      const uint32_t* __restrict__ order, const uint32_t* __restrict__ in_cam,
                                                      unsigned* __restrict__ obs_cam,
                                                      unsigned* __restrict__ obs_lm, int* __restrict__ lm_ptr,
                                                      uint32_t* __restrict__ cam_keys, uint32_t* __restrict__ cam_vals,
                                                      PrepCounters* __restrict__ cnt) {
    // A workgroup takes GATHER_CHUNKS consecutive chunks of TPB positions and adds its count of first positions to n_used ONCE: as one
    // atomic per wave (7 800 of them on S200, all to one address: ~12 ns each at the L2) the counter alone was 94 us of this kernel.
    __shared__ int s_first[TPB / 64];
    int mine = 0;   // first positions seen by this wave (lane 0 carries the count)
#pragma unroll 1
    for (int i = 0; i < GATHER_CHUNKS; ++i) {
        const int p = (blockIdx.x * GATHER_CHUNKS + i) * TPB + threadIdx.x;
        bool first = false;
        if (p < no) {
            const uint32_t o = order[p];
            const int lm = (int)lm_sorted[p];
            uint32_t c = in_cam[o];
            if (c >= (uint32_t)nc) c = 0;  // (flagged by prep_keys_lm: the call fails, the kernels stay in range)
            obs_cam[p] = c;
            obs_lm[p] = (unsigned)lm;
            cam_keys[p] = c;
            cam_vals[p] = (uint32_t)p;
            const int prev = p > 0 ? (int)lm_sorted[p - 1] : -1;
            first = lm != prev;
            if (first) fill_ptr(lm_ptr, p, prev, lm);
            if (p == no - 1) fill_ptr(lm_ptr, no, lm, nl);
        }
        mine += __popcll(__ballot(first));
    }
    if ((threadIdx.x & 63) == 0) s_first[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < TPB / 64; ++w) total += s_first[w];
        if (total) atomicAdd(&cnt->n_used, total);
    }
}

__global__ __launch_bounds__(TPB) void prep_gather_cam(int no, int nc, const uint32_t* __restrict__ cam_sorted,
                                                       const uint32_t* __restrict__ cam_obs_u, const unsigned* __restrict__ obs_lm,
                                                       int* __restrict__ cam_obs,
                                                       int* __restrict__ obs_pos, int* __restrict__ pos_cam, int* __restrict__ cam_lm,
                                                       int* __restrict__ cam_ptr) {
    const int q = blockIdx.x * TPB + threadIdx.x;
    if (q >= no) return;
    const int p = (int)cam_obs_u[q], c = (int)cam_sorted[q];
    cam_obs[q] = p;
    obs_pos[p] = q;
    pos_cam[q] = c;
    cam_lm[q] = (int)obs_lm[p];
    const int prev = q > 0 ? (int)cam_sorted[q - 1] : -1;
    if (c != prev) fill_ptr(cam_ptr, q, prev, c);
    if (q == no - 1) fill_ptr(cam_ptr, no, c, nc);
}
// The measurements in landmark order and in camera order, straight from the caller's array. A kernel of its own, LATE in the first
// half: the 16 bytes per observation are the largest of the caller's arrays (8 MB on S200, ~0.25 ms of PCIe from pageable memory)
// and nothing before the group rows needs them — the upload runs on the second stream while the sorts do (it sat between the
// landmark sort and its gather before, with the device idle for its whole length).
__global__ __launch_bounds__(TPB) void prep_gather_uv(int no, const uint32_t* __restrict__ order, const int* __restrict__ cam_obs,
                                                      const double* __restrict__ in_uv, double* __restrict__ obs_uv, double* __restrict__ cam_uv) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= no) return;
    const double2 a = *reinterpret_cast<const double2*>(&in_uv[2 * (size_t)order[i]]);
    const double2 b = *reinterpret_cast<const double2*>(&in_uv[2 * (size_t)order[cam_obs[i]]]);
    *reinterpret_cast<double2*>(&obs_uv[2 * (size_t)i]) = a;
    *reinterpret_cast<double2*>(&cam_uv[2 * (size_t)i]) = b;
}

__global__ __launch_bounds__(TPB) void prep_cam_chunk_counts(int nc, const int* __restrict__ cam_ptr, int* __restrict__ nchunk) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c < nc) nchunk[c] = (cam_ptr[c + 1] - cam_ptr[c] + TPB - 1) / TPB;
}
__global__ __launch_bounds__(TPB) void prep_cam_chunk_fill(int nc, const int* __restrict__ cam_ptr, const int* __restrict__ cam_chunk_ptr,
                                                           int2* __restrict__ cam_chunks) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c >= nc) return;
    int k = cam_chunk_ptr[c];
    for (int q = cam_ptr[c]; q < cam_ptr[c + 1]; q += TPB) cam_chunks[k++] = make_int2(q, min(TPB, cam_ptr[c + 1] - q));
}

// pair-list entries that start at observation a (landmark order): (a, a) and one per later observation b of the landmark,
// two when a and b sit in the same camera
__global__ __launch_bounds__(TPB) void prep_pair_counts(int no, const unsigned* __restrict__ obs_cam, const unsigned* __restrict__ obs_lm,
                                                        const int* __restrict__ lm_ptr, long long* __restrict__ cnt) {
    const int a = blockIdx.x * TPB + threadIdx.x;
    if (a >= no) return;
    const int a1 = lm_ptr[obs_lm[a] + 1];
    const unsigned ca = obs_cam[a];
    int c = 1;
    for (int b = a + 1; b < a1; ++b) c += 1 + (obs_cam[b] == ca ? 1 : 0);
    cnt[a] = c;
}

// camera graph of the reduced system: adj[c][c'] = 1 iff the two cameras share a landmark (benign races: every writer
// stores 1). The host's ordering + symbolic analysis starts from it while the device is still building the pair lists.
__global__ __launch_bounds__(TPB) void prep_cam_adjacency(int no, int nc, const unsigned* __restrict__ obs_cam, const unsigned* __restrict__ obs_lm,
                                                          const int* __restrict__ lm_ptr, unsigned char* __restrict__ adj) {
    const int a = blockIdx.x * TPB + threadIdx.x;
    if (a >= no) return;
    const int a1 = lm_ptr[obs_lm[a] + 1];
    const int ca = (int)obs_cam[a];
    for (int b = a + 1; b < a1; ++b) {
        const int cb = (int)obs_cam[b];
        if (ca < cb) adj[(size_t)ca * nc + cb] = 1;
        else if (cb < ca) adj[(size_t)cb * nc + ca] = 1;
    }
}

// the entries in (landmark, a, b) order = the order of the host loops; key = camera block, value = the two Et positions
__global__ __launch_bounds__(TPB) void prep_expand(int no, int nc, const unsigned* __restrict__ obs_cam, const unsigned* __restrict__ obs_lm,
                                                   const int* __restrict__ lm_ptr, const int* __restrict__ obs_pos,
                                                   const long long* __restrict__ off, uint32_t* __restrict__ keys,
                                                   int2* __restrict__ vals) {
    const int a = blockIdx.x * TPB + threadIdx.x;
    if (a >= no) return;
    const int a1 = lm_ptr[obs_lm[a] + 1];
    const int ca = (int)obs_cam[a], pa = obs_pos[a];
    long long e = off[a];
    const int diag = block_row_start(ca, nc);  // block (ca, ca); block (c, c2 >= c) = row_start(c) + c2 - c
    keys[e] = (uint32_t)diag;
    vals[e++] = make_int2(pa, pa);
    for (int b = a + 1; b < a1; ++b) {
        const int cb = (int)obs_cam[b], pb = obs_pos[b];
        if (ca < cb) {
            const int k = diag + (cb - ca);
            keys[e] = (uint32_t)k;
            vals[e++] = make_int2(pa, pb);
        } else if (ca > cb) {
            const int k = block_row_start(cb, nc) + (ca - cb);
            keys[e] = (uint32_t)k;
            vals[e++] = make_int2(pb, pa);
        } else {
            keys[e] = (uint32_t)diag;
            vals[e++] = make_int2(pa, pb);
            keys[e] = (uint32_t)diag;
            vals[e++] = make_int2(pb, pa);
        }
    }
}

// entries per camera block from the SORTED keys: the first and the one-past-last position of every run (2.75 M atomic increments
// on 20 k counters, most of them on the diagonal blocks, were 0.28 of the 0.75 ms the preparation's kernels took on S200)
__global__ __launch_bounds__(TPB) void prep_block_runs(int n, const uint32_t* __restrict__ keys, int* __restrict__ first, int* __restrict__ last) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) first[k] = i;
    if (i == n - 1 || keys[i + 1] != k) last[k] = i + 1;
}

// per block of the upper triangle: {entries, present, chunks}; absent off-diagonal blocks stay out of the tables
__global__ __launch_bounds__(TPB) void prep_block_counts(int nc, int nblk, const int* __restrict__ first, const int* __restrict__ last,
                                                         int* __restrict__ bcount, prim::I3* __restrict__ t) {
    const int b = blockIdx.x * TPB + threadIdx.x;
    if (b >= nblk) return;
    int lo = 0, hi = nc - 1;  // largest c with row_start(c) <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block_row_start(mid, nc) <= b) lo = mid;
        else hi = mid - 1;
    }
    const int cnt = last[b] - first[b];   // (both zero for a block without entries)
    bcount[b] = cnt;
    const bool present = cnt > 0 || b == block_row_start(lo, nc);
    t[b] = prim::I3{cnt, present ? 1 : 0, present ? (cnt + PAIR_CHUNK - 1) / PAIR_CHUNK : 0};
}
__global__ __launch_bounds__(TPB) void prep_block_fill(int nc, int nblk, const int* __restrict__ bcount, const prim::I3* __restrict__ s,
                                                       int4* __restrict__ blocks, int4* __restrict__ chunks) {
    const int b = blockIdx.x * TPB + threadIdx.x;
    if (b >= nblk) return;
    int lo = 0, hi = nc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block_row_start(mid, nc) <= b) lo = mid;
        else hi = mid - 1;
    }
    const int c = lo, c2 = c + (b - block_row_start(c, nc));
    const int cnt = bcount[b];
    if (cnt == 0 && c != c2) return;
    const prim::I3 at = s[b];  // {first entry, block index, first chunk}
    const int k = (cnt + PAIR_CHUNK - 1) / PAIR_CHUNK;
    blocks[at.b] = make_int4(c, c2, at.c, k);
    for (int i = 0; i < k; ++i) chunks[at.c + i] = make_int4(at.b, at.a + PAIR_CHUNK * i, min(PAIR_CHUNK, cnt - PAIR_CHUNK * i), i);
}

struct Bump {  // carves 256-byte aligned arrays out of a scratch buffer (base == nullptr: sizes only)
    char* base;
    size_t off = 0;
    explicit Bump(void* b) : base((char*)b) {}
    template <class T>
    T* take(size_t n) {
        T* p = base ? (T*)(base + off) : nullptr;
        off += (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

// ---- device form of the landmark-major structure (ba_groups.hpp holds the definition; these kernels reproduce build_groups
// bit for bit: tests/test_ba_prepare_gpu.py compares every array) -------------------------------------------------------------
struct GrpCounters {
    int cmax, emax;            // largest landmark cost / entry count (atomicMax: order-free)
    int ng, n_long, cost_total, row_total;
    int any_dup, pad;          // some landmark sees a camera twice: the sort-free entries kernel does not apply
    prim::I3 item_totals;      // {mandatory items, blocks, .} of the sorted item list
    prim::I3 totals;           // {chunks, uint4-rows of entries, segments} over all groups
};

// per landmark: sort key, rows, entries, cost (ba_groups.hpp step 1-2)
__global__ __launch_bounds__(TPB) void prep_grp_keys(int nl, const int* __restrict__ lm_ptr, const unsigned* __restrict__ obs_cam,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int* __restrict__ cost,
                                                     GrpCounters* __restrict__ gc) {
    // (the two maxima leave the workgroup as ONE atomic each: per thread they were 100 000 atomics to two addresses on S200)
    __shared__ int s_cmax, s_emax;
    if (threadIdx.x == 0) s_cmax = 0, s_emax = 0;
    __syncthreads();
    const int j = blockIdx.x * TPB + threadIdx.x;
    int c = 0, ec = 0;
    if (j < nl) {
        const int a0 = lm_ptr[j], a1 = lm_ptr[j + 1], m = a1 - a0;
        vals[j] = (uint32_t)j;
        if (m == 0) {
            keys[j] = 0xffffffffu;
            cost[j] = 0;
        } else {
            unsigned mn = obs_cam[a0], mx = mn;
            long long e = (long long)(m + 1) * (m + 2) / 2;
            for (int a = a0; a < a1; ++a) {
                const unsigned ca = obs_cam[a];
                mn = min(mn, ca), mx = max(mx, ca);
                for (int b = a + 1; b < a1; ++b) e += obs_cam[b] == ca ? 1 : 0;
            }
            if (e != (long long)(m + 1) * (m + 2) / 2) gc->any_dup = 1;  // (benign race: every writer stores 1)
            keys[j] = grp_morton(mn, mx);
            ec = e > 0x3fffffffLL ? 0x3fffffff : (int)e;
            c = max(max(m + 1, 4), (ec + GRP_ENT_PER_ROW - 1) / GRP_ENT_PER_ROW);  // ceil(e rows / ent_max), ent_max = 16 rows
            cost[j] = c;
        }
    }
    if (c > 0) atomicMax(&s_cmax, c);      // (LDS: order-free maxima; counters start at 0 and every real value is positive)
    if (ec > 0) atomicMax(&s_emax, ec);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_cmax > 0) atomicMax(&gc->cmax, s_cmax);
        if (s_emax > 0) atomicMax(&gc->emax, s_emax);
    }
}
// in sorted order: rows and cost of rank k (zero beyond the used landmarks, which the key puts last)
__global__ __launch_bounds__(TPB) void prep_grp_sorted(int nl, const uint32_t* __restrict__ lm_sorted, const int* __restrict__ lm_ptr,
                                                       const int* __restrict__ cost, int* __restrict__ rowsS, int* __restrict__ costS) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= nl) return;
    const int j = (int)lm_sorted[k], m = lm_ptr[j + 1] - lm_ptr[j];
    rowsS[k] = m > 0 ? m + 1 : 0;
    costS[k] = cost[j];
}
// group g starts at the first rank whose cost prefix reaches g R0 (empty groups: the rank behind them); lm0[ng] = n_used
__global__ __launch_bounds__(TPB) void prep_grp_bounds(int nu, const int* __restrict__ cstart, int R0, int* __restrict__ lm0, GrpCounters* __restrict__ gc) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= nu) return;
    const int g = cstart[k] / R0, gp = k > 0 ? cstart[k - 1] / R0 : -1;
    for (int gg = gp + 1; gg <= g; ++gg) lm0[gg] = k;   // (a group no rank starts in is empty: lm0[gg] = lm0[gg + 1])
    if (k == nu - 1) {
        lm0[g + 1] = nu;
        gc->ng = g + 1;
    }
}
// per rank: the padded per-group arrays (landmark ids, own rows, row records, measurements) and the group record's first half
__global__ __launch_bounds__(TPB) void prep_grp_fill(int nu, int nc, int R, int R0, const uint32_t* __restrict__ lm_sorted, const int* __restrict__ cstart,
                                                     const int* __restrict__ rstart, const int* __restrict__ lm0, const int* __restrict__ lm_ptr,
                                                     const unsigned* __restrict__ obs_cam, const double* __restrict__ obs_uv, int* __restrict__ lmid,
                                                     int* __restrict__ lmrow, int2* __restrict__ rowinfo, double* __restrict__ uv) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= nu) return;
    const int g = cstart[k] / R0, k0 = lm0[g], t = k - k0, j = (int)lm_sorted[k];
    const int a0 = lm_ptr[j], m = lm_ptr[j + 1] - a0, r0 = rstart[k] - rstart[k0];
    const int LMAX = R / 4;
    lmid[(size_t)g * LMAX + t] = j;
    lmrow[(size_t)g * LMAX + t] = r0 + m;
    const size_t base = (size_t)g * R + r0;
    for (int i = 0; i < m; ++i) {
        rowinfo[base + i] = make_int2((int)obs_cam[a0 + i], t);
        uv[2 * (base + i)] = obs_uv[2 * (size_t)(a0 + i)];
        uv[2 * (base + i) + 1] = obs_uv[2 * (size_t)(a0 + i) + 1];
    }
    rowinfo[base + m] = make_int2(nc, t);
}

__global__ __launch_bounds__(TPB) void prep_grp_records(const GrpCounters* __restrict__ gc, const int* __restrict__ lm0, const int* __restrict__ rstart,
                                                        BaGroup* __restrict__ groups) {
    const int g = blockIdx.x * TPB + threadIdx.x;
    if (g >= gc->ng) return;
    const int k0 = lm0[g], k1 = lm0[g + 1];  // (rstart has an entry behind the last rank: the total)
    groups[g] = BaGroup{k0, k1 - k0, rstart[k0], rstart[k1] - rstart[k0], 0, 0, 0, 0};
}

// One workgroup per group: ba_groups.hpp step 3-4. The group's entries are generated as 64-bit words
//   block key << 31 | emission index << 18 | r1 << 9 | r2        (rows <= 511, <= 8192 entries)
// and sorted by a bitonic network in LDS (the word order IS (block key, emission index)); runs, slices, lanes, segments and
// chunks follow from scans over the sorted list. PASS 0 only counts ({chunks, uint4-rows, segments} -> counts[g]); PASS 1, given
// the scanned bases, writes the chunk table, the entries, the lane records and the (key, where) list of the segments.
constexpr int GE_THREADS = 256;
constexpr int GE_MAXE = 8192;   // most entries of a group the kernel is ever asked for (rows <= 512)
template <class T>
__device__ __forceinline__ T ge_block_scan(T v, T* lds, T* total) {  // exclusive scan over the workgroup, GE_THREADS values
    const int tid = threadIdx.x;
    lds[tid] = v;
    __syncthreads();
    for (int off = 1; off < GE_THREADS; off <<= 1) {
        const T u = tid >= off ? lds[tid - off] : T(0);
        __syncthreads();
        lds[tid] += u;
        __syncthreads();
    }
    const T incl = lds[tid];
    *total = lds[GE_THREADS - 1];
    __syncthreads();
    return incl - v;
}
template <int PASS>
__global__ __launch_bounds__(GE_THREADS) void prep_grp_entries(int emax /* power of two >= the group bound on entries */, int nc, int R, const GrpCounters* __restrict__ gc, BaGroup* __restrict__ groups,
                                                               const int2* __restrict__ rowinfo, const int* __restrict__ lmrow,
                                                               prim::I3* __restrict__ counts, const prim::I3* __restrict__ bases,
                                                               BaChunk* __restrict__ chunks, uint32_t* __restrict__ ent, uint32_t* __restrict__ laneinfo,
                                                               uint32_t* __restrict__ seg_key, uint32_t* __restrict__ seg_where, int seg_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long ge_lds[];
    const int g = blockIdx.x, tid = threadIdx.x;
    if (g >= gc->ng) {
        if (PASS == 0 && tid == 0) counts[g] = prim::I3{0, 0, 0};
        return;
    }
    const BaGroup G = groups[g];
    const int LMAX = R / 4;
    unsigned long long* words = ge_lds;                         // [emax]
    uint32_t* lane_key = reinterpret_cast<uint32_t*>(ge_lds + emax);                // [emax] (a lane holds at least one entry)
    unsigned short* lane_first = reinterpret_cast<unsigned short*>(lane_key + emax);  // [emax]
    unsigned char* lane_len = reinterpret_cast<unsigned char*>(lane_first + emax);    // [emax]
    int* scan_buf = reinterpret_cast<int*>(lane_len + emax);    // [GE_THREADS]
    int* eoff = scan_buf + GE_THREADS;                          // [LMAX] first entry of landmark t
    // ---- entries per landmark (recomputed from the rows: cheap) and their offsets ----
    int my_e = 0, my_r0 = 0, my_m = 0;
    if (tid < G.nlm) {
        const int own = lmrow[(size_t)g * LMAX + tid];
        const int prev = tid > 0 ? lmrow[(size_t)g * LMAX + tid - 1] + 1 : 0;
        my_r0 = prev, my_m = own - prev;
        my_e = (my_m + 1) * (my_m + 2) / 2;
        const int2* ri = rowinfo + (size_t)g * R + my_r0;
        for (int a = 0; a < my_m; ++a)
            for (int b = a + 1; b < my_m; ++b) my_e += ri[a].x == ri[b].x ? 1 : 0;
    }
    int ne = 0;
    {
        // LMAX may exceed GE_THREADS only for rows > 1024 (not supported): one landmark per thread
        const int off = ge_block_scan<int>(my_e, scan_buf, &ne);
        if (tid < G.nlm) eoff[tid] = off;
    }
    __syncthreads();
    int npow = 64;
    while (npow < ne) npow <<= 1;
    const unsigned long long W = (unsigned long long)nc + 1;
    for (int i = tid; i < npow; i += GE_THREADS) words[i] = ~0ull;
    __syncthreads();
    if (tid < G.nlm) {
        const int2* ri = rowinfo + (size_t)g * R + my_r0;
        unsigned long long idx = (unsigned long long)eoff[tid];
        auto emit = [&](unsigned long long c1, unsigned long long c2, int r1, int r2) {
            words[idx] = ((c1 * W + c2) << 31) | (idx << 18) | ((unsigned long long)r1 << 9) | (unsigned long long)r2;
            ++idx;
        };
        for (int a = 0; a <= my_m; ++a) {
            const int ca = ri[a].x, ra = my_r0 + a;
            emit(ca, ca, ra, ra);
            for (int b = a + 1; b <= my_m; ++b) {
                const int cb = ri[b].x, rb = my_r0 + b;
                if (ca < cb) emit(ca, cb, ra, rb);
                else if (ca > cb) emit(cb, ca, rb, ra);
                else emit(ca, ca, ra, rb), emit(ca, ca, rb, ra);
            }
        }
    }
    __syncthreads();
    // ---- bitonic sort, ascending ----
    for (int kk = 2; kk <= npow; kk <<= 1)
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < npow; i += GE_THREADS) {
                const int ixj = i ^ jj;
                if (ixj > i) {
                    const unsigned long long a = words[i], b = words[ixj];
                    const bool up = (i & kk) == 0;
                    if ((a > b) == up) words[i] = b, words[ixj] = a;
                }
            }
            __syncthreads();
        }
    // ---- runs -> lanes ----
    // per sorted position: run start flag; every thread handles a contiguous strip of positions
    const int per = (npow + GE_THREADS - 1) / GE_THREADS;  // positions per thread (strip)
    const int p0 = tid * per, p1 = min(p0 + per, ne);
    // pass over the strip: number of runs that START in it, by class (long: L > 4 counted in lanes; short: lanes = 1)
    // Run lengths need the next start: a thread finishes a run that started in its strip by walking on (runs are short: <= rows).
    int lanes_long = 0, lanes_short = 0;
    for (int i = p0; i < p1; ++i) {
        const unsigned long long key = words[i] >> 31;
        if (i > 0 && (words[i - 1] >> 31) == key) continue;
        int e = i + 1;
        while (e < ne && (words[e] >> 31) == key) ++e;
        const int L = e - i;
        if (L > 4) lanes_long += (L + GRP_SLICE - 1) / GRP_SLICE;
        else lanes_short += 1;
    }
    int tot_long = 0, tot_short = 0;
    const int off_long = ge_block_scan<int>(lanes_long, scan_buf, &tot_long);
    const int off_short = ge_block_scan<int>(lanes_short, scan_buf, &tot_short);
    const int nlanes = tot_long + tot_short, nchunks = (nlanes + 63) / 64;
    {
        int ll = off_long, ls = tot_long + off_short;
        for (int i = p0; i < p1; ++i) {
            const unsigned long long key = words[i] >> 31;
            if (i > 0 && (words[i - 1] >> 31) == key) continue;
            int e = i + 1;
            while (e < ne && (words[e] >> 31) == key) ++e;
            const int L = e - i;
            if (L > 4) {
                const int n = (L + GRP_SLICE - 1) / GRP_SLICE;
                for (int q = 0, at = i; q < n; ++q) {
                    const int len = L / n + (q < L % n ? 1 : 0);
                    lane_first[ll] = (unsigned short)at, lane_len[ll] = (unsigned char)len, lane_key[ll] = (uint32_t)key;
                    at += len, ++ll;
                }
            } else {
                lane_first[ls] = (unsigned short)i, lane_len[ls] = (unsigned char)L, lane_key[ls] = (uint32_t)key;
                ++ls;
            }
        }
    }
    __syncthreads();
    // ---- chunks: steps per chunk, segments ----
    // thread c < nchunks: longest lane of chunk c (nchunks <= GE_MAXE / 64 = 128 <= GE_THREADS)
    int my_n4 = 0;
    if (tid < nchunks) {
        int longest = 0;
        for (int l = 64 * tid; l < min(nlanes, 64 * tid + 64); ++l) longest = max(longest, (int)lane_len[l]);
        my_n4 = (longest + 3) / 4;
    }
    int tot_n4 = 0;
    const int off_n4 = ge_block_scan<int>(my_n4, scan_buf, &tot_n4);
    __shared__ int s_chunk_ent0[GE_MAXE / 64], s_chunk_n4[GE_MAXE / 64];
    if (tid < nchunks) s_chunk_ent0[tid] = off_n4, s_chunk_n4[tid] = my_n4;
    // segment heads: lane li is a head iff (li - h0) % GRP_SEG == 0, h0 = first lane of its key at or after the chunk start
    int my_heads = 0;
    const int lper = (nlanes + GE_THREADS - 1) / GE_THREADS;
    const int l0 = tid * lper, l1 = min(l0 + lper, nlanes);
    for (int li = l0; li < l1; ++li) {
        const uint32_t key = lane_key[li];
        int h = li;
        const int cs = li & ~(GRP_ROW - 1);   // a segment stays inside its row of 16 lanes
        while (h > cs && lane_key[h - 1] == key) --h;
        if (((li - h) % GRP_SEG) == 0) ++my_heads;
    }
    int tot_heads = 0;
    const int off_heads = ge_block_scan<int>(my_heads, scan_buf, &tot_heads);
    if (PASS == 0) {
        if (tid == 0) counts[g] = prim::I3{nchunks, tot_n4, tot_heads};
        return;
    }
    __syncthreads();
    // ---- PASS 1: write ----
    const prim::I3 base = bases[g];   // {first chunk, first uint4-row, first segment}
    if (tid == 0) {
        groups[g].chunk0 = base.a; groups[g].nchunks = nchunks; groups[g].n_entries = ne; groups[g].n_segments = tot_heads;
    }
    if (tid < nchunks) chunks[base.a + tid] = BaChunk{base.b + s_chunk_ent0[tid], s_chunk_n4[tid]};
    // lanes: records + segment list
    {
        int hs = off_heads;
        for (int li = l0; li < l1; ++li) {
            const uint32_t key = lane_key[li];
            const int cs = li & ~(GRP_ROW - 1), ce = min(nlanes, cs + GRP_ROW);
            int h = li;
            while (h > cs && lane_key[h - 1] == key) --h;
            h += (li - h) / GRP_SEG * GRP_SEG;
            int e = h;
            while (e < ce && e < h + GRP_SEG && lane_key[e] == key) ++e;
            uint32_t info = (uint32_t)(e - 1 - li) << 28;
            const size_t where = (size_t)64 * base.a + li;
            if (li == h) {
                seg_key[seg_off + base.c + hs] = key;
                seg_where[seg_off + base.c + hs] = (uint32_t)where;
                ++hs;
                info |= 1u;  // (prep_grp_slots writes the slot)
            }
            laneinfo[where] = info;
        }
        for (int li = nlanes + tid; li < 64 * nchunks; li += GE_THREADS) laneinfo[(size_t)64 * base.a + li] = 0;
    }
    // entries: [chunk][step][lane][4], null beyond a slice
    const uint32_t null_ent = (uint32_t)R | ((uint32_t)R << 16);
    for (int c = 0; c < nchunks; ++c) {
        const int n4 = s_chunk_n4[c];
        uint32_t* dst = ent + (size_t)(base.b + s_chunk_ent0[c]) * 256;
        for (int x = tid; x < n4 * 256; x += GE_THREADS) {
            const int step = x >> 8, l = (x >> 2) & 63, i = 4 * step + (x & 3), li = 64 * c + l;
            uint32_t v = null_ent;
            if (li < nlanes && i < (int)lane_len[li]) {
                const unsigned long long w = words[lane_first[li] + i];
                v = (uint32_t)((w >> 9) & 0x1ffu) | ((uint32_t)(w & 0x1ffu) << 16);
            }
            dst[x] = v;
        }
    }
}
// The same two passes WITHOUT a sort, for problems in which no landmark sees a camera twice (prep_grp_keys knows; the general
// kernel above serves the others): with the group's cameras numbered locally in ascending order, the run of block (la, lb) is
// the landmarks that see both — M[la] & M[lb] for per-camera bit masks over the group's <= 128 landmarks — in ascending
// landmark order, which IS the (block key, emission index) order of the definition. Pairs are walked in key order by strips,
// lanes / chunks / segments follow from scans as above, and a lane's entries are read off the mask (the bits from its start
// position on; the two rows by a search through the landmark's few rows). 0.56 ms per pass on S200 with the bitonic sort,
// the per-group time here is a small sort of the <= 512 row cameras and a few scans.
__device__ __forceinline__ int gf_pairs_before(int la, int U) { return la * U - la * (la - 1) / 2; }
template <int PASS>
__global__ __launch_bounds__(GE_THREADS) void prep_grp_entries_fast(int lane_cap, int nc, int R, const GrpCounters* __restrict__ gc, BaGroup* __restrict__ groups,
                                                                    const int2* __restrict__ rowinfo, const int* __restrict__ lmrow,
                                                                    prim::I3* __restrict__ counts, const prim::I3* __restrict__ bases,
                                                                    BaChunk* __restrict__ chunks, uint32_t* __restrict__ ent, uint32_t* __restrict__ laneinfo,
                                                                    uint32_t* __restrict__ seg_key, uint32_t* __restrict__ seg_where, int seg_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long gf_lds[];
    const int g = blockIdx.x, tid = threadIdx.x;
    if (g >= gc->ng) {
        if (PASS == 0 && tid == 0) counts[g] = prim::I3{0, 0, 0};
        return;
    }
    const BaGroup G = groups[g];
    const int LMAX = R / 4;
    int RP2 = 64;
    while (RP2 < R) RP2 <<= 1;
    unsigned long long* mask = gf_lds;                                   // [2 R] landmarks that see local camera la (two words)
    int* camsort = reinterpret_cast<int*>(mask + 2 * R);                 // [RP2]
    int* ucam = camsort + RP2;                                           // [R] local camera -> camera
    int* scan_buf = ucam + R;                                            // [GE_THREADS]
    uint32_t* lane_q = reinterpret_cast<uint32_t*>(scan_buf + GE_THREADS);  // [lane_cap] pair of the lane
    unsigned short* row_la = reinterpret_cast<unsigned short*>(lane_q + lane_cap);  // [R] local camera of a row
    unsigned short* lm_r0 = row_la + R;                                  // [LMAX + 1] first row of landmark t
    unsigned char* lane_at = reinterpret_cast<unsigned char*>(lm_r0 + LMAX + 2);    // [lane_cap] first position of the lane inside its run
    unsigned char* lane_len = lane_at + lane_cap;                        // [lane_cap]
    __shared__ int s_chunk_ent0[GE_MAXE / 64], s_chunk_n4[GE_MAXE / 64];
    // ---- the group's cameras, ascending, without repetition ----
    for (int r = tid; r < RP2; r += GE_THREADS) {
        const int c = r < G.nrows ? rowinfo[(size_t)g * R + r].x : 0x7fffffff;
        camsort[r] = c;
    }
    for (int t = tid; t <= G.nlm; t += GE_THREADS) lm_r0[t] = (unsigned short)(t == 0 ? 0 : lmrow[(size_t)g * LMAX + t - 1] + 1);
    for (int i = tid; i < 2 * R; i += GE_THREADS) mask[i] = 0ull;
    __syncthreads();
    for (int kk = 2; kk <= RP2; kk <<= 1)
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < RP2; i += GE_THREADS) {
                const int ixj = i ^ jj;
                if (ixj > i) {
                    const int a = camsort[i], b = camsort[ixj];
                    if ((a > b) == ((i & kk) == 0)) camsort[i] = b, camsort[ixj] = a;
                }
            }
            __syncthreads();
        }
    int U = 0;
    {
        const int per = RP2 / GE_THREADS > 0 ? RP2 / GE_THREADS : 1;   // consecutive elements per thread
        const int i0 = tid * per, i1 = min(i0 + per, RP2);
        int mine = 0;
        for (int i = i0; i < i1; ++i) mine += (camsort[i] != 0x7fffffff && (i == 0 || camsort[i] != camsort[i - 1])) ? 1 : 0;
        int at = ge_block_scan<int>(i0 < RP2 ? mine : 0, scan_buf, &U);
        for (int i = i0; i < i1; ++i)
            if (camsort[i] != 0x7fffffff && (i == 0 || camsort[i] != camsort[i - 1])) ucam[at++] = camsort[i];
    }
    __syncthreads();
    for (int r = tid; r < G.nrows; r += GE_THREADS) {
        const int2 ri = rowinfo[(size_t)g * R + r];
        int lo = 0, hi = U - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ucam[mid] < ri.x) lo = mid + 1;
            else hi = mid;
        }
        row_la[r] = (unsigned short)lo;
        atomicOr(&mask[2 * lo + (ri.y >> 6)], 1ull << (ri.y & 63));
    }
    __syncthreads();
    // ---- pairs (la <= lb) in key order, by strips ----
    const int P = U * (U + 1) / 2;
    const int per = (P + GE_THREADS - 1) / GE_THREADS;
    const int q0 = min(tid * per, P), q1 = min(q0 + per, P);
    auto pair_of = [&](int q, int& la, int& lb) {
        int lo = 0, hi = U - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (gf_pairs_before(mid, U) <= q) lo = mid;
            else hi = mid - 1;
        }
        la = lo, lb = lo + (q - gf_pairs_before(lo, U));
    };
    auto run_len = [&](int la, int lb) {
        return __popcll(mask[2 * la] & mask[2 * lb]) + __popcll(mask[2 * la + 1] & mask[2 * lb + 1]);
    };
    int lanes_long = 0, lanes_short = 0, my_entries = 0;
    if (q0 < q1) {
        int la, lb;
        pair_of(q0, la, lb);
        for (int q = q0; q < q1; ++q) {
            const int L = run_len(la, lb);
            my_entries += L;
            if (L > 4) lanes_long += (L + GRP_SLICE - 1) / GRP_SLICE;
            else if (L > 0) lanes_short += 1;
            if (++lb == U) ++la, lb = la;
        }
    }
    int tot_long = 0, tot_short = 0, ne = 0;
    const int off_long = ge_block_scan<int>(lanes_long, scan_buf, &tot_long);
    const int off_short = ge_block_scan<int>(lanes_short, scan_buf, &tot_short);
    (void)ge_block_scan<int>(my_entries, scan_buf, &ne);
    const int nlanes = tot_long + tot_short, nchunks = (nlanes + 63) / 64;
    if (q0 < q1) {
        int la, lb, ll = off_long, ls = tot_long + off_short;
        pair_of(q0, la, lb);
        for (int q = q0; q < q1; ++q) {
            const int L = run_len(la, lb);
            if (L > 4) {
                const int n = (L + GRP_SLICE - 1) / GRP_SLICE;
                for (int k = 0, at = 0; k < n; ++k) {
                    const int len = L / n + (k < L % n ? 1 : 0);
                    lane_q[ll] = (uint32_t)q, lane_at[ll] = (unsigned char)at, lane_len[ll] = (unsigned char)len;
                    at += len, ++ll;
                }
            } else if (L > 0) {
                lane_q[ls] = (uint32_t)q, lane_at[ls] = 0, lane_len[ls] = (unsigned char)L;
                ++ls;
            }
            if (++lb == U) ++la, lb = la;
        }
    }
    __syncthreads();
    // ---- chunks and segments, as in the general kernel (a lane's block = its pair) ----
    int my_n4 = 0;
    if (tid < nchunks) {
        int longest = 0;
        for (int l = 64 * tid; l < min(nlanes, 64 * tid + 64); ++l) longest = max(longest, (int)lane_len[l]);
        my_n4 = (longest + 3) / 4;
    }
    int tot_n4 = 0;
    const int off_n4 = ge_block_scan<int>(my_n4, scan_buf, &tot_n4);
    if (tid < nchunks) s_chunk_ent0[tid] = off_n4, s_chunk_n4[tid] = my_n4;
    int my_heads = 0;
    const int lper = (nlanes + GE_THREADS - 1) / GE_THREADS;
    const int l0 = min(tid * lper, nlanes), l1 = min(l0 + lper, nlanes);
    for (int li = l0; li < l1; ++li) {
        const uint32_t q = lane_q[li];
        int h = li;
        const int cs = li & ~(GRP_ROW - 1);   // a segment stays inside its row of 16 lanes
        while (h > cs && lane_q[h - 1] == q) --h;
        if (((li - h) % GRP_SEG) == 0) ++my_heads;
    }
    int tot_heads = 0;
    const int off_heads = ge_block_scan<int>(my_heads, scan_buf, &tot_heads);
    if (PASS == 0) {
        if (tid == 0) counts[g] = prim::I3{nchunks, tot_n4, tot_heads};
        return;
    }
    __syncthreads();
    // ---- PASS 1: write ----
    const prim::I3 base = bases[g];   // {first chunk, first uint4-row, first segment}
    if (tid == 0) {
        groups[g].chunk0 = base.a; groups[g].nchunks = nchunks; groups[g].n_entries = ne; groups[g].n_segments = tot_heads;
    }
    if (tid < nchunks) chunks[base.a + tid] = BaChunk{base.b + s_chunk_ent0[tid], s_chunk_n4[tid]};
    const uint32_t W = (uint32_t)nc + 1;
    {
        int hs = off_heads;
        for (int li = l0; li < l1; ++li) {
            const uint32_t q = lane_q[li];
            const int cs = li & ~(GRP_ROW - 1), ce = min(nlanes, cs + GRP_ROW);
            int h = li;
            while (h > cs && lane_q[h - 1] == q) --h;
            h += (li - h) / GRP_SEG * GRP_SEG;
            int e = h;
            while (e < ce && e < h + GRP_SEG && lane_q[e] == q) ++e;
            uint32_t info = (uint32_t)(e - 1 - li) << 28;
            const size_t where = (size_t)64 * base.a + li;
            if (li == h) {
                int la, lb;
                pair_of((int)q, la, lb);
                seg_key[seg_off + base.c + hs] = (uint32_t)ucam[la] * W + (uint32_t)ucam[lb];
                seg_where[seg_off + base.c + hs] = (uint32_t)where;
                ++hs;
                info |= 1u;  // (prep_grp_slots writes the slot)
            }
            laneinfo[where] = info;
        }
        for (int li = nlanes + tid; li < 64 * nchunks; li += GE_THREADS) laneinfo[(size_t)64 * base.a + li] = 0;
    }
    // entries, a lane per thread: the landmarks of the run from position lane_at on; the two rows by a search through the landmark's rows
    const uint32_t null_ent = (uint32_t)R | ((uint32_t)R << 16);
    for (int li = tid; li < 64 * nchunks; li += GE_THREADS) {
        const int c = li >> 6, l = li & 63, n4 = s_chunk_n4[c];
        uint4* dst = reinterpret_cast<uint4*>(ent + (size_t)(base.b + s_chunk_ent0[c]) * 256) + l;
        int len = 0, la = 0, lb = 0;
        unsigned long long m0 = 0, m1 = 0;
        if (li < nlanes) {
            pair_of((int)lane_q[li], la, lb);
            m0 = mask[2 * la] & mask[2 * lb], m1 = mask[2 * la + 1] & mask[2 * lb + 1];
            len = lane_len[li];
            for (int skip = lane_at[li]; skip > 0; --skip) {  // drop the run's first lane_at landmarks
                if (m0) m0 &= m0 - 1;
                else m1 &= m1 - 1;
            }
        }
        for (int step = 0; step < n4; ++step) {
            uint32_t v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = null_ent;
                if (4 * step + i < len) {
                    int t;
                    if (m0) t = __builtin_ctzll(m0), m0 &= m0 - 1;
                    else t = 64 + __builtin_ctzll(m1), m1 &= m1 - 1;
                    int r1 = 0, r2 = 0;
                    for (int r = lm_r0[t]; r < (int)lm_r0[t + 1]; ++r) {
                        const int x = row_la[r];
                        r1 = x == la ? r : r1;
                        r2 = x == lb ? r : r2;
                    }
                    v[i] = (uint32_t)r1 | ((uint32_t)r2 << 16);
                }
            }
            dst[(size_t)step * 64] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
}
__host__ inline size_t grp_entries_fast_lds_bytes(int lane_cap, int R) {
    int RP2 = 64;
    while (RP2 < R) RP2 <<= 1;
    return sizeof(unsigned long long) * 2 * R + sizeof(int) * ((size_t)RP2 + R + GE_THREADS) + sizeof(uint32_t) * (size_t)lane_cap +
           sizeof(unsigned short) * ((size_t)R + R / 4 + 2) + 2 * (size_t)lane_cap + 16;
}
__host__ inline size_t grp_entries_lds_bytes(int emax, int R) {
    return (size_t)emax * (8 + 4 + 2 + 1) + sizeof(int) * ((size_t)GE_THREADS + R / 4 + 8);
}

// the mandatory blocks, first in the input of the stable sort by block key: (c, c), (c, K) for every camera, then (K, K)
__global__ __launch_bounds__(TPB) void prep_grp_mandatory(int nc, uint32_t* __restrict__ seg_key, uint32_t* __restrict__ seg_where) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i > 2 * nc) return;
    const uint32_t W = (uint32_t)nc + 1;
    seg_key[i] = i == 2 * nc ? (uint32_t)nc * W + nc : (i & 1) ? (uint32_t)(i >> 1) * W + nc : (uint32_t)(i >> 1) * W + (i >> 1);
    seg_where[i] = 0xffffffffu;
}
// over the sorted items: {mandatory, run start, 0} flags for the scan
__global__ __launch_bounds__(TPB) void prep_grp_flags(int n, const uint32_t* __restrict__ key, const uint32_t* __restrict__ where, prim::I3* __restrict__ f) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    f[i] = prim::I3{where[i] == 0xffffffffu ? 1 : 0, (i == 0 || key[i - 1] != key[i]) ? 1 : 0, 0};
}
// slot of every segment (its rank among the segments in sorted order) into its lane record; the block table from the runs
__global__ __launch_bounds__(TPB) void prep_grp_slots(int n, int nc, const uint32_t* __restrict__ key, const uint32_t* __restrict__ where,
                                                      const prim::I3* __restrict__ s, uint32_t* __restrict__ laneinfo, int4* __restrict__ blk,
                                                      int* __restrict__ blk_first_item) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const prim::I3 at = s[i];  // {mandatory items before i, run starts before i, .}
    if (where[i] != 0xffffffffu) laneinfo[where[i]] = (laneinfo[where[i]] & 0xf0000000u) | (uint32_t)(i - at.a + 1);
    if (i == 0 || key[i - 1] != key[i]) {
        const uint32_t W = (uint32_t)nc + 1;
        blk[at.b] = make_int4((int)(key[i] / W), (int)(key[i] % W), i - at.a, 0);
        blk_first_item[at.b] = i;
    }
}
// count of every block = items of its run - its mandatory item; the long-block flags
__global__ __launch_bounds__(TPB) void prep_grp_blk_counts(const GrpCounters* __restrict__ gc, int n_items, const uint32_t* __restrict__ where,
                                                           const int* __restrict__ blk_first_item, int4* __restrict__ blk, int* __restrict__ longflag) {
    const int b = blockIdx.x * TPB + threadIdx.x, nblk = gc->item_totals.b;
    if (b >= n_items) return;
    if (b >= nblk) {
        longflag[b] = 0;
        return;
    }
    const int i0 = blk_first_item[b], i1 = b + 1 < nblk ? blk_first_item[b + 1] : n_items;
    const int cnt = i1 - i0 - (where[i0] == 0xffffffffu ? 1 : 0);  // (the mandatory item, if any, is the run's first: stable sort, first in the input)
    blk[b].w = cnt;
    longflag[b] = cnt > GRP_LONG ? 1 : 0;
}
__global__ __launch_bounds__(TPB) void prep_grp_long(const GrpCounters* __restrict__ gc, const int* __restrict__ longflag, const int* __restrict__ pos, int* __restrict__ longblk) {
    const int b = blockIdx.x * TPB + threadIdx.x;
    if (b < gc->item_totals.b && longflag[b]) longblk[pos[b]] = b;
}

static int ba_scratch(eacham_ctx* ctx, int which, size_t bytes, void** out) {
    BaScratch& sc = ctx->ba_scratch[which];
    if (bytes > sc.bytes) {
        if (sc.dev) {
            EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(sc.dev);
            sc.dev = nullptr;
            sc.bytes = 0;
        }
        const size_t want = bytes + bytes / 4;
        EACHAM_HIP_TRY(ctx, hipMalloc(&sc.dev, want));
        sc.bytes = want;
    }
    *out = sc.dev;
    return EACHAM_OK;
}

static int ba_prepare_device(eacham_ctx* ctx, const eacham_ba_problem* P, eacham_ba_handle** out) {
    const int nc = P->n_cams, nl = P->n_points, no = P->n_obs;
    const long long nblk_all = (long long)nc * (nc + 1) / 2;
    if (nblk_all > 0x3fffffffLL) return ctx->fail(EACHAM_ERR_UNSUPPORTED, "too many cameras (%d) for the camera-block index", nc);
    const int nblk = (int)nblk_all;
    eacham_ba_handle* h = new eacham_ba_handle();
    const auto t_begin = std::chrono::steady_clock::now();
    auto us_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    BaDev& D = h->D;
    memset(&D, 0, sizeof(D));
    D.nc = nc; D.nl = nl; D.no = no; D.n = 6 * nc + 5;
    D.nz = make_noise();
    D.n_lm_blocks = std::max(1, (nl + TPB - 1) / TPB);
    D.lpl = ctx->ba_lpl_lin > 0 ? ctx->ba_lpl_lin : (nl <= 8192 ? 8 : 2);
    D.n_ll_blocks = std::max(1, (int)(((long long)nl * D.lpl + TPB - 1) / TPB));
    D.lpl_step = ctx->ba_lpl_step > 0 ? ctx->ba_lpl_step : (nl <= 8192 ? 8 : 2);
    D.n_step_blocks = std::max(1, (int)(((long long)nl * D.lpl_step + TPB - 1) / TPB));
    hipStream_t st = ctx->stream;
    int rc = EACHAM_OK;
    std::thread plan_thread;
    std::vector<unsigned char> adj_h;
    auto fail = [&](int code) {
        if (plan_thread.joinable()) plan_thread.join();
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(ctx->stream2);
        if (h->block >= 0) ctx->ba_pool[h->block].busy = false;
        if (h->block2 >= 0) ctx->ba_pool[h->block2].busy = false;
        delete h;
        return code;
    };
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
    const int max_cam_chunks = no / TPB + nc + 1;
    // ---- arena A: everything whose size follows from (nc, nl, no) ----
    int *lm_ptr = nullptr, *cam_ptr = nullptr, *cam_obs = nullptr, *cam_lm = nullptr, *cam_chunk_ptr = nullptr, *obs_pos = nullptr, *pos_cam = nullptr, *fixed = nullptr;
    int2* cam_chunks = nullptr;
    double *lmprior = nullptr, *obs_uv = nullptr, *cam_uv = nullptr;
    unsigned *obs_cam = nullptr, *obs_lm = nullptr;
    auto layout_a = [&]() -> int {
        TRY(dev_alloc(ctx, h, &D.pose0, 12 * (size_t)nc));
        TRY(dev_alloc(ctx, h, &D.pt0, 3 * (size_t)nl));
        TRY(dev_alloc(ctx, h, &D.K0, 8));
        TRY(dev_alloc(ctx, h, &lmprior, 2 * (size_t)nl));
        TRY(dev_alloc(ctx, h, &obs_uv, 2 * (size_t)no));
        TRY(dev_alloc(ctx, h, &fixed, (size_t)nc));
        TRY(dev_alloc(ctx, h, &lm_ptr, (size_t)nl + 1));
        TRY(dev_alloc(ctx, h, &cam_ptr, (size_t)nc + 1));
        TRY(dev_alloc(ctx, h, &cam_obs, (size_t)no));
        TRY(dev_alloc(ctx, h, &cam_lm, (size_t)no));
        TRY(dev_alloc(ctx, h, &cam_chunks, (size_t)max_cam_chunks));
        TRY(dev_alloc(ctx, h, &cam_chunk_ptr, (size_t)nc + 1));
        TRY(dev_alloc(ctx, h, &obs_pos, (size_t)no));
        TRY(dev_alloc(ctx, h, &pos_cam, (size_t)no));
        TRY(dev_alloc(ctx, h, &cam_uv, 2 * (size_t)no));
        TRY(dev_alloc(ctx, h, &obs_cam, (size_t)no));
        TRY(dev_alloc(ctx, h, &obs_lm, (size_t)no));
        TRY(ba_alloc_work_a(ctx, h));
        return EACHAM_OK;
    };
    h->planning = true;
    h->arena_off = 0;
    rc = layout_a();
    if (!rc) rc = ba_block_acquire(ctx, h->arena_off, &h->block);
    if (rc) return fail(rc);
    h->arena = (char*)ctx->ba_pool[h->block].dev;
    h->planning = false;
    h->arena_off = 0;
    rc = layout_a();
    if (rc) return fail(rc);
    D.lmprior = lmprior; D.obs_uv = obs_uv; D.fixed = fixed; D.lm_ptr = lm_ptr; D.cam_ptr = cam_ptr; D.cam_obs = cam_obs; D.cam_lm = cam_lm;
    D.cam_chunks = cam_chunks; D.cam_chunk_ptr = cam_chunk_ptr; D.obs_pos = obs_pos; D.pos_cam = pos_cam; D.cam_uv = cam_uv;
    D.obs_cam = obs_cam; D.obs_lm = obs_lm;
    h->pose_init = D.pose0; h->pt_init = D.pt0; h->K_init = D.K0;
    // ---- scratch 0: the caller's arrays as they are + the temporaries of the observation sorts ----
    void* s0 = nullptr;
    uint32_t *raw_cam, *raw_pt, *kA, *kB, *vA, *vB, *kC, *vC;
    double *raw_uv, *raw_T;
    int *raw_obs, *sort_ws, *nchunk, *nchunk_ws;
    long long *pcnt, *poff, *pws;
    PrepCounters* cnt;
    unsigned char* adj_dev;
    uint32_t *gkA, *gkB, *gvA, *gvB;
    int *gcost, *growsS, *gcostS, *gcstart, *grstart, *gscan_ws, *gsort_ws;
    GrpCounters* gcnt;
    const bool try_groups = ctx->ba_schur_mode != 2 && nc < 65535 && no > 0;
    auto carve0 = [&](void* base) {
        Bump b(base);
        raw_pt = b.take<uint32_t>(no); raw_cam = b.take<uint32_t>(no); raw_uv = b.take<double>(2 * (size_t)no);
        raw_T = b.take<double>(16 * (size_t)nc); raw_obs = b.take<int>(nl);
        kA = b.take<uint32_t>(no); kB = b.take<uint32_t>(no); vA = b.take<uint32_t>(no); vB = b.take<uint32_t>(no);
        kC = b.take<uint32_t>(no); vC = b.take<uint32_t>(no);   // the camera sort's second pair (the landmark order stays: the late uv gather reads it)
        sort_ws = b.take<int>(prim::radix_ws_ints(no));
        nchunk = b.take<int>(nc); nchunk_ws = b.take<int>(prim::scan_ws_elems(nc));
        pcnt = b.take<long long>(no); poff = b.take<long long>((size_t)no + 1); pws = b.take<long long>(prim::scan_ws_elems(no));
        cnt = b.take<PrepCounters>(1);
        adj_dev = b.take<unsigned char>((size_t)nc * nc);
        // the landmark order of the group structure (ba_groups.hpp): keys, ranks, per-rank rows / costs and their prefixes
        gkA = b.take<uint32_t>(nl); gkB = b.take<uint32_t>(nl); gvA = b.take<uint32_t>(nl); gvB = b.take<uint32_t>(nl);
        gcost = b.take<int>(nl); growsS = b.take<int>(nl); gcostS = b.take<int>(nl);
        gcstart = b.take<int>((size_t)nl + 1); grstart = b.take<int>((size_t)nl + 1); gscan_ws = b.take<int>(prim::scan_ws_elems(nl));
        gsort_ws = b.take<int>(prim::radix_ws_ints(nl));
        gcnt = b.take<GrpCounters>(1);
        return b.off;
    };
    rc = ba_scratch(ctx, 0, carve0(nullptr), &s0);
    if (rc) return fail(rc);
    (void)carve0(s0);
#define HIPQ(x) do { if ((x) != hipSuccess) return fail(ctx->fail(EACHAM_ERR_HIP, "%s failed (%s:%d)", #x, __FILE__, __LINE__)); } while (0)
    HIPQ(hipMemsetAsync(cnt, 0, sizeof(PrepCounters), st));
    // the landmark ids first: their sort runs while the host stages the rest
    HIPQ(hipMemcpyAsync(raw_pt, P->obs_point, sizeof(uint32_t) * (size_t)no, hipMemcpyHostToDevice, st));
    HIPQ(hipMemcpyAsync(raw_cam, P->obs_cam, sizeof(uint32_t) * (size_t)no, hipMemcpyHostToDevice, st));
    const unsigned gobs = (unsigned)((no + TPB - 1) / TPB);
    auto bits_for = [](long long n) { int b = 1; while ((1ll << b) < n) ++b; return b; };
    if (no > 0) prep_keys_lm<<<gobs, TPB, 0, st>>>(no, nc, nl, raw_cam, raw_pt, kA, vA, cnt);
    const int w_lm = prim::radix_sort_pairs<uint32_t>(st, kA, vA, kB, vB, no, bits_for(std::max(nl, 2)), sort_ws);
    HIPQ(hipMemcpyAsync(raw_T, P->cam_T_wc, sizeof(double) * 16 * (size_t)nc, hipMemcpyHostToDevice, st));
    HIPQ(hipMemcpyAsync(raw_obs, P->point_observers, sizeof(int) * (size_t)nl, hipMemcpyHostToDevice, st));
    HIPQ(hipMemcpyAsync(D.pt0, P->points, sizeof(double) * 3 * (size_t)nl, hipMemcpyHostToDevice, st));
    HIPQ(hipMemcpyAsync(fixed, P->cam_fixed, sizeof(int) * (size_t)nc, hipMemcpyHostToDevice, st));
    prep_values<<<(unsigned)((nc + nl + TPB) / TPB), TPB, 0, st>>>(nc, nl, raw_T, raw_obs, D.pose0, lmprior, D.K0, P->K[0], P->K[1], P->K[2], P->K[3]);
    uint32_t *lm_sorted = w_lm ? kB : kA, *lm_order = w_lm ? vB : vA, *ck = w_lm ? kA : kB, *cv = w_lm ? vA : vB;  // the other pair of buffers feeds the camera sort
    if (no > 0) {
        prep_gather_lm<<<(gobs + GATHER_CHUNKS - 1) / GATHER_CHUNKS, TPB, 0, st>>>(no, nl, nc, lm_sorted, lm_order, raw_cam, obs_cam, obs_lm, lm_ptr, ck, cv, cnt);
    } else {
        HIPQ(hipMemsetAsync(lm_ptr, 0, sizeof(int) * ((size_t)nl + 1), st));
        HIPQ(hipMemsetAsync(cam_ptr, 0, sizeof(int) * ((size_t)nc + 1), st));
    }
    // the camera graph leaves for the host on the second stream as soon as the landmark grouping exists
    if (nc > 0) HIPQ(hipMemsetAsync(adj_dev, 0, (size_t)nc * nc, st));
    if (no > 0 && nc > 0) prep_cam_adjacency<<<gobs, TPB, 0, st>>>(no, nc, obs_cam, obs_lm, lm_ptr, adj_dev);
    HIPQ(hipEventRecord(ctx->ev_join, st));
    const int w_cam = prim::radix_sort_pairs<uint32_t>(st, ck, cv, kC, vC, no, bits_for(std::max(nc, 2)), sort_ws);
    if (no > 0) {
        const uint32_t *cs = w_cam ? kC : ck, *co = w_cam ? vC : cv;
        prep_gather_cam<<<gobs, TPB, 0, st>>>(no, nc, cs, co, obs_lm, cam_obs, obs_pos, pos_cam, cam_lm, cam_ptr);
        if (!try_groups) prep_pair_counts<<<gobs, TPB, 0, st>>>(no, obs_cam, obs_lm, lm_ptr, pcnt);
    }
    int w_g = 0;
    if (try_groups) {  // ba_groups.hpp steps 1-2 as far as they go without the group size: order, rows, costs, their prefixes
        HIPQ(hipMemsetAsync(gcnt, 0, sizeof(GrpCounters), st));
        const unsigned glm = (unsigned)((nl + TPB - 1) / TPB);
        prep_grp_keys<<<glm, TPB, 0, st>>>(nl, lm_ptr, obs_cam, gkA, gvA, gcost, gcnt);
        // (the key interleaves two camera ids: 2 bits_for(nc + 1) bits; landmarks without observations carry the largest key of that width)
        const int gkey_bits = std::min(32, 2 * bits_for((long long)nc + 1));
        w_g = prim::radix_sort_pairs<uint32_t>(st, gkA, gvA, gkB, gvB, nl, gkey_bits, gsort_ws);
        prep_grp_sorted<<<glm, TPB, 0, st>>>(nl, w_g ? gvB : gvA, lm_ptr, gcost, growsS, gcostS);
        prim::exclusive_scan<int>(st, gcostS, gcstart, nl, gscan_ws, &gcnt->cost_total);
        prim::exclusive_scan<int>(st, growsS, grstart, nl, gscan_ws, &gcnt->row_total);
        HIPQ(hipMemcpyAsync(grstart + nl, &gcnt->row_total, sizeof(int), hipMemcpyDeviceToDevice, st));
    }
    prep_cam_chunk_counts<<<(unsigned)((nc + TPB) / TPB), TPB, 0, st>>>(nc, cam_ptr, nchunk);
    prim::exclusive_scan<int>(st, nchunk, cam_chunk_ptr, nc, nchunk_ws, &cnt->n_cam_chunks);
    HIPQ(hipMemcpyAsync(cam_chunk_ptr + nc, &cnt->n_cam_chunks, sizeof(int), hipMemcpyDeviceToDevice, st));
    prep_cam_chunk_fill<<<(unsigned)((nc + TPB) / TPB), TPB, 0, st>>>(nc, cam_ptr, cam_chunk_ptr, cam_chunks);
    if (!try_groups) prim::exclusive_scan<long long>(st, pcnt, poff, no, pws, &cnt->n_entries);
    // ---- the host's share, beside the device's: ordering, panels, symbolic factor, level schedule (ba_plan.hpp) ----
    int hint = P->ordering;
    if (hint == EACHAM_BA_ORDER_AUTO && ctx->ba_ordering != EACHAM_BA_ORDER_AUTO) hint = ctx->ba_ordering;
    if (hint < EACHAM_BA_ORDER_AUTO || hint > EACHAM_BA_ORDER_ND) return fail(ctx->fail(EACHAM_ERR_INVALID, "unknown BA ordering %d", hint));
    adj_h.resize((size_t)nc * nc + 1);
    HIPQ(hipStreamWaitEvent(ctx->stream2, ctx->ev_join, 0));
    if (nc > 0) HIPQ(hipMemcpyAsync(adj_h.data(), adj_dev, (size_t)nc * nc, hipMemcpyDeviceToHost, ctx->stream2));
    HIPQ(hipStreamSynchronize(ctx->stream2));
    // (no HIP call in there, and no exception may leave it: a failed allocation or thread start inside the analysis comes back as
    // an error code of the call — std::terminate in a process that holds the GPU is not an answer)
    bool plan_failed = false;
    auto plan_body = [&, hint]() {
        try {
            const auto t_plan = std::chrono::steady_clock::now();
            std::vector<std::pair<int, int>> cam_edges;
            for (int c = 0; c < nc; ++c)
                for (int c2 = c + 1; c2 < nc; ++c2)
                    if (adj_h[(size_t)c * nc + c2]) cam_edges.emplace_back(c, c2);
            build_ba_plan(nc, cam_edges, hint, h->plan);
            h->prep_us[1] = us_since(t_plan);
        } catch (...) {
            plan_failed = true;
        }
    };
    try {
        plan_thread = std::thread(plan_body);
    } catch (const std::system_error&) {
        plan_body();  // no thread to be had: the analysis runs here, before the rest of the device work is queued
    }
    // ---- the measurements: uploaded on the second stream beside the kernels queued above, gathered behind them ----
    if (no > 0) {
        HIPQ(hipMemcpyAsync(raw_uv, P->obs_uv, sizeof(double) * 2 * (size_t)no, hipMemcpyHostToDevice, ctx->stream2));
        HIPQ(hipEventRecord(ctx->ev_join, ctx->stream2));
        HIPQ(hipStreamWaitEvent(st, ctx->ev_join, 0));
        prep_gather_uv<<<gobs, TPB, 0, st>>>(no, lm_order, cam_obs, raw_uv, obs_uv, cam_uv);
    }
    // ---- read-back 1: the number of pair entries sizes the next stage ----
    PrepCounters hc;
    GrpCounters hg;
    memset(&hg, 0, sizeof(hg));
    HIPQ(hipMemcpyAsync(&hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
    if (try_groups) HIPQ(hipMemcpyAsync(&hg, gcnt, sizeof(hg), hipMemcpyDeviceToHost, st));
    HIPQ(hipStreamSynchronize(st));
    if (hc.bad) return fail(ctx->fail(EACHAM_ERR_INVALID, "an observation references a camera/point out of range"));
    // the landmark-major structure applies unless a landmark is too heavy for a group (ba_groups.hpp step 2)
    const long long total_rows = (long long)no + hc.n_used;
    const int R = ctx->ba_group_rows > 0 ? ctx->ba_group_rows : grp_rows_for(total_rows);
    const bool use_groups = try_groups && total_rows <= 0x7fffffffLL && R % 4 == 0 && hg.emax <= GRP_ENT_PER_ROW * R && hg.cmax <= R / 2 && std::max(hg.cmax, 4) >= 4;
    if (try_groups && !use_groups) {  // the pair lists of rounds 1-4 after all: their counts were not taken yet
        prep_pair_counts<<<gobs, TPB, 0, st>>>(no, obs_cam, obs_lm, lm_ptr, pcnt);
        prim::exclusive_scan<long long>(st, pcnt, poff, no, pws, &cnt->n_entries);
        HIPQ(hipMemcpyAsync(&hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
        HIPQ(hipStreamSynchronize(st));
    }
    if (use_groups) hc.n_entries = 0;
    if (hc.n_entries > 0x7fffffffLL) return fail(ctx->fail(EACHAM_ERR_UNSUPPORTED, "Schur pair list too large (%lld entries)", hc.n_entries));
    const int n_entries = (int)hc.n_entries;
    h->n_landmarks_used = hc.n_used;
    D.n_cam_chunks = hc.n_cam_chunks;
    // ---- scratch 1: expansion, sort by camera block, block / chunk tables (the pair lists: only when the groups do not apply) ----
    const int max_blocks = (int)std::min<long long>(nblk, (long long)n_entries + nc);
    const int max_chunks = n_entries / PAIR_CHUNK + max_blocks + 1;
    void* s1 = nullptr;
    uint32_t *ekA = nullptr, *ekB = nullptr;
    int2 *evA = nullptr, *evB = nullptr;
    int *esort_ws, *bcount, *bfirst, *blast;
    prim::I3 *bt, *bs, *bws;
    int4 *blocks_tmp = nullptr, *chunks_tmp = nullptr;
    int w_e = 0;
    if (!use_groups) {
        auto carve1 = [&](void* base) {
            Bump b(base);
            ekA = b.take<uint32_t>(n_entries); ekB = b.take<uint32_t>(n_entries); evA = b.take<int2>(n_entries); evB = b.take<int2>(n_entries);
            esort_ws = b.take<int>(prim::radix_ws_ints(n_entries));
            bcount = b.take<int>(nblk); bfirst = b.take<int>(2 * (size_t)nblk); blast = bfirst ? bfirst + nblk : nullptr; bt = b.take<prim::I3>(nblk); bs = b.take<prim::I3>(nblk); bws = b.take<prim::I3>(prim::scan_ws_elems(nblk));
            blocks_tmp = b.take<int4>(max_blocks); chunks_tmp = b.take<int4>(max_chunks);
            return b.off;
        };
        rc = ba_scratch(ctx, 1, carve1(nullptr), &s1);
        if (rc) return fail(rc);
        (void)carve1(s1);
        HIPQ(hipMemsetAsync(bfirst, 0, sizeof(int) * 2 * (size_t)std::max(nblk, 1), st));
        if (no > 0) prep_expand<<<gobs, TPB, 0, st>>>(no, nc, obs_cam, obs_lm, lm_ptr, obs_pos, poff, ekA, evA);
        w_e = prim::radix_sort_pairs<int2>(st, ekA, evA, ekB, evB, n_entries, bits_for(std::max(nblk, 2)), esort_ws);
        if (n_entries > 0) prep_block_runs<<<(unsigned)((n_entries + TPB - 1) / TPB), TPB, 0, st>>>(n_entries, w_e ? ekB : ekA, bfirst, blast);
        const unsigned gblk = (unsigned)((nblk + TPB) / TPB);
        if (nblk > 0) prep_block_counts<<<gblk, TPB, 0, st>>>(nc, nblk, bfirst, blast, bcount, bt);
        prim::exclusive_scan<prim::I3>(st, bt, bs, nblk, bws, &cnt->totals);
        if (nblk > 0) prep_block_fill<<<gblk, TPB, 0, st>>>(nc, nblk, bcount, bs, blocks_tmp, chunks_tmp);
        // ---- read-back 2: the table sizes ----
        HIPQ(hipMemcpyAsync(&hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
        HIPQ(hipStreamSynchronize(st));
        if (hc.totals.a != n_entries) return fail(ctx->fail(EACHAM_ERR_HIP, "BA structure build: %d pair entries counted, %d placed", n_entries, hc.totals.a));
        D.n_blocks = hc.totals.b;
        D.n_chunks = hc.totals.c;
    }
    // ---- the landmark-major structure (ba_groups.hpp steps 2-4 on the device), first half: groups, padded per-group arrays,
    // the counting pass over every group's entries ----
    const int nu = hc.n_used, LMAXg = R / 4, R0g = R - std::max(hg.cmax, 4) + 1;
    const int ng_max = use_groups ? hg.cost_total / R0g + 1 : 0;
    int g_emax = 64;
    while (g_emax < GRP_ENT_PER_ROW * R) g_emax <<= 1;
    BaGroup* tg_groups = nullptr;
    int *tg_lmid = nullptr, *tg_lmrow = nullptr, *tg_lm0 = nullptr;
    int2* tg_rowinfo = nullptr;
    double* tg_uv = nullptr;
    prim::I3 *tg_counts = nullptr, *tg_bases = nullptr, *tg_ws = nullptr;
    int n_items = 0;
    if (use_groups) {
        if (g_emax > GE_MAXE || R > 511) return fail(ctx->fail(EACHAM_ERR_UNSUPPORTED, "group size %d rows is beyond the device construction", R));
        void* s2 = nullptr;
        auto carve2 = [&](void* base) {
            Bump b(base);
            tg_groups = b.take<BaGroup>(ng_max); tg_lmid = b.take<int>((size_t)ng_max * LMAXg); tg_lmrow = b.take<int>((size_t)ng_max * LMAXg);
            tg_rowinfo = b.take<int2>((size_t)ng_max * R); tg_uv = b.take<double>(2 * (size_t)ng_max * R);
            tg_lm0 = b.take<int>((size_t)ng_max + 2);
            tg_counts = b.take<prim::I3>(ng_max); tg_bases = b.take<prim::I3>(ng_max); tg_ws = b.take<prim::I3>(prim::scan_ws_elems(ng_max));
            return b.off;
        };
        rc = ba_scratch(ctx, 2, carve2(nullptr), &s2);
        if (rc) return fail(rc);
        (void)carve2(s2);
        HIPQ(hipMemsetAsync(tg_groups, 0, sizeof(BaGroup) * (size_t)ng_max, st));
        HIPQ(hipMemsetAsync(tg_lmid, 0xff, sizeof(int) * (size_t)ng_max * LMAXg, st));
        HIPQ(hipMemsetAsync(tg_lmrow, 0, sizeof(int) * (size_t)ng_max * LMAXg, st));
        HIPQ(hipMemsetAsync(tg_rowinfo, 0xff, sizeof(int2) * (size_t)ng_max * R, st));
        HIPQ(hipMemsetAsync(tg_uv, 0, sizeof(double) * 2 * (size_t)ng_max * R, st));
        const uint32_t* g_sorted = w_g ? gvB : gvA;
        const unsigned gnu = (unsigned)((nu + TPB - 1) / TPB);
        if (nu > 0) {
            prep_grp_bounds<<<gnu, TPB, 0, st>>>(nu, gcstart, R0g, tg_lm0, gcnt);
            prep_grp_records<<<(unsigned)((ng_max + TPB - 1) / TPB), TPB, 0, st>>>(gcnt, tg_lm0, grstart, tg_groups);
            prep_grp_fill<<<gnu, TPB, 0, st>>>(nu, nc, R, R0g, g_sorted, gcstart, grstart, tg_lm0, lm_ptr, obs_cam, obs_uv, tg_lmid, tg_lmrow, tg_rowinfo, tg_uv);
        }
        if (hg.any_dup) {  // the general form: sorts every group's entries
            const size_t ge_lds = grp_entries_lds_bytes(g_emax, R);
            HIPQ(hipFuncSetAttribute((const void*)prep_grp_entries<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ge_lds));
            HIPQ(hipFuncSetAttribute((const void*)prep_grp_entries<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ge_lds));
            prep_grp_entries<0><<<ng_max, GE_THREADS, ge_lds, st>>>(g_emax, nc, R, gcnt, tg_groups, tg_rowinfo, tg_lmrow, tg_counts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
        } else {
            const size_t gf_lds = grp_entries_fast_lds_bytes(GRP_ENT_PER_ROW * R, R);
            HIPQ(hipFuncSetAttribute((const void*)prep_grp_entries_fast<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gf_lds));
            HIPQ(hipFuncSetAttribute((const void*)prep_grp_entries_fast<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gf_lds));
            prep_grp_entries_fast<0><<<ng_max, GE_THREADS, gf_lds, st>>>(GRP_ENT_PER_ROW * R, nc, R, gcnt, tg_groups, tg_rowinfo, tg_lmrow, tg_counts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
        }
        prim::exclusive_scan<prim::I3>(st, tg_counts, tg_bases, ng_max, tg_ws, &gcnt->totals);
        // ---- read-back 2: groups, chunks, entry rows, segments ----
        HIPQ(hipMemcpyAsync(&hg, gcnt, sizeof(hg), hipMemcpyDeviceToHost, st));
        HIPQ(hipStreamSynchronize(st));
        if (nu == 0) hg.ng = 0;
        if (hg.ng > ng_max) return fail(ctx->fail(EACHAM_ERR_HIP, "BA structure build: %d groups, %d expected at most", hg.ng, ng_max));
        D.g_rows = R; D.g_ngroups = hg.ng; D.g_nchunks = hg.totals.a; D.g_nent4 = hg.totals.b; D.g_nparts = hg.totals.c;
        n_items = 2 * nc + 1 + hg.totals.c;
    }
    // ---- second half of the landmark-major structure: the writing pass over the groups, the slots, the block table. It needs the
    // group counters only, not the plan: it runs HERE, beside the host's analysis (which has become the longer of the two: 0.65 ms
    // against 0.45 ms of device work from the camera graph's read-back on S200, 1.5 against 0.9 ms on config 4), into staging buffers;
    // what it produced is copied into the arena once the plan has sized it (the 17 MB of entries: ~10 us device to device). Until the
    // second half of round 5 it waited for the join: 0.3 ms of every S200 call with the device idle. ----
    BaChunk* sg_chunks = nullptr;
    uint32_t *sg_ent = nullptr, *sg_laneinfo = nullptr;
    int4* sg_blk = nullptr;
    int* sg_longblk = nullptr;
    if (use_groups) {
        const int ngf = D.g_ngroups;
        void* s3 = nullptr;
        auto carve3 = [&](void* base) {
            Bump b(base);
            sg_chunks = b.take<BaChunk>(D.g_nchunks); sg_ent = b.take<uint32_t>(256 * (size_t)D.g_nent4); sg_laneinfo = b.take<uint32_t>(64 * (size_t)D.g_nchunks);
            sg_blk = b.take<int4>(n_items); sg_longblk = b.take<int>((size_t)D.g_nparts / GRP_LONG + 1);
            return b.off;
        };
        rc = ba_scratch(ctx, 3, carve3(nullptr), &s3);
        if (rc) return fail(rc);
        (void)carve3(s3);
        void* s1 = nullptr;
        uint32_t *ikA, *ikB, *iwA, *iwB;
        int *isort_ws, *blk_first, *longflag, *longpos, *long_ws;
        prim::I3 *ifl, *isc, *iws;
        auto carve1g = [&](void* base) {
            Bump b(base);
            ikA = b.take<uint32_t>(n_items); ikB = b.take<uint32_t>(n_items); iwA = b.take<uint32_t>(n_items); iwB = b.take<uint32_t>(n_items);
            isort_ws = b.take<int>(prim::radix_ws_ints(n_items));
            ifl = b.take<prim::I3>(n_items); isc = b.take<prim::I3>(n_items); iws = b.take<prim::I3>(prim::scan_ws_elems(n_items));
            blk_first = b.take<int>(n_items); longflag = b.take<int>(n_items); longpos = b.take<int>(n_items); long_ws = b.take<int>(prim::scan_ws_elems(n_items));
            return b.off;
        };
        rc = ba_scratch(ctx, 1, carve1g(nullptr), &s1);
        if (rc) return fail(rc);
        (void)carve1g(s1);
        const int n_mand = 2 * nc + 1;
        prep_grp_mandatory<<<(unsigned)((n_mand + TPB - 1) / TPB), TPB, 0, st>>>(nc, ikA, iwA);
        if (ngf > 0) {
            if (hg.any_dup)
                prep_grp_entries<1><<<ngf, GE_THREADS, grp_entries_lds_bytes(g_emax, R), st>>>(g_emax, nc, R, gcnt, tg_groups, tg_rowinfo, tg_lmrow, nullptr, tg_bases, sg_chunks, sg_ent, sg_laneinfo, ikA, iwA, n_mand);
            else
                prep_grp_entries_fast<1><<<ngf, GE_THREADS, grp_entries_fast_lds_bytes(GRP_ENT_PER_ROW * R, R), st>>>(GRP_ENT_PER_ROW * R, nc, R, gcnt, tg_groups, tg_rowinfo, tg_lmrow, nullptr, tg_bases, sg_chunks, sg_ent, sg_laneinfo, ikA, iwA, n_mand);
        }
        const int w_i = prim::radix_sort_pairs<uint32_t>(st, ikA, iwA, ikB, iwB, n_items, bits_for((long long)(nc + 1) * (nc + 1)), isort_ws);
        const uint32_t *skey = w_i ? ikB : ikA, *swhere = w_i ? iwB : iwA;
        const unsigned git = (unsigned)((n_items + TPB - 1) / TPB);
        prep_grp_flags<<<git, TPB, 0, st>>>(n_items, skey, swhere, ifl);
        prim::exclusive_scan<prim::I3>(st, ifl, isc, n_items, iws, &gcnt->item_totals);
        prep_grp_slots<<<git, TPB, 0, st>>>(n_items, nc, skey, swhere, isc, sg_laneinfo, sg_blk, blk_first);
        prep_grp_blk_counts<<<git, TPB, 0, st>>>(gcnt, n_items, swhere, blk_first, sg_blk, longflag);
        prim::exclusive_scan<int>(st, longflag, longpos, n_items, long_ws, &gcnt->n_long);
        prep_grp_long<<<git, TPB, 0, st>>>(gcnt, longflag, longpos, sg_longblk);
        // ---- read-back 3: blocks, long blocks ----
        HIPQ(hipMemcpyAsync(&hg, gcnt, sizeof(hg), hipMemcpyDeviceToHost, st));
        HIPQ(hipStreamSynchronize(st));
        D.g_nblk = hg.item_totals.b; D.g_nlong = hg.n_long;
    }
    h->prep_us[0] = us_since(t_begin);
    if (plan_thread.joinable()) plan_thread.join();  // prep_us[1] = the plan's own time; what of it was not hidden behind the device shows in [2]
    if (plan_failed) return fail(ctx->fail(EACHAM_ERR_HIP, "BA preparation: the analysis of the reduced system ran out of memory or threads"));
    const auto t_upload = std::chrono::steady_clock::now();
    const BaPlan& plan = h->plan;
    D.sp_npan = plan.npan; D.sp_ntiles = plan.ntiles; D.sp_posK = plan.posK; D.sp_rhs_row = plan.rhs_row;
    D.sp_n_pad = (int)plan.pad_cols.size();
    std::vector<int2> bs_ent(plan.bs_ent.size());
    for (size_t e = 0; e < bs_ent.size(); ++e) bs_ent[e] = make_int2(plan.bs_ent[e].first, plan.bs_ent[e].second);
    // ---- arena B: the pair lists, the plan's tables, what is sized by them ----
    int2* pair_entries = nullptr;
    int4 *pair_chunks = nullptr, *blocks = nullptr;
    BaGroup* fg_groups = nullptr;
    int *fg_lmid = nullptr, *fg_lmrow = nullptr, *fg_longblk = nullptr;
    int2* fg_rowinfo = nullptr;
    double* fg_uv = nullptr;
    BaChunk* fg_chunks = nullptr;
    uint32_t *fg_ent = nullptr, *fg_laneinfo = nullptr;
    int4* fg_blk = nullptr;
    const int ngf = D.g_ngroups;
    size_t plan_lo = 0, plan_hi = 0;   // the plan's eleven tables in the arena: sent as ONE copy of their host image
    auto layout_b = [&]() -> int {
        TRY(dev_alloc(ctx, h, &pair_entries, (size_t)n_entries));
        TRY(dev_alloc(ctx, h, &pair_chunks, (size_t)D.n_chunks));
        TRY(dev_alloc(ctx, h, &blocks, (size_t)D.n_blocks));
        if (use_groups) {
            TRY(dev_alloc(ctx, h, &fg_groups, (size_t)ngf));
            TRY(dev_alloc(ctx, h, &fg_lmid, (size_t)ngf * LMAXg));
            TRY(dev_alloc(ctx, h, &fg_lmrow, (size_t)ngf * LMAXg));
            TRY(dev_alloc(ctx, h, &fg_rowinfo, (size_t)ngf * R));
            TRY(dev_alloc(ctx, h, &fg_uv, 2 * (size_t)ngf * R));
            TRY(dev_alloc(ctx, h, &fg_chunks, (size_t)D.g_nchunks));
            TRY(dev_alloc(ctx, h, &fg_ent, 256 * (size_t)D.g_nent4));
            TRY(dev_alloc(ctx, h, &fg_laneinfo, 64 * (size_t)D.g_nchunks));
            TRY(dev_alloc(ctx, h, &fg_blk, (size_t)n_items));                      // (blocks <= items)
            TRY(dev_alloc(ctx, h, &fg_longblk, (size_t)D.g_nparts / GRP_LONG + 1));  // (a long block holds more than GRP_LONG segments)
        }
        plan_lo = h->arena_off;
        TRY(ba_upload_plan(ctx, h, plan, bs_ent));
        plan_hi = h->arena_off;
        TRY(ba_alloc_work_b(ctx, h));
        return EACHAM_OK;
    };
    char* arena_a = h->arena;
    const size_t off_a = h->arena_off;
    h->planning = true;
    h->arena = nullptr;
    h->arena_off = 0;
    rc = layout_b();
    if (!rc) rc = ba_block_acquire(ctx, h->arena_off, &h->block2);
    if (rc) return fail(rc);
    h->arena = (char*)ctx->ba_pool[h->block2].dev;
    h->planning = false;
    h->arena_off = 0;
    h->stage_base = plan_lo;
    h->stage.assign(plan_hi - plan_lo, 0);
    rc = layout_b();
    if (rc) return fail(rc);
    if (!h->stage.empty()) HIPQ(hipMemcpyAsync(h->arena + h->stage_base, h->stage.data(), h->stage.size(), hipMemcpyHostToDevice, st));
    (void)arena_a; (void)off_a;
    if (n_entries > 0) HIPQ(hipMemcpyAsync(pair_entries, w_e ? evB : evA, sizeof(int2) * (size_t)n_entries, hipMemcpyDeviceToDevice, st));
    if (D.n_chunks > 0) HIPQ(hipMemcpyAsync(pair_chunks, chunks_tmp, sizeof(int4) * (size_t)D.n_chunks, hipMemcpyDeviceToDevice, st));
    if (D.n_blocks > 0) HIPQ(hipMemcpyAsync(blocks, blocks_tmp, sizeof(int4) * (size_t)D.n_blocks, hipMemcpyDeviceToDevice, st));
    D.pair_entries = pair_entries; D.pair_chunks = pair_chunks; D.blocks = blocks;
    if (use_groups) {   // what the two halves staged -> the arena
        if (ngf > 0) {
            HIPQ(hipMemcpyAsync(fg_lmid, tg_lmid, sizeof(int) * (size_t)ngf * LMAXg, hipMemcpyDeviceToDevice, st));
            HIPQ(hipMemcpyAsync(fg_lmrow, tg_lmrow, sizeof(int) * (size_t)ngf * LMAXg, hipMemcpyDeviceToDevice, st));
            HIPQ(hipMemcpyAsync(fg_rowinfo, tg_rowinfo, sizeof(int2) * (size_t)ngf * R, hipMemcpyDeviceToDevice, st));
            HIPQ(hipMemcpyAsync(fg_uv, tg_uv, sizeof(double) * 2 * (size_t)ngf * R, hipMemcpyDeviceToDevice, st));
            HIPQ(hipMemcpyAsync(fg_groups, tg_groups, sizeof(BaGroup) * (size_t)ngf, hipMemcpyDeviceToDevice, st));
        }
        if (D.g_nchunks > 0) {
            HIPQ(hipMemcpyAsync(fg_chunks, sg_chunks, sizeof(BaChunk) * (size_t)D.g_nchunks, hipMemcpyDeviceToDevice, st));
            HIPQ(hipMemcpyAsync(fg_laneinfo, sg_laneinfo, sizeof(uint32_t) * 64 * (size_t)D.g_nchunks, hipMemcpyDeviceToDevice, st));
        }
        if (D.g_nent4 > 0) HIPQ(hipMemcpyAsync(fg_ent, sg_ent, sizeof(uint32_t) * 256 * (size_t)D.g_nent4, hipMemcpyDeviceToDevice, st));
        if (n_items > 0) HIPQ(hipMemcpyAsync(fg_blk, sg_blk, sizeof(int4) * (size_t)n_items, hipMemcpyDeviceToDevice, st));
        HIPQ(hipMemcpyAsync(fg_longblk, sg_longblk, sizeof(int) * ((size_t)D.g_nparts / GRP_LONG + 1), hipMemcpyDeviceToDevice, st));
        D.g_groups = fg_groups; D.g_lmid = fg_lmid; D.g_lmrow = fg_lmrow; D.g_rowinfo = fg_rowinfo; D.g_uv = fg_uv; D.g_chunks = fg_chunks;
        D.g_ent = fg_ent; D.g_laneinfo = fg_laneinfo; D.g_blk = fg_blk; D.g_longblk = fg_longblk;
    }
    HIPQ(hipStreamSynchronize(st));  // the plan's tables were copied out of host vectors
    h->prep_us[2] = us_since(t_upload);
#undef HIPQ
#undef TRY
    h->bytes_linearize = (size_t)no * (24 + 144) + (size_t)no * 24 + (size_t)nl * (24 + LMLIN * 8) + (size_t)nc * (96 + CAMLIN * 8);
    h->bytes_try = (size_t)no * (144 * 2 + 144 + 144) + (size_t)n_entries * 8 + (size_t)nl * (LMLIN * 8 * 3 + 48) +
                   (size_t)no * 24 + (size_t)D.n * D.n * 8;
    *out = h;
    return EACHAM_OK;
}

static int ba_prepare(eacham_ctx* ctx, const eacham_ba_problem* P, eacham_ba_handle** out, bool lm_direct = false) {
    if (!P || P->n_cams < 0 || P->n_points < 0 || P->n_obs < 0) return ctx->fail(EACHAM_ERR_INVALID, "bad BA problem");
    if (P->n_obs > 0 && (!P->obs_cam || !P->obs_point || !P->obs_uv)) return ctx->fail(EACHAM_ERR_INVALID, "null observation arrays");
    if ((P->n_cams > 0 && (!P->cam_T_wc || !P->cam_fixed)) || (P->n_points > 0 && (!P->points || !P->point_observers)))
        return ctx->fail(EACHAM_ERR_INVALID, "null camera/point arrays");
    // a local window (the reference's per-frame call, ~10 k observations) is cheaper through the host loops: the device
    // construction is ~45 dependent launches and three read-backs whatever the size
    const bool device = ctx->ba_prepare_mode == 2 || (ctx->ba_prepare_mode == 0 && P->n_obs >= 65536);
    // the dense form of the Schur stage (ba_window.hpp) is a measured alternative, not the default: EACHAM_BA_SCHUR=dense selects it
    // (it carries the direct Levenberg-Marquardt solve only). On the 19-camera windows of the TUM stand-in its one launch takes
    // 39 us where ba_linearize + ba_eliminate + ba_schur_pairs take 28 (profiles/r05_ba_windows_dense_form.txt).
    (void)lm_direct;
    const bool allow_dense = ctx->ba_schur_mode == 3;
    return device ? ba_prepare_device(ctx, P, out) : ba_prepare_host(ctx, P, out, allow_dense);
}

static void ba_release(eacham_ctx* ctx, eacham_ba_handle* h) {
    if (!h) return;
    (void)hipStreamSynchronize(ctx->stream);
    if (h->block >= 0) ctx->ba_pool[h->block].busy = false;  // the arena and its pinned scalars stay with the context
    if (h->block2 >= 0) ctx->ba_pool[h->block2].busy = false;
    delete h;
}

// resets the values to the uploaded initial state
static int ba_reset(eacham_ctx* ctx, eacham_ba_handle* h) {
    h->finish_pending = false;
    BaDev& D = h->D;
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(D.pose, h->pose_init, sizeof(double) * 12 * (size_t)D.nc, hipMemcpyDeviceToDevice, ctx->stream));
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(D.pt, h->pt_init, sizeof(double) * 3 * (size_t)D.nl, hipMemcpyDeviceToDevice, ctx->stream));
    EACHAM_HIP_TRY(ctx, hipMemcpyAsync(D.Kc, h->K_init, sizeof(double) * 5, hipMemcpyDeviceToDevice, ctx->stream));
    return EACHAM_OK;
}

// graph.error at (pose, pt, K) -> scal[0]
// the three kernels that walk a landmark's observations, in the lanes-per-landmark variant of the problem (BaDev::lpl)
// error at (pose, pt, Kc) + the final sums (made by the launch's last workgroup): ba_retract_cameras ran before it
static void launch_error_landmarks(eacham_ctx* ctx, eacham_ba_handle* h, const double* pose, const double* pt, const double* Kc, int n_lin) {
    const BaDev& D = h->D;
    const double ticket = (double)++h->ticket;
    switch (D.lpl_step) {
        case 8: ba_error_landmarks<8><<<D.n_step_blocks, TPB, 0, ctx->stream>>>(D, pose, pt, Kc, h->err_cam, h->lin_cam, n_lin, ticket); break;
        case 4: ba_error_landmarks<4><<<D.n_step_blocks, TPB, 0, ctx->stream>>>(D, pose, pt, Kc, h->err_cam, h->lin_cam, n_lin, ticket); break;
        case 2: ba_error_landmarks<2><<<D.n_step_blocks, TPB, 0, ctx->stream>>>(D, pose, pt, Kc, h->err_cam, h->lin_cam, n_lin, ticket); break;
        default: ba_error_landmarks<1><<<D.n_step_blocks, TPB, 0, ctx->stream>>>(D, pose, pt, Kc, h->err_cam, h->lin_cam, n_lin, ticket);
    }
}
static void launch_linearize_both(eacham_ctx* ctx, const BaDev& D, double* clpart) {
    const int ncb = D.nc * LSEG;
    switch (D.lpl) {
        case 8: ba_linearize<8><<<ncb + D.n_ll_blocks, TPB, 0, ctx->stream>>>(D, clpart, ncb); break;
        case 4: ba_linearize<4><<<ncb + D.n_ll_blocks, TPB, 0, ctx->stream>>>(D, clpart, ncb); break;
        case 2: ba_linearize<2><<<ncb + D.n_ll_blocks, TPB, 0, ctx->stream>>>(D, clpart, ncb); break;
        default: ba_linearize<1><<<ncb + D.n_ll_blocks, TPB, 0, ctx->stream>>>(D, clpart, ncb);
    }
}
static void launch_step_landmarks(eacham_ctx* ctx, eacham_ba_handle* h, double lambda) {
    const BaDev& D = h->D;
    const int grid = D.n_step_blocks + (D.nc + 1 + TPB - 1) / TPB;  // + the camera retraction
    const double ticket = (double)++h->ticket;
    switch (D.lpl_step) {
        case 8: ba_step_landmarks<8><<<grid, TPB, 0, ctx->stream>>>(D, lambda, h->err_cam, h->lin_cam, ticket); break;
        case 4: ba_step_landmarks<4><<<grid, TPB, 0, ctx->stream>>>(D, lambda, h->err_cam, h->lin_cam, ticket); break;
        case 2: ba_step_landmarks<2><<<grid, TPB, 0, ctx->stream>>>(D, lambda, h->err_cam, h->lin_cam, ticket); break;
        default: ba_step_landmarks<1><<<grid, TPB, 0, ctx->stream>>>(D, lambda, h->err_cam, h->lin_cam, ticket);
    }
}

static void launch_error(eacham_ctx* ctx, eacham_ba_handle* h, const double* pose, const double* pt, const double* Kc) {
    BaDev& D = h->D;
    ProfileScope ps(ctx, EACHAM_KERNEL_BA_ERROR);
    ba_retract_cameras<<<(D.nc + 1 + 63) / 64, 64, 0, ctx->stream>>>(D, 0.0, pose, D.pose_new, Kc, D.K_new, 0, h->err_cam, h->lin_cam);
    launch_error_landmarks(ctx, h, pose, pt, Kc, 0);
}

static void launch_linearize(eacham_ctx* ctx, eacham_ba_handle* h) {
    BaDev& D = h->D;
    ProfileScope ps(ctx, EACHAM_KERNEL_BA_LINEARIZE);
    launch_linearize_both(ctx, D, h->kpart);
    h->finish_pending = true;  // rides in the next try's first launch (launch_try) or is flushed by finish_linearize_now
}
static void finish_linearize_now(eacham_ctx* ctx, eacham_ba_handle* h) {
    if (!h->finish_pending) return;
    ProfileScope ps(ctx, EACHAM_KERNEL_BA_LINEARIZE);
    ba_finish_linearize<<<KLIN + h->D.nc, 64, 0, ctx->stream>>>(h->D, h->kpart);
    h->finish_pending = false;
}

// one tryLambda(): builds and solves the damped system, writes tentative values and
// scal = {new error, linearised cost change, flags}
static int launch_try(eacham_ctx* ctx, eacham_ba_handle* h, double lambda, double* S_copy /* host, optional */) {
    BaDev& D = h->D;
    const int n = D.n;
    {
        ProfileScope ps(ctx, EACHAM_KERNEL_BA_SCHUR);
        const int n_finish = h->finish_pending ? KLIN + D.nc : 0;
        h->finish_pending = false;
        if (D.w_rows > 0) {  // a local window: from the values to the groups' dense partials in one launch, then their sum
            ba_schur_dense<<<D.w_ngroups, WIN_TPB, win_lds_bytes(D.nc, D.w_rows), ctx->stream>>>(D, lambda);
            const unsigned neb = (unsigned)((win_nblk(D.nc) * 36 * WIN_ASM_LANES + TPB - 1) / TPB);
            ba_assemble_dense<<<neb + 1, TPB, 0, ctx->stream>>>(D, lambda, neb);
        } else if (D.g_rows > 0) {  // landmark groups: elimination, Et (in LDS) and the pair products in one launch
            if (D.g_rows <= TPB) ba_schur_groups<1><<<std::max(D.g_ngroups, 1) + n_finish, TPB, schur_groups_lds_bytes(D.g_rows), ctx->stream>>>(D, lambda, h->kpart, n_finish);
            else ba_schur_groups<2><<<std::max(D.g_ngroups, 1) + n_finish, TPB, schur_groups_lds_bytes(D.g_rows), ctx->stream>>>(D, lambda, h->kpart, n_finish);
            const unsigned nbg = (unsigned)((D.g_nblk + ASM_TPB / 64 - 1) / (ASM_TPB / 64));  // a wave per block
            ba_assemble_groups<<<D.g_nlong + nbg + 1, ASM_TPB, 0, ctx->stream>>>(D, lambda, nbg);
        } else {
            ba_eliminate<<<D.n_cam_chunks + D.n_lm_blocks + n_finish, TPB, 0, ctx->stream>>>(D, lambda, h->kpart, D.n_lm_blocks, n_finish);
            if (D.n_chunks > 0) ba_schur_pairs<<<((D.n_chunks + TPB / 64 - 1) / (TPB / 64) + 7) / 8 * 8, TPB, 0, ctx->stream>>>(D);
            const unsigned nbg = (unsigned)(((long long)D.n_blocks * 36 + TPB - 1) / TPB);
            ba_assemble<<<nbg + D.nc + 2, TPB, 0, ctx->stream>>>(D, lambda, nbg);  // camera blocks | border per camera | K corner | padding
        }
    }
    if (S_copy) {  // diagnostic read-back: S and its right-hand side in the caller's order, from the tiles
        const BaPlan& pl = h->plan;
        std::vector<double> tiles((size_t)pl.ntiles * TILE);
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        EACHAM_HIP_TRY(ctx, hipMemcpy(tiles.data(), D.T, sizeof(double) * tiles.size(), hipMemcpyDeviceToHost));
        std::vector<int> colpos((size_t)n);
        for (int c = 0; c < D.nc; ++c)
            for (int a = 0; a < 6; ++a) colpos[6 * (size_t)c + a] = pl.pos[c] + a;
        for (int k = 0; k < 5; ++k) colpos[6 * (size_t)D.nc + k] = pl.posK + k;
        auto at = [&](int r, int q) -> double {
            if (r < q) std::swap(r, q);
            const int t = pl.tile(r >> 6, q >> 6);
            return t < 0 ? 0.0 : tiles[(size_t)t * TILE + (r & 63) * PB + (q & 63)];
        };
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) S_copy[(size_t)i * n + j] = at(colpos[i], colpos[j]);
        for (int j = 0; j < n; ++j) S_copy[(size_t)n * n + j] = at(pl.rhs_row, colpos[j]);
    }
#ifdef EXP_BA_FAULT
    {
        const char* f = getenv("EACHAM_FAULT");
        const int kind = !f ? 0 : (!strcmp(f, "handoff") ? 1 : (!strcmp(f, "progress") ? 2 : 0));
        EACHAM_HIP_TRY(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(eacham::g_ba_fault), &kind, sizeof(int), 0, hipMemcpyHostToDevice, ctx->stream));
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
#endif
    {
        // the factorisation, level by level of the elimination tree, and the back-substitution (one launch)
        ProfileScope ps(ctx, EACHAM_KERNEL_BA_SOLVE);
        const BaPlan& pl = h->plan;
        sp_diag<<<(unsigned)pl.leaves.size(), TPB, 0, ctx->stream>>>(D.T, h->sp_leaves, D.sp_diag_tile, D.Winv, D.Wops, D.Xrow, D.flags);
        for (const auto& la : pl.launches)
            sp_level<<<(unsigned)la.count, TPB, 0, ctx->stream>>>(D.T, h->sp_items + la.first, h->sp_srcs, la.n_first, D.Winv, D.Wops, D.Xrow, D.flags);
        sp_backsolve<<<(unsigned)pl.npan, BS_THREADS, 0, ctx->stream>>>(D.T, h->bs_order, h->bs_ptr, h->bs_ent, D.Winv, D.Xrow, D.zsol,
                                                                        h->sp_col_dest, D.delta_c, D.flags);
    }
    {
        ProfileScope ps(ctx, EACHAM_KERNEL_BA_ERROR);
        launch_step_landmarks(ctx, h, lambda);  // back-substitution, retraction, error at the tentative values, final sums
    }
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    return EACHAM_OK;
}

// tryLambda() with the iterative solve of BundleAdjuster.cpp:192-200: PCG + block-Jacobi on the damped full system
// (needs E: the linearisation ran with store_E). Same outputs as launch_try. *pcg_iterations (optional) += iterations.
static int launch_try_pcg(eacham_ctx* ctx, eacham_ba_handle* h, double lambda, long long* pcg_iterations) {
    BaDev& D = h->D;
    finish_linearize_now(ctx, h);
    const int n_update_blocks = (D.nl + D.nc + 1 + TPB - 1) / TPB;
    {
        ProfileScope ps(ctx, EACHAM_KERNEL_BA_SOLVE);
        pcg_setup_landmarks<<<D.n_lm_blocks, TPB, 0, ctx->stream>>>(D, lambda);
        pcg_setup_cameras<<<1, 1024, 0, ctx->stream>>>(D, lambda);
        double st[2] = {0.0, 0.0};
        for (int done_it = 0; done_it < PCG_MAX_IT; done_it += PCG_BATCH) {
            for (int k = 0; k < PCG_BATCH; ++k) {
                pcg_apply_landmarks<<<D.n_lm_blocks, TPB, 0, ctx->stream>>>(D, lambda);
                if (D.n_cam_chunks > 0) pcg_apply_cam_chunks<<<D.n_cam_chunks, TPB, 0, ctx->stream>>>(D);
                pcg_apply_finish<<<1, 1024, 0, ctx->stream>>>(D, lambda);
                pcg_update<<<n_update_blocks, TPB, 0, ctx->stream>>>(D);
                pcg_update_finish<<<1, 1024, 0, ctx->stream>>>(D, n_update_blocks);
            }
            EACHAM_HIP_TRY(ctx, hipMemcpyAsync(st, D.pcg_s + 4, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
            EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (st[0] != 0.0) break;
        }
        if (pcg_iterations) *pcg_iterations += (long long)st[1];
    }
    {
        ProfileScope ps(ctx, EACHAM_KERNEL_BA_ERROR);
        pcg_landmark_step<<<D.n_lm_blocks, TPB, 0, ctx->stream>>>(D, lambda);
        ba_retract_cameras<<<(D.nc + 1 + 63) / 64, 64, 0, ctx->stream>>>(D, lambda, D.pose, D.pose_new, D.Kc, D.K_new, 1, h->err_cam, h->lin_cam);
        launch_error_landmarks(ctx, h, D.pose_new, D.pt_new, D.K_new, D.n_lm_blocks /* pcg_landmark_step's grid */);
    }
    EACHAM_HIP_TRY(ctx, hipGetLastError());
    return EACHAM_OK;
}

static int read_scal(eacham_ctx* ctx, eacham_ba_handle* h, double* out3) {
    // ba_final_sums stored them into the pinned block itself, its ticket last. The LM loop is one host round trip per
    // tryLambda(): waking up from hipStreamSynchronize and only then deciding and launching left the device idle for
    // ~25 us per inner iteration on S200, so the host first watches the ticket for a bounded while (the stream is
    // still in order: what is launched next runs behind that kernel) and only then falls back to the blocking wait.
    {
        volatile double* sh = h->scal_host;
        const double want = (double)h->ticket;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (int spin = 0;; ++spin) {
            if (sh[3] == want) { seen = true; break; }
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
        }
        if (seen) std::atomic_thread_fence(std::memory_order_acquire);
        else EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    for (int k = 0; k < 3; ++k) out3[k] = h->scal_host[k];
    return EACHAM_OK;
}

// LevenbergMarquardtOptimizer::optimize with SetCeresDefaults + the reference's overrides
// (BundleAdjuster.cpp:184-190, :216); control flow as in SURVEY.md Appendix A.4.
static int ba_run(eacham_ctx* ctx, eacham_ba_handle* h, const eacham_ba_options* O, eacham_ba_result* R) {
    if (!O || !R || (h->D.nc > 0 && !R->cam_T_wc) || (h->D.nl > 0 && !R->points)) return ctx->fail(EACHAM_ERR_INVALID, "null BA options/result");
    BaDev& D = h->D;
    R->trace_len = 0;
    R->outer_iterations = R->inner_iterations = 0;
    R->final_lambda = 0.0;
    int rc = ba_reset(ctx, h);
    if (rc) return rc;
    auto download = [&](void) -> int {
        std::vector<double> pose(12 * (size_t)D.nc), K5(5);
        if (D.nc) EACHAM_HIP_TRY(ctx, hipMemcpyAsync(pose.data(), D.pose, sizeof(double) * pose.size(), hipMemcpyDeviceToHost, ctx->stream));
        if (D.nl) EACHAM_HIP_TRY(ctx, hipMemcpyAsync(R->points, D.pt, sizeof(double) * 3 * (size_t)D.nl, hipMemcpyDeviceToHost, ctx->stream));
        EACHAM_HIP_TRY(ctx, hipMemcpyAsync(K5.data(), D.Kc, sizeof(double) * 5, hipMemcpyDeviceToHost, ctx->stream));
        EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int c = 0; c < D.nc; ++c) pose_to_Twc(&pose[12 * (size_t)c], R->cam_T_wc + 16 * (size_t)c);
        R->K[0] = K5[0]; R->K[1] = K5[1]; R->K[2] = K5[3]; R->K[3] = K5[4];  // fx fy px py (:224-227)
        return EACHAM_OK;
    };
    if (h->n_landmarks_used < O->min_landmarks) {  // BundleAdjuster.cpp:166-169
        R->status = EACHAM_BA_SKIPPED;
        R->initial_error = R->final_error = NAN;
        return download();
    }
    if (O->method != EACHAM_BA_LM && O->method != EACHAM_BA_DOGLEG) return ctx->fail(EACHAM_ERR_INVALID, "unknown BA method %d", O->method);
    if (O->lm_factor_policy != EACHAM_BA_LM_FACTOR_RESET && O->lm_factor_policy != EACHAM_BA_LM_FACTOR_DOUBLE)
        return ctx->fail(EACHAM_ERR_INVALID, "unknown lm_factor_policy %d", O->lm_factor_policy);
    // use_preconditioner (BundleAdjuster.cpp:192-200) selects GTSAM's iterative solve, PCG + block-Jacobi at 1e-10, for
    // the LM steps (the option sits inside the LM branch of the reference: DogLeg never sees it)
    const bool pcg = O->method == EACHAM_BA_LM && O->use_preconditioner != 0;
    if (D.w_rows > 0 && (pcg || O->method == EACHAM_BA_DOGLEG))
        return ctx->fail(EACHAM_ERR_INVALID, "this problem was prepared for the direct Levenberg-Marquardt solve only (EACHAM_BA_SCHUR=dense)");
    long long pcg_iterations = 0;
    D.store_E = (O->method == EACHAM_BA_DOGLEG || pcg) ? 1 : 0;  // the dog-leg forms and the PCG operator read E after the linearisation

    const double lambdaUpper = 1e32, lambdaLower = 1e-16, minModelFidelity = 1e-3;
    const double relTol = (double)O->max_tolerance, absTol = (double)O->max_tolerance, errorTol = 0.0;
    const double lambdaFactor = 2.0;  // SetCeresDefaults
    double lambda = 1e-4, factor = lambdaFactor;
    int iterations = 0, inner = 0;
    double sc[3];
    launch_error(ctx, h, D.pose, D.pt, D.Kc);
    rc = read_scal(ctx, h, sc);
    if (rc) return rc;
    double error = sc[0];
    R->initial_error = error;
    double newErrorOuter = error, currentError = error;
    bool indeterminate = false;
    if (O->method == EACHAM_BA_DOGLEG) {
        // DoglegOptimizer (BundleAdjuster.cpp:204-214; GTSAM 4.1.1 DoglegOptimizer.cpp, DoglegOptimizerImpl.h):
        // per iterate() one linearisation, the Gauss-Newton step n (the LM pipeline at lambda = 0), the
        // steepest-descent point u = (g.g / g^T H g) g, then DoglegOptimizerImpl::Iterate in mode
        // ONE_STEP_PER_ITERATION. Every candidate x_d = cu u + cn n needs one retraction + error pass; the
        // model decrease M(0) - M(x_d) = g.x_d - 1/2 x_d^T H x_d comes from six scalars computed once.
        double delta = (double)O->delta;
        if (error > errorTol && iterations < O->max_iter) {
            for (;;) {  // NonlinearOptimizer::defaultOptimize
                currentError = newErrorOuter;
                launch_linearize(ctx, h);
                rc = launch_try(ctx, h, 0.0, nullptr);
                if (rc) return rc;
                EACHAM_HIP_TRY(ctx, hipMemcpyAsync(D.dl_nc, D.delta_c, sizeof(double) * (size_t)D.n, hipMemcpyDeviceToDevice, ctx->stream));
                if (D.nl) EACHAM_HIP_TRY(ctx, hipMemcpyAsync(D.dl_nl, D.delta_l, sizeof(double) * 3 * (size_t)D.nl, hipMemcpyDeviceToDevice, ctx->stream));
                {
                    ProfileScope ps(ctx, EACHAM_KERNEL_BA_ERROR);
                    ba_dl_forms_landmarks<<<D.n_lm_blocks, TPB, 0, ctx->stream>>>(D);
                    ba_dl_forms_final<<<1, TPB, 0, ctx->stream>>>(D);
                }
                double s10[10];
                EACHAM_HIP_TRY(ctx, hipMemcpyAsync(s10, D.scal, sizeof(s10), hipMemcpyDeviceToHost, ctx->stream));
                EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (((int)s10[2] & 4) != 0) return ctx->fail(EACHAM_ERR_HIP, "BA solve: an in-kernel hand-off timed out (flags %d)", (int)s10[2]);
                if (!(s10[2] == 0.0) || !std::isfinite(s10[1])) {  // GTSAM throws IndeterminantLinearSystemException here
                    indeterminate = true;
                    break;
                }
                const double gg = s10[4], gn0 = s10[5], nn = s10[6], gHg = s10[7], gHn = s10[8], nHn = s10[9];
                const double alpha = gg / gHg;  // u = alpha g
                const double uu = alpha * alpha * gg, un = alpha * gn0, uHu = alpha * alpha * gHg, uHn = alpha * gHn;
                const double gu = alpha * gg, gn = gn0;
                double f_new = error;
                bool zero_step = false;
                for (bool stay = true; stay;) {
                    double cu, cn;  // DoglegOptimizerImpl::ComputeDoglegPoint / ComputeBlend
                    const double DeltaSq = delta * delta;
                    if (DeltaSq < uu) {
                        cu = std::sqrt(DeltaSq / uu);
                        cn = 0.0;
                    } else if (DeltaSq < nn) {
                        const double a = uu - 2. * un + nn, b = 2. * (un - uu), c = uu - DeltaSq;
                        const double sq = std::sqrt(b * b - 4 * a * c);
                        const double tau1 = (-b + sq) / (2. * a), tau2 = (-b - sq) / (2. * a);
                        const double tau = (0.0 <= tau1 && tau1 <= 1.0) ? tau1 : tau2;
                        cu = 1. - tau;
                        cn = tau;
                    } else {
                        cu = 0.0;
                        cn = 1.0;
                    }
                    const double dnorm = std::sqrt(cu * cu * uu + 2 * cu * cn * un + cn * cn * nn);
                    {
                        ProfileScope ps(ctx, EACHAM_KERNEL_BA_ERROR);
                        const int na = std::max(3 * D.nl, D.n);
                        ba_dl_apply<<<(na + TPB - 1) / TPB, TPB, 0, ctx->stream>>>(D, cu * alpha, cn);
                        ba_retract_cameras<<<(D.nc + 1 + 63) / 64, 64, 0, ctx->stream>>>(D, 0.0, D.pose, D.pose_new, D.Kc, D.K_new, 1, h->err_cam, h->lin_cam);
                        launch_error_landmarks(ctx, h, D.pose_new, D.pt_new, D.K_new, 0);
                    }
                    rc = read_scal(ctx, h, sc);
                    if (rc) return rc;
                    f_new = sc[0];
                    const double decrease = (cu * gu + cn * gn) - 0.5 * (cu * cu * uHu + 2 * cu * cn * uHn + cn * cn * nHn);  // M(0) - M(x_d)
                    const double rho = (std::fabs(error - f_new) < 1e-15 || std::fabs(decrease) < 1e-15) ? 0.5 : (error - f_new) / decrease;
                    bool accepted = true;
                    const double delta_used = delta;
                    if (rho >= 0.75) {
                        delta = std::max(delta, 3.0 * dnorm);
                        stay = false;
                    } else if (rho >= 0.25) {
                        stay = false;
                    } else if (rho >= 0.0) {
                        if (delta > 1e-5) delta = 0.5 * delta;
                        stay = false;  // ONE_STEP_PER_ITERATION
                    } else {           // f increased (or NaN): shrink the region and try again
                        accepted = false;
                        if (delta > 1e-5) {
                            delta *= 0.5;
                        } else {
                            zero_step = true;  // dx_d.setZero(): the values and the error stay
                            stay = false;
                        }
                    }
                    if (R->trace && R->trace_len < R->trace_cap) {
                        eacham_ba_trace_row* tr = &R->trace[R->trace_len++];
                        tr->lambda = delta_used; tr->new_error = f_new; tr->lin_change = decrease;
                        tr->accepted = accepted ? 1 : 0; tr->outer = iterations;
                    }
                    ++inner;
                }
                if (!zero_step) {
                    std::swap(D.pose, D.pose_new);
                    std::swap(D.pt, D.pt_new);
                    std::swap(D.Kc, D.K_new);
                    error = f_new;
                }
                ++iterations;
                newErrorOuter = error;
                if (newErrorOuter <= errorTol) break;
                const double absDec = currentError - newErrorOuter, relDec = absDec / currentError;
                const bool converged = (relTol != 0.0 && relDec <= relTol) || (absDec <= absTol);
                if (!(iterations < O->max_iter) || converged || !std::isfinite(currentError)) break;
            }
        }
        lambda = delta;  // reported as final_lambda
    } else if (error > errorTol && iterations < O->max_iter) {
        for (;;) {  // NonlinearOptimizer::defaultOptimize
            currentError = newErrorOuter;
            if (D.w_rows == 0) launch_linearize(ctx, h);  // iterate(): linearize once, then tryLambda until it returns true (the dense form linearises inside the try)
            for (;;) {
                bool success = false, stop = false;
                double newError = INFINITY, linChange = NAN, fidelity = 0.0;
                rc = pcg ? launch_try_pcg(ctx, h, lambda, &pcg_iterations) : launch_try(ctx, h, lambda, nullptr);
                if (rc) return rc;
                rc = read_scal(ctx, h, sc);
                if (rc) return rc;
                // a hand-off time-out (flags bit 2) is a failure of the machine, not of the matrix: it must not
                // be taken for "not positive definite" and silently raise lambda
                if (((int)sc[2] & 4) != 0) return ctx->fail(EACHAM_ERR_HIP, "BA solve: an in-kernel hand-off timed out (flags %d)", (int)sc[2]);
                const bool solved = sc[2] == 0.0 && std::isfinite(sc[1]);
                if (solved) {
                    linChange = sc[1];
                    // oldLinearizedError = 1/2 |b|^2; the threshold eps * oldLin is far below any
                    // representable decrease here, so `> 0` decides exactly as GTSAM's test does
                    if (linChange >= 0) {
                        newError = sc[0];
                        const double cost = error - newError;
                        if (linChange > 0.0) {
                            fidelity = cost / linChange;
                            success = fidelity > minModelFidelity;
                        }
                        if (std::fabs(cost) < relTol * error) stop = true;
                    }
                }
                if (R->trace && R->trace_len < R->trace_cap) {
                    eacham_ba_trace_row* tr = &R->trace[R->trace_len++];
                    tr->lambda = lambda; tr->new_error = newError; tr->lin_change = linChange;
                    tr->accepted = success ? 1 : 0; tr->outer = iterations;
                }
                ++inner;
                if (success) {  // decreaseLambda
                    double m = 1.0 - std::pow(2.0 * fidelity - 1.0, 3);
                    if (m < 1.0 / 3.0) m = 1.0 / 3.0;
                    lambda *= m;
                    // LevenbergMarquardtState::decreaseLambda: see EACHAM_BA_LM_FACTOR_* in eacham_hip.h
                    factor = O->lm_factor_policy == EACHAM_BA_LM_FACTOR_DOUBLE ? 2.0 * factor : 2.0 * lambdaFactor;
                    if (lambda < lambdaLower) lambda = lambdaLower;
                    std::swap(D.pose, D.pose_new);
                    std::swap(D.pt, D.pt_new);
                    std::swap(D.Kc, D.K_new);
                    error = newError;
                    ++iterations;
                    break;
                } else if (!stop) {  // increaseLambda
                    lambda *= factor;
                    factor *= 2.0;
                    if (lambda >= lambdaUpper) break;
                } else {
                    break;
                }
            }
            newErrorOuter = error;
            if (newErrorOuter <= errorTol) break;
            const double absDec = currentError - newErrorOuter, relDec = absDec / currentError;
            const bool converged = (relTol != 0.0 && relDec <= relTol) || (absDec <= absTol);
            if (!(iterations < O->max_iter) || converged || !std::isfinite(currentError)) break;
        }
    }
#ifdef EXP_WD_STAMPS
    if (D.w_rows > 0) {
        unsigned long long st[16] = {0};
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(eacham::g_wd_stamp), sizeof(st));
        fprintf(stderr, "[wd stamps, us]");
        for (int i = 1; i <= 8; ++i) fprintf(stderr, " %d:%.2f", i, (double)(st[i] - st[i - 1]) * 0.01);
        fprintf(stderr, " 9:%.2f wg0-all-waves-out-of-loop-after-7:%.2f", (double)(st[9] - st[8]) * 0.01, (double)(st[14] - st[7]) * 0.01);
        fprintf(stderr, " | over the workgroups of the last launch: first start -> last at 4: %.2f, at 7: %.2f, at 8: %.2f\n", (double)(st[13] - st[10]) * 0.01,
                (double)(st[12] - st[10]) * 0.01, (double)(st[11] - st[10]) * 0.01);
        {
            unsigned long long init[16] = {0};
            init[10] = ~0ull;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(eacham::g_wd_stamp), init, sizeof(init));
        }
    }
#endif
    R->status = indeterminate ? EACHAM_BA_INDETERMINATE : EACHAM_BA_DONE;
    R->final_error = error;
    R->final_lambda = lambda;
    R->outer_iterations = iterations;
    R->inner_iterations = inner;
    R->reserved = (int32_t)std::min<long long>(pcg_iterations, 0x7fffffff);  // PCG iterations in total (0 for the direct solve)
    return download();
}

}  // namespace eacham

extern "C" {

int eacham_ba_prepare(eacham_ctx* ctx, const eacham_ba_problem* problem, eacham_ba_handle** out_handle) {
    if (!ctx || !out_handle) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    *out_handle = nullptr;
    return ba_prepare(ctx, problem, out_handle);
}

int eacham_ba_run(eacham_ctx* ctx, eacham_ba_handle* handle, const eacham_ba_options* options, eacham_ba_result* result) {
    if (!ctx || !handle) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return ba_run(ctx, handle, options, result);
}

void eacham_ba_release(eacham_ctx* ctx, eacham_ba_handle* handle) {
    if (!ctx || !handle) return;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    ba_release(ctx, handle);
}

int eacham_ba_get_plan_info(eacham_ctx* ctx, const eacham_ba_handle* handle, eacham_ba_plan_info* out) {
    if (!ctx || !handle || !out) return EACHAM_ERR_INVALID;
    const BaPlan& pl = handle->plan;
    out->n_panels = pl.npan; out->n_tiles = pl.ntiles; out->n_levels = pl.n_levels;
    out->ordering = pl.ordering; out->nd_leaf = pl.nd_leaf; out->reserved = 0;
    out->tile_updates = pl.tile_updates;
    out->est_us = pl.est_us;
    out->prepare_us[0] = handle->prep_us[0];
    out->prepare_us[1] = handle->prep_us[1];
    out->prepare_us[2] = handle->prep_us[2];
    return EACHAM_OK;
}

int eacham_ba_debug_structure(eacham_ctx* ctx, const eacham_ba_handle* h, int which, void* out, int64_t cap_bytes, int64_t* out_bytes) {
    if (!ctx || !h || !out_bytes) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    const BaDev& D = h->D;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0: src = D.lm_ptr; bytes = sizeof(int) * ((size_t)D.nl + 1); break;
        case 1: src = D.cam_ptr; bytes = sizeof(int) * ((size_t)D.nc + 1); break;
        case 2: src = D.cam_obs; bytes = sizeof(int) * (size_t)D.no; break;
        case 3: src = D.obs_pos; bytes = sizeof(int) * (size_t)D.no; break;
        case 4: src = D.obs_cam; bytes = sizeof(unsigned) * (size_t)D.no; break;
        case 5: src = D.obs_lm; bytes = sizeof(unsigned) * (size_t)D.no; break;
        case 6: src = D.obs_uv; bytes = sizeof(double) * 2 * (size_t)D.no; break;
        case 7: src = D.cam_uv; bytes = sizeof(double) * 2 * (size_t)D.no; break;
        case 8: src = D.cam_lm; bytes = sizeof(int) * (size_t)D.no; break;
        case 9: src = D.pos_cam; bytes = sizeof(int) * (size_t)D.no; break;
        case 10: src = D.cam_chunks; bytes = sizeof(int2) * (size_t)D.n_cam_chunks; break;
        case 11: src = D.cam_chunk_ptr; bytes = sizeof(int) * ((size_t)D.nc + 1); break;
        case 12: src = D.blocks; bytes = sizeof(int4) * (size_t)D.n_blocks; break;
        case 13: src = D.pair_chunks; bytes = sizeof(int4) * (size_t)D.n_chunks; break;
        case 14: {
            long long n = 0;  // entries = first entry + count of the last chunk
            if (D.n_chunks > 0) {
                int4 last;
                EACHAM_HIP_TRY(ctx, hipMemcpy(&last, D.pair_chunks + (D.n_chunks - 1), sizeof(int4), hipMemcpyDeviceToHost));
                n = (long long)last.y + last.z;
            }
            src = D.pair_entries; bytes = sizeof(int2) * (size_t)n;
            break;
        }
        case 15: src = D.pose0; bytes = sizeof(double) * 12 * (size_t)D.nc; break;
        case 16: src = D.pt0; bytes = sizeof(double) * 3 * (size_t)D.nl; break;
        case 17: src = D.lmprior; bytes = sizeof(double) * 2 * (size_t)D.nl; break;
        case 18: src = D.K0; bytes = sizeof(double) * 5; break;
        case 19: src = D.fixed; bytes = sizeof(int) * (size_t)D.nc; break;
        // the landmark-major structure of the Schur stage (ba_groups.hpp); all empty when the pair lists serve
        case 20: src = D.g_groups; bytes = D.g_rows ? sizeof(BaGroup) * (size_t)D.g_ngroups : 0; break;
        case 21: src = D.g_lmid; bytes = D.g_rows ? sizeof(int) * (size_t)D.g_ngroups * (D.g_rows / 4) : 0; break;
        case 22: src = D.g_lmrow; bytes = D.g_rows ? sizeof(int) * (size_t)D.g_ngroups * (D.g_rows / 4) : 0; break;
        case 23: src = D.g_rowinfo; bytes = D.g_rows ? sizeof(int2) * (size_t)D.g_ngroups * D.g_rows : 0; break;
        case 24: src = D.g_uv; bytes = D.g_rows ? sizeof(double) * 2 * (size_t)D.g_ngroups * D.g_rows : 0; break;
        case 25: src = D.g_ent; bytes = D.g_rows ? sizeof(uint32_t) * 256 * (size_t)D.g_nent4 : 0; break;
        case 26: src = D.g_chunks; bytes = D.g_rows ? sizeof(BaChunk) * (size_t)D.g_nchunks : 0; break;
        case 27: src = D.g_laneinfo; bytes = D.g_rows ? sizeof(uint32_t) * 64 * (size_t)D.g_nchunks : 0; break;
        case 28: src = D.g_blk; bytes = D.g_rows ? sizeof(int4) * (size_t)D.g_nblk : 0; break;
        case 29: src = D.g_longblk; bytes = D.g_rows ? sizeof(int) * (size_t)D.g_nlong : 0; break;
        // the dense form for a local window (ba_window.hpp); empty when another form serves
        case 30: src = D.w_groups; bytes = D.w_rows ? sizeof(int2) * (size_t)D.w_ngroups : 0; break;
        case 31: src = D.w_lmid; bytes = D.w_rows ? sizeof(int) * (size_t)D.w_ngroups * (D.w_rows / 4) : 0; break;
        case 32: src = D.w_lmrow; bytes = D.w_rows ? sizeof(int) * (size_t)D.w_ngroups * (D.w_rows / 4) : 0; break;
        case 33: src = D.w_rowinfo; bytes = D.w_rows ? sizeof(int2) * (size_t)D.w_ngroups * D.w_rows : 0; break;
        case 34: src = D.w_uv; bytes = D.w_rows ? sizeof(double) * 2 * (size_t)D.w_ngroups * D.w_rows : 0; break;
        default: return ctx->fail(EACHAM_ERR_INVALID, "unknown structure array %d", which);
    }
    *out_bytes = (int64_t)bytes;
    if ((int64_t)bytes > cap_bytes) return ctx->fail(EACHAM_ERR_CAPACITY, "structure array %d needs %zu bytes", which, bytes);
    EACHAM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (bytes) EACHAM_HIP_TRY(ctx, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    return EACHAM_OK;
}

int eacham_ba_solve(eacham_ctx* ctx, const eacham_ba_problem* problem, const eacham_ba_options* options, eacham_ba_result* result) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    eacham_ba_handle* h = nullptr;
    const bool lm_direct = options && options->method == EACHAM_BA_LM && options->use_preconditioner == 0;
    int rc = ba_prepare(ctx, problem, &h, lm_direct);
    if (rc) return rc;
    rc = ba_run(ctx, h, options, result);
    ba_release(ctx, h);
    return rc;
}

int eacham_ba_debug_step(eacham_ctx* ctx, const eacham_ba_problem* problem, double lambda, double* S, double* g,
                         double* delta_cams, double* delta_points, double* error, double* lin_change) {
    if (!ctx) return EACHAM_ERR_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    eacham_ba_handle* h = nullptr;
    int rc = ba_prepare(ctx, problem, &h);
    if (rc) return rc;
    BaDev& D = h->D;
    const int n = D.n;
    double sc[3];
    std::vector<double> Sg((size_t)(n + 1) * n);
    rc = ba_reset(ctx, h);
    if (!rc) {
        launch_error(ctx, h, D.pose, D.pt, D.Kc);
        rc = read_scal(ctx, h, sc);
    }
    if (!rc && error) *error = sc[0];
    if (!rc) {
        launch_linearize(ctx, h);
        rc = launch_try(ctx, h, lambda, Sg.data());
    }
    if (!rc) rc = read_scal(ctx, h, sc);
    if (!rc) {
        if (lin_change) *lin_change = sc[1];
        if (S) {
            memcpy(S, Sg.data(), sizeof(double) * (size_t)n * n);
        }
        if (g) memcpy(g, Sg.data() + (size_t)n * n, sizeof(double) * (size_t)n);
        hipError_t e1 = delta_cams ? hipMemcpy(delta_cams, D.delta_c, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) : hipSuccess;
        hipError_t e2 = (delta_points && D.nl) ? hipMemcpy(delta_points, D.delta_l, sizeof(double) * 3 * (size_t)D.nl, hipMemcpyDeviceToHost) : hipSuccess;
        if (e1 != hipSuccess || e2 != hipSuccess) rc = ctx->fail(EACHAM_ERR_HIP, "debug_step download failed");
        if (!rc && ((int)sc[2] & 4) != 0) rc = ctx->fail(EACHAM_ERR_HIP, "BA solve: an in-kernel hand-off timed out (flags %d)", (int)sc[2]);
        if (!rc && sc[2] != 0.0) rc = ctx->fail(EACHAM_ERR_INVALID, "reduced system not positive definite (flags %d)", (int)sc[2]);
    }
    ba_release(ctx, h);
    return rc;
}

}  // extern "C"
