/*
 * ba_oracle.c — CPU restatement of eacham's bundle-adjustment hot path (RefineBA).
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (eacham_amd/, include/) may call this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * PARITY UNPINNED: the reference has no tests/golden vectors for this path and
 * its arithmetic lives in GTSAM 4.1.1 (conanfile.txt:2), absent here and unbuildable (needs Boost,
 * Eigen, TBB). This file restates the published algorithm of the call sites in
 *   /root/reference/modules/sfm/reconstruction/BundleAdjuster.cpp:28-250
 * with GTSAM 4.1.1 semantics as recorded in SURVEY.md Appendix A (from memory):
 *   - GeneralSFMFactor2<Cal3_S2>::evaluateError (:95-98)        -> reproj_eval()
 *   - noise models + Huber IRLS (:28-33,:60-63,:71,:89-91,:110-113,:173-175) -> robust_*()
 *   - PriorFactor<Pose3|Point3|Cal3_S2> (:72,:76,:114,:177)       e = -Local(x, prior), H = I
 *   - LevenbergMarquardtParams::SetCeresDefaults + overrides (:184-190) -> lm_*()
 *   - LevenbergMarquardtOptimizer::optimize (:216)                -> oracle_ba_solve()
 * Chosen build options of GTSAM (conan recipe defaults of 4.1.1): GTSAM_POSE3_EXPMAP=OFF,
 * GTSAM_ROT3_EXPMAP=OFF -> first-order Pose3 chart with the Cayley map on Rot3.
 * The linear solve is a landmark-first Schur complement + dense Cholesky, mathematically the
 * multifrontal Cholesky GTSAM runs (same delta up to rounding); a full dense solve of all
 * variables (mode 1) cross-checks it in the tests.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/eacham_hip.h"

/* ---- noise parameters, with the reference's float arithmetic (BundleAdjuster.cpp:28-33) -------- */
static double rot_sigma(float deg) { /* CreateNoise6_2_1: rot * 3.141592f / 180.0f */
    const float r = deg * 3.141592f / 180.0f;
    return (double)r;
}
static const float POSE_POS_SIGMA = 0.35f, POSE_ROT_DEG = 45.0f, POSE_HUBER = 2.5f;  /* :60-63 */
static const float FIXED_SIGMA = 0.0001f;                                             /* :71 */
static const float PIX_SIGMA = 1.5f, PIX_HUBER = 3.0f;                                 /* :89-91 */
static const double K_SIGMA[5] = {25, 25, 0.00001, 0.0001, 0.0001};                   /* :173-175 */

/* mEstimator::Huber: weight and loss of the whitened residual norm */
static inline double huber_weight(double n, double k) { return n <= k ? 1.0 : k / n; }
static inline double huber_loss(double n, double k) { return n <= k ? 0.5 * n * n : k * (n - 0.5 * k); }

typedef struct {
    double R[9]; /* camera->world rotation, row-major */
    double t[3]; /* camera centre in world */
} pose_t;

typedef struct {
    int nc, nl, no;
    pose_t* pose;       /* current values */
    double* pt;         /* nl x 3 */
    double K[5];        /* fx fy s u0 v0 */
    pose_t* pose0;      /* prior means */
    double* pt0;
    double K0[5];
} state_t;

/* ---- small linear algebra ------------------------------------------------------------------- */
static void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void mat3_tmul(const double* A, const double* B, double* C) { /* A^T B */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

/* Rot3 Cayley chart: R = ((4 - |w|^2) I + 2 w w^T + 4 [w]x) / (4 + |w|^2) */
static void cayley(const double* w, double* R) {
    const double x = w[0], y = w[1], z = w[2];
    const double x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
    const double f = 1.0 / (4.0 + x2 + y2 + z2), f2 = 2.0 * f;
    R[0] = (4 + x2 - y2 - z2) * f; R[1] = (xy - 2 * z) * f2;      R[2] = (xz + 2 * y) * f2;
    R[3] = (xy + 2 * z) * f2;      R[4] = (4 - x2 + y2 - z2) * f; R[5] = (yz - 2 * x) * f2;
    R[6] = (xz - 2 * y) * f2;      R[7] = (yz + 2 * x) * f2;      R[8] = (4 - x2 - y2 + z2) * f;
}
/* inverse chart: w = 2 vee(R - R^T) / (1 + tr R) */
static void cayley_local(const double* R, double* w) {
    const double s = 2.0 / (1.0 + R[0] + R[4] + R[8]);
    w[0] = s * (R[7] - R[5]);
    w[1] = s * (R[2] - R[6]);
    w[2] = s * (R[3] - R[1]);
}

/* Pose3(position) with position = Node::transform^-1 (BundleAdjuster.cpp:65-67): rigid inverse */
static void pose_from_Twc(const double* T, pose_t* x) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) x->R[3 * i + j] = T[4 * j + i];
    for (int i = 0; i < 3; ++i) x->t[i] = -(x->R[3 * i] * T[3] + x->R[3 * i + 1] * T[7] + x->R[3 * i + 2] * T[11]);
}
/* result.inverse() (BundleAdjuster.cpp:242-247) */
static void pose_to_Twc(const pose_t* x, double* T) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = x->R[3 * j + i];
        T[4 * i + 3] = -(x->R[i] * x->t[0] + x->R[3 + i] * x->t[1] + x->R[6 + i] * x->t[2]);
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}
/* Pose3 retract, first-order chart: (R Cayley(w), t + R v) */
static void pose_retract(const pose_t* x, const double* d, pose_t* out) {
    double C[9];
    cayley(d, C);
    mat3_mul(x->R, C, out->R);
    for (int i = 0; i < 3; ++i)
        out->t[i] = x->t[i] + x->R[3 * i] * d[3] + x->R[3 * i + 1] * d[4] + x->R[3 * i + 2] * d[5];
}
/* Local(x, prior) = chart coordinates of x^-1 * prior */
static void pose_local(const pose_t* x, const pose_t* prior, double* xi) {
    double Rd[9], dt[3];
    mat3_tmul(x->R, prior->R, Rd);
    cayley_local(Rd, xi);
    for (int i = 0; i < 3; ++i) dt[i] = prior->t[i] - x->t[i];
    for (int i = 0; i < 3; ++i) xi[3 + i] = x->R[i] * dt[0] + x->R[3 + i] * dt[1] + x->R[6 + i] * dt[2];
}

/* ---- GeneralSFMFactor2<Cal3_S2>::evaluateError ---------------------- */
/* r = project(pose, point, K) - uv; Jp 2x6 ([w, v] order), Jl 2x3, Jk 2x5. Returns 0 on a
 * cheirality failure (z <= 0): residual and Jacobians are zero, as GTSAM's catch block sets them. */
static int reproj_eval(const pose_t* x, const double* l, const double* K, const double* uv, double* r,
                       double* Jp, double* Jl, double* Jk) {
    const double dx = l[0] - x->t[0], dy = l[1] - x->t[1], dz = l[2] - x->t[2];
    const double qx = x->R[0] * dx + x->R[3] * dy + x->R[6] * dz; /* R^T (l - t) */
    const double qy = x->R[1] * dx + x->R[4] * dy + x->R[7] * dz;
    const double qz = x->R[2] * dx + x->R[5] * dy + x->R[8] * dz;
    if (qz <= 0.0) {
        r[0] = r[1] = 0.0;
        if (Jp) memset(Jp, 0, 12 * sizeof(double));
        if (Jl) memset(Jl, 0, 6 * sizeof(double));
        if (Jk) memset(Jk, 0, 10 * sizeof(double));
        return 0;
    }
    const double d = 1.0 / qz, u = qx * d, v = qy * d;
    const double fx = K[0], fy = K[1], s = K[2];
    r[0] = fx * u + s * v + K[3] - uv[0];
    r[1] = fy * v + K[4] - uv[1];
    if (Jp) {
        const double Dn[12] = {u * v, -1 - u * u, v, -d, 0, d * u, 1 + v * v, -u * v, -u, 0, -d, d * v};
        for (int j = 0; j < 6; ++j) {
            Jp[j] = fx * Dn[j] + s * Dn[6 + j];
            Jp[6 + j] = fy * Dn[6 + j];
        }
    }
    if (Jl) {
        for (int j = 0; j < 3; ++j) { /* d * [Rt.row(0) - u Rt.row(2); Rt.row(1) - v Rt.row(2)], Rt = R^T */
            const double a0 = d * (x->R[3 * j] - u * x->R[3 * j + 2]);
            const double a1 = d * (x->R[3 * j + 1] - v * x->R[3 * j + 2]);
            Jl[j] = fx * a0 + s * a1;
            Jl[3 + j] = fy * a1;
        }
    }
    if (Jk) {
        const double Dk[10] = {u, 0, v, 1, 0, 0, v, 0, 0, 1};
        memcpy(Jk, Dk, sizeof(Dk));
    }
    return 1;
}

/* raw factor evaluation for the Jacobian tests: pose given as world->camera 4x4 */
int oracle_ba_project(const double* T_wc, const double* point, const double* K5, const double* uv,
                      double* r, double* Jp, double* Jl, double* Jk) {
    pose_t x;
    pose_from_Twc(T_wc, &x);
    return reproj_eval(&x, point, K5, uv, r, Jp, Jl, Jk);
}
/* chart helpers for the tests: out = world->camera of retract(pose(T_wc), xi) */
void oracle_ba_pose_retract(const double* T_wc, const double* xi, double* T_out) {
    pose_t x, y;
    pose_from_Twc(T_wc, &x);
    pose_retract(&x, xi, &y);
    pose_to_Twc(&y, T_out);
}
void oracle_ba_pose_local(const double* T_wc, const double* T_prior_wc, double* xi) {
    pose_t x, p;
    pose_from_Twc(T_wc, &x);
    pose_from_Twc(T_prior_wc, &p);
    pose_local(&x, &p, xi);
}

/* ---- problem setup ------------------------------------------------------------------------------ */
typedef struct {
    const eacham_ba_problem* P;
    state_t S;
    double pose_sigma[6], fixed_sigma[6];
    int* lm_ptr; /* CSR of observations by landmark */
    int* lm_obs;
    int n_landmarks_used; /* landmarks with >= 1 observation in the problem (mapIds.size()) */
} ba_t;

static void ba_free(ba_t* B) {
    free(B->S.pose); free(B->S.pose0); free(B->S.pt); free(B->S.pt0); free(B->lm_ptr); free(B->lm_obs);
}

static int ba_init(ba_t* B, const eacham_ba_problem* P) {
    memset(B, 0, sizeof(*B));
    B->P = P;
    state_t* S = &B->S;
    S->nc = P->n_cams; S->nl = P->n_points; S->no = P->n_obs;
    S->pose = (pose_t*)malloc(sizeof(pose_t) * (size_t)(S->nc > 0 ? S->nc : 1));
    S->pose0 = (pose_t*)malloc(sizeof(pose_t) * (size_t)(S->nc > 0 ? S->nc : 1));
    S->pt = (double*)malloc(sizeof(double) * 3 * (size_t)(S->nl > 0 ? S->nl : 1));
    S->pt0 = (double*)malloc(sizeof(double) * 3 * (size_t)(S->nl > 0 ? S->nl : 1));
    for (int i = 0; i < S->nc; ++i) {
        pose_from_Twc(P->cam_T_wc + 16 * (size_t)i, &S->pose[i]);
        S->pose0[i] = S->pose[i];
    }
    memcpy(S->pt, P->points, sizeof(double) * 3 * (size_t)S->nl);
    memcpy(S->pt0, P->points, sizeof(double) * 3 * (size_t)S->nl);
    S->K[0] = P->K[0]; S->K[1] = P->K[1]; S->K[2] = 0.0; S->K[3] = P->K[2]; S->K[4] = P->K[3]; /* :47-49 */
    memcpy(S->K0, S->K, sizeof(S->K));
    for (int k = 0; k < 3; ++k) {
        B->pose_sigma[k] = rot_sigma(POSE_ROT_DEG);
        B->pose_sigma[3 + k] = (double)POSE_POS_SIGMA;
        B->fixed_sigma[k] = rot_sigma(FIXED_SIGMA);
        B->fixed_sigma[3 + k] = (double)FIXED_SIGMA;
    }
    B->lm_ptr = (int*)calloc((size_t)S->nl + 2, sizeof(int));
    B->lm_obs = (int*)malloc(sizeof(int) * (size_t)(S->no > 0 ? S->no : 1));
    for (int o = 0; o < S->no; ++o) {
        if (P->obs_point[o] >= (uint32_t)S->nl || P->obs_cam[o] >= (uint32_t)S->nc) return -1;
        B->lm_ptr[P->obs_point[o] + 1]++;
    }
    for (int j = 0; j < S->nl; ++j) {
        if (B->lm_ptr[j + 1] > 0) B->n_landmarks_used++;
        B->lm_ptr[j + 1] += B->lm_ptr[j];
    }
    int* fill = (int*)malloc(sizeof(int) * ((size_t)S->nl + 1));
    memcpy(fill, B->lm_ptr, sizeof(int) * ((size_t)S->nl + 1));
    for (int o = 0; o < S->no; ++o) B->lm_obs[fill[P->obs_point[o]]++] = o;
    free(fill);
    return 0;
}

/* landmark prior noise (BundleAdjuster.cpp:109-113): sigma = 1.0f/obs, Huber k = 3.0f/obs, float */
static inline void lm_prior_params(int observers, double* sigma, double* k) {
    const float o = (float)(observers > 0 ? observers : 1);
    *sigma = (double)(1.0f / o);
    *k = (double)(3.0f / o);
}

/* Sums over the observations run in fixed blocks: a block is added up in order by one thread, the block sums are added in block
 * order afterwards — the result does not depend on the number of threads or on their timing (an OpenMP `reduction` combines the
 * threads' partial sums in whatever order they finish: the last bits of graph.error then varied from run to run, which a test on
 * a marginal accept / reject decision of LM saw as a rare failure). */
#define OBS_BLOCK 4096

/* ---- nonlinear error: graph.error(values) (Appendix A.3) ---------------------------------------- */
static double graph_error(const ba_t* B, const state_t* S) {
    const eacham_ba_problem* P = B->P;
    double err = 0.0;
    const double sig = (double)PIX_SIGMA, kh = (double)PIX_HUBER;
    {
        const int nb = (S->no + OBS_BLOCK - 1) / OBS_BLOCK;
        double* part = (double*)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
#pragma omp parallel for schedule(static)
        for (int blk = 0; blk < nb; ++blk) {
            double e = 0.0;
            const int o1 = (blk + 1) * OBS_BLOCK < S->no ? (blk + 1) * OBS_BLOCK : S->no;
            for (int o = blk * OBS_BLOCK; o < o1; ++o) {
                double r[2];
                reproj_eval(&S->pose[P->obs_cam[o]], S->pt + 3 * (size_t)P->obs_point[o], S->K, P->obs_uv + 2 * (size_t)o, r, 0, 0, 0);
                const double n = sqrt(r[0] * r[0] + r[1] * r[1]) / sig;
                e += huber_loss(n, kh);
            }
            part[blk] = e;
        }
        for (int blk = 0; blk < nb; ++blk) err += part[blk];
        free(part);
    }
    for (int i = 0; i < S->nc; ++i) {
        double xi[6], n2 = 0.0;
        pose_local(&S->pose[i], &S->pose0[i], xi);
        const double* sg = P->cam_fixed[i] ? B->fixed_sigma : B->pose_sigma;
        for (int k = 0; k < 6; ++k) n2 += (xi[k] / sg[k]) * (xi[k] / sg[k]);
        err += P->cam_fixed[i] ? 0.5 * n2 : huber_loss(sqrt(n2), (double)POSE_HUBER);
    }
    for (int j = 0; j < S->nl; ++j) {
        if (B->lm_ptr[j + 1] == B->lm_ptr[j]) continue; /* not part of the graph */
        double sg, k, n2 = 0.0;
        lm_prior_params(P->point_observers[j], &sg, &k);
        for (int a = 0; a < 3; ++a) {
            const double e = (S->pt[3 * j + a] - S->pt0[3 * j + a]) / sg;
            n2 += e * e;
        }
        err += huber_loss(sqrt(n2), k);
    }
    for (int a = 0; a < 5; ++a) {
        const double e = (S->K[a] - S->K0[a]) / K_SIGMA[a];
        err += 0.5 * e * e;
    }
    return err;
}

/* ---- linearisation: whitened, robust-reweighted Jacobian factors --------------------------------- */
typedef struct {
    double* Ap; /* no x 12 */
    double* Al; /* no x 6  */
    double* Ak; /* no x 10 */
    double* b;  /* no x 2  */
    double* Pw; /* nc x 6: diagonal of the pose-prior A (sqrt(w)/sigma) */
    double* Pb; /* nc x 6: pose-prior b */
    double* Lw; /* nl: landmark-prior A scale */
    double* Lb; /* nl x 3 */
    double Kw[5], Kb[5];
} lin_t;

static void lin_alloc(lin_t* L, const state_t* S) {
    const size_t no = (size_t)(S->no > 0 ? S->no : 1), nc = (size_t)(S->nc > 0 ? S->nc : 1), nl = (size_t)(S->nl > 0 ? S->nl : 1);
    L->Ap = (double*)malloc(sizeof(double) * 12 * no);
    L->Al = (double*)malloc(sizeof(double) * 6 * no);
    L->Ak = (double*)malloc(sizeof(double) * 10 * no);
    L->b = (double*)malloc(sizeof(double) * 2 * no);
    L->Pw = (double*)malloc(sizeof(double) * 6 * nc);
    L->Pb = (double*)malloc(sizeof(double) * 6 * nc);
    L->Lw = (double*)malloc(sizeof(double) * nl);
    L->Lb = (double*)malloc(sizeof(double) * 3 * nl);
}
static void lin_free(lin_t* L) {
    free(L->Ap); free(L->Al); free(L->Ak); free(L->b); free(L->Pw); free(L->Pb); free(L->Lw); free(L->Lb);
}

static void linearize(const ba_t* B, const state_t* S, lin_t* L) {
    const eacham_ba_problem* P = B->P;
    const double sig = (double)PIX_SIGMA, kh = (double)PIX_HUBER;
#pragma omp parallel for schedule(static)
    for (int o = 0; o < S->no; ++o) {
        double r[2], Jp[12], Jl[6], Jk[10];
        reproj_eval(&S->pose[P->obs_cam[o]], S->pt + 3 * (size_t)P->obs_point[o], S->K, P->obs_uv + 2 * (size_t)o, r, Jp, Jl, Jk);
        const double e0 = r[0] / sig, e1 = r[1] / sig;
        const double sw = sqrt(huber_weight(sqrt(e0 * e0 + e1 * e1), kh)), sc = sw / sig; /* Robust::WhitenSystem */
        for (int k = 0; k < 12; ++k) L->Ap[12 * (size_t)o + k] = sc * Jp[k];
        for (int k = 0; k < 6; ++k) L->Al[6 * (size_t)o + k] = sc * Jl[k];
        for (int k = 0; k < 10; ++k) L->Ak[10 * (size_t)o + k] = sc * Jk[k];
        L->b[2 * (size_t)o] = -sw * e0;
        L->b[2 * (size_t)o + 1] = -sw * e1;
    }
    for (int i = 0; i < S->nc; ++i) { /* PriorFactor<Pose3>: e = -Local(x, prior), H = I */
        double xi[6], e[6], n2 = 0.0;
        pose_local(&S->pose[i], &S->pose0[i], xi);
        const double* sg = P->cam_fixed[i] ? B->fixed_sigma : B->pose_sigma;
        for (int k = 0; k < 6; ++k) {
            e[k] = -xi[k] / sg[k];
            n2 += e[k] * e[k];
        }
        const double sw = P->cam_fixed[i] ? 1.0 : sqrt(huber_weight(sqrt(n2), (double)POSE_HUBER));
        for (int k = 0; k < 6; ++k) {
            L->Pw[6 * i + k] = sw / sg[k];
            L->Pb[6 * i + k] = -sw * e[k];
        }
    }
    for (int j = 0; j < S->nl; ++j) { /* PriorFactor<Point3> */
        double sg, k, e[3], n2 = 0.0;
        lm_prior_params(P->point_observers[j], &sg, &k);
        for (int a = 0; a < 3; ++a) {
            e[a] = (S->pt[3 * j + a] - S->pt0[3 * j + a]) / sg;
            n2 += e[a] * e[a];
        }
        const double sw = sqrt(huber_weight(sqrt(n2), k));
        L->Lw[j] = sw / sg;
        for (int a = 0; a < 3; ++a) L->Lb[3 * j + a] = -sw * e[a];
    }
    for (int a = 0; a < 5; ++a) { /* PriorFactor<Cal3_S2>, Gaussian */
        L->Kw[a] = 1.0 / K_SIGMA[a];
        L->Kb[a] = -(S->K[a] - S->K0[a]) / K_SIGMA[a];
    }
}

/* ---- dense SPD solve (blocked right-looking Cholesky), returns 0 if not positive definite ------- */
static int cholesky_solve(double* A, double* x, int n) { /* A row-major, lower triangle used; x = rhs -> solution */
    const int NB = 48;
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = (k0 + NB < n) ? NB : n - k0, k1 = k0 + kb;
        for (int j = k0; j < k1; ++j) { /* unblocked factor of the diagonal block */
            double d = A[(size_t)j * n + j];
            for (int k = k0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
            if (!(d > 0.0) || !isfinite(d)) return 0;
            d = sqrt(d);
            A[(size_t)j * n + j] = d;
            for (int i = j + 1; i < k1; ++i) {
                double s = A[(size_t)i * n + j];
                for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
                A[(size_t)i * n + j] = s / d;
            }
        }
#pragma omp parallel for schedule(static)
        for (int i = k1; i < n; ++i) /* panel: L21 = A21 L11^-T */
            for (int j = k0; j < k1; ++j) {
                double s = A[(size_t)i * n + j];
                for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
                A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
            }
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = k1; i < n; ++i) /* trailing update A22 -= L21 L21^T (lower part) */
            for (int j = k1; j <= i; ++j) {
                double s = 0.0;
                for (int k = k0; k < k1; ++k) s += A[(size_t)i * n + k] * A[(size_t)j * n + k];
                A[(size_t)i * n + j] -= s;
            }
    }
    for (int i = 0; i < n; ++i) { /* forward */
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * x[k];
        x[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) { /* backward */
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k];
        x[i] = s / A[(size_t)i * n + i];
    }
    return 1;
}

static int inv3_spd(const double* H, double* V) { /* inverse of a symmetric 3x3 via cofactors */
    const double a = H[0], b = H[1], c = H[2], d = H[4], e = H[5], f = H[8];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    if (!(det > 0.0) || !isfinite(det)) return 0;
    const double id = 1.0 / det;
    V[0] = c00 * id; V[1] = V[3] = c01 * id; V[2] = V[6] = c02 * id;
    V[4] = (a * f - c * c) * id; V[5] = V[7] = (b * c - a * e) * id; V[8] = (a * d - b * b) * id;
    return 1;
}

static inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/*
 * One damped step: solves (H + lambda * diag(clamp(diag H, 1e-6, 1e32))) delta = g.
 * dc: 6*nc + 5 (cameras, then K), dl: 3*nl. mode 0 = Schur complement, 1 = dense over all variables.
 * Sout/gout (optional, mode 0): reduced system before factorisation (full symmetric, row-major).
 * Returns 0 when the system is not positive definite (IndeterminantLinearSystemException).
 */
static int solve_step(const ba_t* B, const lin_t* L, double lambda, int mode, double* dc, double* dl, double* Sout, double* gout) {
    const eacham_ba_problem* P = B->P;
    const state_t* S = &B->S;
    const int nc = S->nc, nl = S->nl, n = 6 * nc + 5;
    int ok = 1;
    /* per-variable Hessian blocks of the undamped system */
    double* Hcc = (double*)calloc((size_t)36 * (nc > 0 ? nc : 1), sizeof(double));
    double* HcK = (double*)calloc((size_t)30 * (nc > 0 ? nc : 1), sizeof(double));
    double* gc = (double*)calloc((size_t)6 * (nc > 0 ? nc : 1), sizeof(double));
    double* Hll = (double*)calloc((size_t)9 * (nl > 0 ? nl : 1), sizeof(double));
    double* gl = (double*)calloc((size_t)3 * (nl > 0 ? nl : 1), sizeof(double));
    double* ElK = (double*)calloc((size_t)15 * (nl > 0 ? nl : 1), sizeof(double)); /* sum Ak^T Al, 5x3 */
    double HKK[25] = {0}, gK[5] = {0};
    for (int o = 0; o < S->no; ++o) {
        const int c = (int)P->obs_cam[o], j = (int)P->obs_point[o];
        const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o, *Ak = L->Ak + 10 * (size_t)o, *b = L->b + 2 * (size_t)o;
        for (int a = 0; a < 6; ++a) {
            for (int bb = 0; bb < 6; ++bb) Hcc[36 * c + 6 * a + bb] += Ap[a] * Ap[bb] + Ap[6 + a] * Ap[6 + bb];
            for (int bb = 0; bb < 5; ++bb) HcK[30 * c + 5 * a + bb] += Ap[a] * Ak[bb] + Ap[6 + a] * Ak[5 + bb];
            gc[6 * c + a] += Ap[a] * b[0] + Ap[6 + a] * b[1];
        }
        for (int a = 0; a < 3; ++a) {
            for (int bb = 0; bb < 3; ++bb) Hll[9 * j + 3 * a + bb] += Al[a] * Al[bb] + Al[3 + a] * Al[3 + bb];
            gl[3 * j + a] += Al[a] * b[0] + Al[3 + a] * b[1];
        }
        for (int a = 0; a < 5; ++a) {
            for (int bb = 0; bb < 5; ++bb) HKK[5 * a + bb] += Ak[a] * Ak[bb] + Ak[5 + a] * Ak[5 + bb];
            for (int bb = 0; bb < 3; ++bb) ElK[15 * j + 3 * a + bb] += Ak[a] * Al[bb] + Ak[5 + a] * Al[3 + bb];
            gK[a] += Ak[a] * b[0] + Ak[5 + a] * b[1];
        }
    }
    for (int i = 0; i < nc; ++i)
        for (int a = 0; a < 6; ++a) {
            Hcc[36 * i + 7 * a] += L->Pw[6 * i + a] * L->Pw[6 * i + a];
            gc[6 * i + a] += L->Pw[6 * i + a] * L->Pb[6 * i + a];
        }
    for (int j = 0; j < nl; ++j) {
        if (B->lm_ptr[j + 1] == B->lm_ptr[j]) continue;
        for (int a = 0; a < 3; ++a) {
            Hll[9 * j + 4 * a] += L->Lw[j] * L->Lw[j];
            gl[3 * j + a] += L->Lw[j] * L->Lb[3 * j + a];
        }
    }
    for (int a = 0; a < 5; ++a) {
        HKK[6 * a] += L->Kw[a] * L->Kw[a];
        gK[a] += L->Kw[a] * L->Kb[a];
    }
    /* diagonal damping: lambda * clamp(diag(H)) (minDiagonal 1e-6, maxDiagonal 1e32) */
    for (int i = 0; i < nc; ++i)
        for (int a = 0; a < 6; ++a) Hcc[36 * i + 7 * a] += lambda * clampd(Hcc[36 * i + 7 * a], 1e-6, 1e32);
    for (int j = 0; j < nl; ++j)
        for (int a = 0; a < 3; ++a) Hll[9 * j + 4 * a] += lambda * clampd(Hll[9 * j + 4 * a], 1e-6, 1e32);
    for (int a = 0; a < 5; ++a) HKK[6 * a] += lambda * clampd(HKK[6 * a], 1e-6, 1e32);

    if (mode == 1) { /* dense system over [cams, K, points] */
        const int N = n + 3 * nl;
        double* A = (double*)calloc((size_t)N * N, sizeof(double));
        double* x = (double*)calloc((size_t)N, sizeof(double));
        for (int i = 0; i < nc; ++i) {
            for (int a = 0; a < 6; ++a) {
                for (int bb = 0; bb < 6; ++bb) A[(size_t)(6 * i + a) * N + 6 * i + bb] = Hcc[36 * i + 6 * a + bb];
                for (int bb = 0; bb < 5; ++bb) {
                    A[(size_t)(6 * i + a) * N + 6 * nc + bb] = HcK[30 * i + 5 * a + bb];
                    A[(size_t)(6 * nc + bb) * N + 6 * i + a] = HcK[30 * i + 5 * a + bb];
                }
                x[6 * i + a] = gc[6 * i + a];
            }
        }
        for (int a = 0; a < 5; ++a) {
            for (int bb = 0; bb < 5; ++bb) A[(size_t)(6 * nc + a) * N + 6 * nc + bb] = HKK[5 * a + bb];
            x[6 * nc + a] = gK[a];
        }
        for (int j = 0; j < nl; ++j)
            for (int a = 0; a < 3; ++a) {
                for (int bb = 0; bb < 3; ++bb) A[(size_t)(n + 3 * j + a) * N + n + 3 * j + bb] = Hll[9 * j + 3 * a + bb];
                if (B->lm_ptr[j + 1] == B->lm_ptr[j]) A[(size_t)(n + 3 * j + a) * N + n + 3 * j + a] = 1.0; /* unused landmark */
                x[n + 3 * j + a] = gl[3 * j + a];
                for (int bb = 0; bb < 5; ++bb) {
                    A[(size_t)(n + 3 * j + a) * N + 6 * nc + bb] = ElK[15 * j + 3 * bb + a];
                    A[(size_t)(6 * nc + bb) * N + n + 3 * j + a] = ElK[15 * j + 3 * bb + a];
                }
            }
        for (int o = 0; o < S->no; ++o) {
            const int c = (int)P->obs_cam[o], j = (int)P->obs_point[o];
            const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o;
            for (int a = 0; a < 6; ++a)
                for (int bb = 0; bb < 3; ++bb) {
                    const double v = Ap[a] * Al[bb] + Ap[6 + a] * Al[3 + bb];
                    A[(size_t)(6 * c + a) * N + n + 3 * j + bb] += v;
                    A[(size_t)(n + 3 * j + bb) * N + 6 * c + a] += v;
                }
        }
        ok = cholesky_solve(A, x, N);
        if (ok) {
            memcpy(dc, x, sizeof(double) * (size_t)n);
            memcpy(dl, x + n, sizeof(double) * 3 * (size_t)nl);
        }
        free(A); free(x);
    } else {
        double* Sm = (double*)calloc((size_t)n * n, sizeof(double));
        double* V = (double*)malloc(sizeof(double) * 9 * (size_t)(nl > 0 ? nl : 1));
        for (int i = 0; i < nc; ++i)
            for (int a = 0; a < 6; ++a) {
                for (int bb = 0; bb < 6; ++bb) Sm[(size_t)(6 * i + a) * n + 6 * i + bb] = Hcc[36 * i + 6 * a + bb];
                for (int bb = 0; bb < 5; ++bb) {
                    Sm[(size_t)(6 * i + a) * n + 6 * nc + bb] = HcK[30 * i + 5 * a + bb];
                    Sm[(size_t)(6 * nc + bb) * n + 6 * i + a] = HcK[30 * i + 5 * a + bb];
                }
                dc[6 * i + a] = gc[6 * i + a];
            }
        for (int a = 0; a < 5; ++a) {
            for (int bb = 0; bb < 5; ++bb) Sm[(size_t)(6 * nc + a) * n + 6 * nc + bb] = HKK[5 * a + bb];
            dc[6 * nc + a] = gK[a];
        }
        for (int j = 0; j < nl; ++j) {
            if (B->lm_ptr[j + 1] == B->lm_ptr[j]) { memset(V + 9 * j, 0, 9 * sizeof(double)); continue; }
            if (!inv3_spd(Hll + 9 * j, V + 9 * j)) ok = 0;
        }
        if (ok) {
            /* E_o = Ap^T Al (6x3) per observation */
            double* Eall = (double*)malloc(sizeof(double) * 18 * (size_t)(S->no > 0 ? S->no : 1));
#pragma omp parallel for schedule(static)
            for (int o = 0; o < S->no; ++o) {
                const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o;
                for (int a = 0; a < 6; ++a)
                    for (int bb = 0; bb < 3; ++bb) Eall[18 * (size_t)o + 3 * a + bb] = Ap[a] * Al[bb] + Ap[6 + a] * Al[3 + bb];
            }
            /* landmark elimination; camera block-rows are owned by threads -> deterministic, no atomics */
#pragma omp parallel
            {
#ifdef _OPENMP
                const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
                const int tid = 0, nt = 1;
#endif
                for (int j = 0; j < nl; ++j) {
                    const int o0 = B->lm_ptr[j], o1 = B->lm_ptr[j + 1];
                    if (o0 == o1) continue;
                    const double* Vj = V + 9 * j;
                    double VEk[15], Vg[3]; /* V * ElK^T (3x5), V * gl */
                    int mine = 0;
                    for (int p = o0; p < o1; ++p) mine |= ((int)P->obs_cam[B->lm_obs[p]] % nt) == tid;
                    if (!mine) continue;
                    for (int a = 0; a < 3; ++a) {
                        for (int bb = 0; bb < 5; ++bb)
                            VEk[5 * a + bb] = Vj[3 * a] * ElK[15 * j + 3 * bb] + Vj[3 * a + 1] * ElK[15 * j + 3 * bb + 1] + Vj[3 * a + 2] * ElK[15 * j + 3 * bb + 2];
                        Vg[a] = Vj[3 * a] * gl[3 * j] + Vj[3 * a + 1] * gl[3 * j + 1] + Vj[3 * a + 2] * gl[3 * j + 2];
                    }
                    for (int p = o0; p < o1; ++p) {
                        const int o = B->lm_obs[p], c = (int)P->obs_cam[o];
                        if (c % nt != tid) continue;
                        const double* E = Eall + 18 * (size_t)o;
                        double EV[18]; /* E V */
                        for (int a = 0; a < 6; ++a)
                            for (int bb = 0; bb < 3; ++bb) EV[3 * a + bb] = E[3 * a] * Vj[bb] + E[3 * a + 1] * Vj[3 + bb] + E[3 * a + 2] * Vj[6 + bb];
                        for (int q = o0; q < o1; ++q) {
                            const int o2 = B->lm_obs[q], c2 = (int)P->obs_cam[o2];
                            const double* E2 = Eall + 18 * (size_t)o2;
                            double* Sb = Sm + (size_t)(6 * c) * n + 6 * c2;
                            for (int a = 0; a < 6; ++a)
                                for (int bb = 0; bb < 6; ++bb)
                                    Sb[(size_t)a * n + bb] -= EV[3 * a] * E2[3 * bb] + EV[3 * a + 1] * E2[3 * bb + 1] + EV[3 * a + 2] * E2[3 * bb + 2];
                        }
                        for (int a = 0; a < 6; ++a) {
                            for (int bb = 0; bb < 5; ++bb) /* S[c,K] -= E V ElK^T */
                                Sm[(size_t)(6 * c + a) * n + 6 * nc + bb] -= E[3 * a] * VEk[bb] + E[3 * a + 1] * VEk[5 + bb] + E[3 * a + 2] * VEk[10 + bb];
                            dc[6 * c + a] -= E[3 * a] * Vg[0] + E[3 * a + 1] * Vg[1] + E[3 * a + 2] * Vg[2];
                        }
                    }
                }
            }
            free(Eall);
            for (int j = 0; j < nl; ++j) { /* K row/column: serial (one owner) */
                if (B->lm_ptr[j + 1] == B->lm_ptr[j]) continue;
                const double* Vj = V + 9 * j;
                double VEk[15], Vg[3];
                for (int a = 0; a < 3; ++a) {
                    for (int bb = 0; bb < 5; ++bb)
                        VEk[5 * a + bb] = Vj[3 * a] * ElK[15 * j + 3 * bb] + Vj[3 * a + 1] * ElK[15 * j + 3 * bb + 1] + Vj[3 * a + 2] * ElK[15 * j + 3 * bb + 2];
                    Vg[a] = Vj[3 * a] * gl[3 * j] + Vj[3 * a + 1] * gl[3 * j + 1] + Vj[3 * a + 2] * gl[3 * j + 2];
                }
                for (int a = 0; a < 5; ++a) {
                    for (int bb = 0; bb < 5; ++bb)
                        Sm[(size_t)(6 * nc + a) * n + 6 * nc + bb] -= ElK[15 * j + 3 * a] * VEk[bb] + ElK[15 * j + 3 * a + 1] * VEk[5 + bb] + ElK[15 * j + 3 * a + 2] * VEk[10 + bb];
                    dc[6 * nc + a] -= ElK[15 * j + 3 * a] * Vg[0] + ElK[15 * j + 3 * a + 1] * Vg[1] + ElK[15 * j + 3 * a + 2] * Vg[2];
                }
            }
            for (int i = 0; i < 6 * nc; ++i) /* mirror the K block-column into the K block-row */
                for (int bb = 0; bb < 5; ++bb) Sm[(size_t)(6 * nc + bb) * n + i] = Sm[(size_t)i * n + 6 * nc + bb];
            if (Sout) memcpy(Sout, Sm, sizeof(double) * (size_t)n * n);
            if (gout) memcpy(gout, dc, sizeof(double) * (size_t)n);
            ok = cholesky_solve(Sm, dc, n);
        }
        if (ok) { /* back-substitution */
#pragma omp parallel for schedule(static)
            for (int j = 0; j < nl; ++j) {
                double rhs[3] = {gl[3 * j], gl[3 * j + 1], gl[3 * j + 2]};
                for (int p = B->lm_ptr[j]; p < B->lm_ptr[j + 1]; ++p) {
                    const int o = B->lm_obs[p], c = (int)P->obs_cam[o];
                    const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o;
                    double t0 = 0, t1 = 0; /* Ap * dc_c (2) */
                    for (int a = 0; a < 6; ++a) { t0 += Ap[a] * dc[6 * c + a]; t1 += Ap[6 + a] * dc[6 * c + a]; }
                    for (int a = 0; a < 3; ++a) rhs[a] -= Al[a] * t0 + Al[3 + a] * t1;
                }
                for (int a = 0; a < 3; ++a)
                    for (int bb = 0; bb < 5; ++bb) rhs[a] -= ElK[15 * j + 3 * bb + a] * dc[6 * nc + bb];
                const double* Vj = V + 9 * j;
                for (int a = 0; a < 3; ++a) dl[3 * j + a] = Vj[3 * a] * rhs[0] + Vj[3 * a + 1] * rhs[1] + Vj[3 * a + 2] * rhs[2];
            }
        }
        free(Sm); free(V);
    }
    free(Hcc); free(HcK); free(gc); free(Hll); free(gl); free(ElK);
    return ok;
}

/* linear.error(delta) = 1/2 sum |A delta - b|^2 over the reweighted, undamped factors */
static double linear_error(const ba_t* B, const lin_t* L, const double* dc, const double* dl) {
    const eacham_ba_problem* P = B->P;
    const state_t* S = &B->S;
    const int nc = S->nc;
    double err = 0.0;
    {
        const int nb = (S->no + OBS_BLOCK - 1) / OBS_BLOCK;
        double* part = (double*)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
#pragma omp parallel for schedule(static)
        for (int blk = 0; blk < nb; ++blk) {
            double e = 0.0;
            const int o1 = (blk + 1) * OBS_BLOCK < S->no ? (blk + 1) * OBS_BLOCK : S->no;
            for (int o = blk * OBS_BLOCK; o < o1; ++o) {
        const int c = (int)P->obs_cam[o], j = (int)P->obs_point[o];
        const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o, *Ak = L->Ak + 10 * (size_t)o;
        double r0 = -L->b[2 * (size_t)o], r1 = -L->b[2 * (size_t)o + 1];
        if (dc) {
            for (int a = 0; a < 6; ++a) { r0 += Ap[a] * dc[6 * c + a]; r1 += Ap[6 + a] * dc[6 * c + a]; }
            for (int a = 0; a < 3; ++a) { r0 += Al[a] * dl[3 * j + a]; r1 += Al[3 + a] * dl[3 * j + a]; }
            for (int a = 0; a < 5; ++a) { r0 += Ak[a] * dc[6 * nc + a]; r1 += Ak[5 + a] * dc[6 * nc + a]; }
        }
                e += 0.5 * (r0 * r0 + r1 * r1);
            }
            part[blk] = e;
        }
        for (int blk = 0; blk < nb; ++blk) err += part[blk];
        free(part);
    }
    for (int i = 0; i < nc; ++i)
        for (int a = 0; a < 6; ++a) {
            const double r = L->Pw[6 * i + a] * (dc ? dc[6 * i + a] : 0.0) - L->Pb[6 * i + a];
            err += 0.5 * r * r;
        }
    for (int j = 0; j < S->nl; ++j) {
        if (B->lm_ptr[j + 1] == B->lm_ptr[j]) continue;
        for (int a = 0; a < 3; ++a) {
            const double r = L->Lw[j] * (dl ? dl[3 * j + a] : 0.0) - L->Lb[3 * j + a];
            err += 0.5 * r * r;
        }
    }
    for (int a = 0; a < 5; ++a) {
        const double r = L->Kw[a] * (dc ? dc[6 * nc + a] : 0.0) - L->Kb[a];
        err += 0.5 * r * r;
    }
    return err;
}

static void retract(const state_t* S, const double* dc, const double* dl, state_t* out) {
    for (int i = 0; i < S->nc; ++i) pose_retract(&S->pose[i], dc + 6 * i, &out->pose[i]);
    for (int k = 0; k < 3 * S->nl; ++k) out->pt[k] = S->pt[k] + dl[k];
    for (int a = 0; a < 5; ++a) out->K[a] = S->K[a] + dc[6 * S->nc + a];
}

/* ---- DoglegOptimizer pieces (GTSAM 4.1.1 DoglegOptimizer.cpp / DoglegOptimizerImpl.{h,cpp}) ----- */
/* g = A^T b over all factors (GTSAM's gradientAtZero is -g), layout like dc | dl */
static void gradient(const ba_t* B, const lin_t* L, double* gc, double* gl) {
    const eacham_ba_problem* P = B->P;
    const state_t* S = &B->S;
    const int nc = S->nc;
    memset(gc, 0, sizeof(double) * (size_t)(6 * nc + 5));
    memset(gl, 0, sizeof(double) * 3 * (size_t)S->nl);
    for (int o = 0; o < S->no; ++o) {
        const int c = (int)P->obs_cam[o], j = (int)P->obs_point[o];
        const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o, *Ak = L->Ak + 10 * (size_t)o, *b = L->b + 2 * (size_t)o;
        for (int a = 0; a < 6; ++a) gc[6 * c + a] += Ap[a] * b[0] + Ap[6 + a] * b[1];
        for (int a = 0; a < 3; ++a) gl[3 * j + a] += Al[a] * b[0] + Al[3 + a] * b[1];
        for (int a = 0; a < 5; ++a) gc[6 * nc + a] += Ak[a] * b[0] + Ak[5 + a] * b[1];
    }
    for (int i = 0; i < nc; ++i)
        for (int a = 0; a < 6; ++a) gc[6 * i + a] += L->Pw[6 * i + a] * L->Pb[6 * i + a];
    for (int j = 0; j < S->nl; ++j) {
        if (B->lm_ptr[j + 1] == B->lm_ptr[j]) continue;
        for (int a = 0; a < 3; ++a) gl[3 * j + a] += L->Lw[j] * L->Lb[3 * j + a];
    }
    for (int a = 0; a < 5; ++a) gc[6 * nc + a] += L->Kw[a] * L->Kb[a];
}

/* (A x) . (A y) over all factors */
static double hessian_form(const ba_t* B, const lin_t* L, const double* xc, const double* xl, const double* yc, const double* yl) {
    const eacham_ba_problem* P = B->P;
    const state_t* S = &B->S;
    const int nc = S->nc;
    double tot = 0.0;
    {
        const int nb = (S->no + OBS_BLOCK - 1) / OBS_BLOCK;
        double* part = (double*)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
#pragma omp parallel for schedule(static)
        for (int blk = 0; blk < nb; ++blk) {
            double e = 0.0;
            const int o1 = (blk + 1) * OBS_BLOCK < S->no ? (blk + 1) * OBS_BLOCK : S->no;
            for (int o = blk * OBS_BLOCK; o < o1; ++o) {
        const int c = (int)P->obs_cam[o], j = (int)P->obs_point[o];
        const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o, *Ak = L->Ak + 10 * (size_t)o;
        double x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        for (int a = 0; a < 6; ++a) { x0 += Ap[a] * xc[6 * c + a]; x1 += Ap[6 + a] * xc[6 * c + a]; y0 += Ap[a] * yc[6 * c + a]; y1 += Ap[6 + a] * yc[6 * c + a]; }
        for (int a = 0; a < 3; ++a) { x0 += Al[a] * xl[3 * j + a]; x1 += Al[3 + a] * xl[3 * j + a]; y0 += Al[a] * yl[3 * j + a]; y1 += Al[3 + a] * yl[3 * j + a]; }
        for (int a = 0; a < 5; ++a) { x0 += Ak[a] * xc[6 * nc + a]; x1 += Ak[5 + a] * xc[6 * nc + a]; y0 += Ak[a] * yc[6 * nc + a]; y1 += Ak[5 + a] * yc[6 * nc + a]; }
                e += x0 * y0 + x1 * y1;
            }
            part[blk] = e;
        }
        for (int blk = 0; blk < nb; ++blk) tot += part[blk];
        free(part);
    }
    for (int i = 0; i < nc; ++i)
        for (int a = 0; a < 6; ++a) tot += L->Pw[6 * i + a] * L->Pw[6 * i + a] * xc[6 * i + a] * yc[6 * i + a];
    for (int j = 0; j < S->nl; ++j) {
        if (B->lm_ptr[j + 1] == B->lm_ptr[j]) continue;
        for (int a = 0; a < 3; ++a) tot += L->Lw[j] * L->Lw[j] * xl[3 * j + a] * yl[3 * j + a];
    }
    for (int a = 0; a < 5; ++a) tot += L->Kw[a] * L->Kw[a] * xc[6 * nc + a] * yc[6 * nc + a];
    return tot;
}

static double dot2(const double* ac, const double* al, const double* bc, const double* bl, int n, int nl3) {
    double t = 0.0;
    for (int k = 0; k < n; ++k) t += ac[k] * bc[k];
    for (int k = 0; k < nl3; ++k) t += al[k] * bl[k];
    return t;
}

/* ---- the iterative solve the reference can select: PCG + block-Jacobi (BundleAdjuster.cpp:192-200) -------------
 * params.linearSolverType = Iterative with gtsam::PCGSolverParameters{preconditioner = BlockJacobi, epsilon_abs =
 * epsilon_rel = 1e-10}. GTSAM 4.1.1 (gtsam/linear/PCGSolver.cpp, iterative-inl.h preconditionedConjugateGradient,
 * Preconditioner.cpp BlockJacobiPreconditioner; not in the reference tree, restated from memory like Appendix A):
 *   the system is the DAMPED Gauss-Newton system over ALL variables (poses, landmarks, calibration); the
 *   preconditioner is the Cholesky factor of its diagonal blocks (6x6, 3x3, 5x5), applied split (L^-1 . L^-T);
 *   x0 = 0; gamma = |L^-1 r|^2; stop when gamma <= max(epsilon_abs, epsilon_rel^2 * gamma0) or after
 *   maxIterations = 500 (ConjugateGradientParameters defaults: minIterations 1, reset 501).
 * The split form and the textbook form z = M^-1 r, M = L L^T generate the same iterates; the latter is coded.
 * Returns the number of iterations (0 if a diagonal block is not positive definite). */
typedef struct { double* c; double* l; } vec_t; /* [6 nc + 5] cameras + K, [3 nl] landmarks */

static void hess_apply(const ba_t* B, const lin_t* L, double lambda, const double* Dc, const double* Dl, const vec_t x, vec_t y) {
    const eacham_ba_problem* P = B->P;
    const state_t* S = &B->S;
    const int nc = S->nc, n = 6 * nc + 5;
    memset(y.c, 0, sizeof(double) * (size_t)n);
    memset(y.l, 0, sizeof(double) * 3 * (size_t)S->nl);
    for (int o = 0; o < S->no; ++o) { /* y += A_o^T (A_o x) */
        const int c = (int)P->obs_cam[o], j = (int)P->obs_point[o];
        const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o, *Ak = L->Ak + 10 * (size_t)o;
        double r0 = 0, r1 = 0;
        for (int a = 0; a < 6; ++a) { r0 += Ap[a] * x.c[6 * c + a]; r1 += Ap[6 + a] * x.c[6 * c + a]; }
        for (int a = 0; a < 3; ++a) { r0 += Al[a] * x.l[3 * j + a]; r1 += Al[3 + a] * x.l[3 * j + a]; }
        for (int a = 0; a < 5; ++a) { r0 += Ak[a] * x.c[6 * nc + a]; r1 += Ak[5 + a] * x.c[6 * nc + a]; }
        for (int a = 0; a < 6; ++a) y.c[6 * c + a] += Ap[a] * r0 + Ap[6 + a] * r1;
        for (int a = 0; a < 3; ++a) y.l[3 * j + a] += Al[a] * r0 + Al[3 + a] * r1;
        for (int a = 0; a < 5; ++a) y.c[6 * nc + a] += Ak[a] * r0 + Ak[5 + a] * r1;
    }
    for (int i = 0; i < nc; ++i)
        for (int a = 0; a < 6; ++a) y.c[6 * i + a] += L->Pw[6 * i + a] * L->Pw[6 * i + a] * x.c[6 * i + a];
    for (int j = 0; j < S->nl; ++j) {
        if (B->lm_ptr[j + 1] == B->lm_ptr[j]) { for (int a = 0; a < 3; ++a) y.l[3 * j + a] = x.l[3 * j + a]; continue; } /* unused: identity */
        for (int a = 0; a < 3; ++a) y.l[3 * j + a] += L->Lw[j] * L->Lw[j] * x.l[3 * j + a];
    }
    for (int a = 0; a < 5; ++a) y.c[6 * nc + a] += L->Kw[a] * L->Kw[a] * x.c[6 * nc + a];
    for (int k = 0; k < n; ++k) y.c[k] += lambda * Dc[k] * x.c[k];
    for (int k = 0; k < 3 * S->nl; ++k) y.l[k] += lambda * Dl[k] * x.l[k];
}

static int chol_small(double* A, int m) { /* in place, lower; 0 if not positive definite */
    for (int j = 0; j < m; ++j) {
        double d = A[j * m + j];
        for (int k = 0; k < j; ++k) d -= A[j * m + k] * A[j * m + k];
        if (!(d > 0.0) || !isfinite(d)) return 0;
        d = sqrt(d);
        A[j * m + j] = d;
        for (int i = j + 1; i < m; ++i) {
            double v = A[i * m + j];
            for (int k = 0; k < j; ++k) v -= A[i * m + k] * A[j * m + k];
            A[i * m + j] = v / d;
        }
    }
    return 1;
}
static void chol_small_solve(const double* Lf, int m, const double* r, double* z) { /* z = (L L^T)^-1 r */
    double t[6];
    for (int i = 0; i < m; ++i) {
        double v = r[i];
        for (int k = 0; k < i; ++k) v -= Lf[i * m + k] * t[k];
        t[i] = v / Lf[i * m + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double v = t[i];
        for (int k = i + 1; k < m; ++k) v -= Lf[k * m + i] * z[k];
        z[i] = v / Lf[i * m + i];
    }
}

static int solve_step_pcg(const ba_t* B, const lin_t* L, double lambda, double* dc, double* dl, int* iterations) {
    const eacham_ba_problem* P = B->P;
    const state_t* S = &B->S;
    const int nc = S->nc, nl = S->nl, n = 6 * nc + 5, nl3 = 3 * nl;
    const double eps_abs = 1e-10, eps_rel = 1e-10; /* BundleAdjuster.cpp:197-198 */
    const int max_it = 500, min_it = 1, reset = 501;
    /* undamped diagonal blocks, the damping diagonal D = clamp(diag H), the right-hand side g = A^T b */
    double* Hcc = (double*)calloc((size_t)36 * (nc > 0 ? nc : 1), sizeof(double));
    double* Hll = (double*)calloc((size_t)9 * (nl > 0 ? nl : 1), sizeof(double));
    double HKK[25] = {0};
    for (int o = 0; o < S->no; ++o) {
        const int c = (int)P->obs_cam[o], j = (int)P->obs_point[o];
        const double *Ap = L->Ap + 12 * (size_t)o, *Al = L->Al + 6 * (size_t)o, *Ak = L->Ak + 10 * (size_t)o;
        for (int a = 0; a < 6; ++a)
            for (int bb = 0; bb < 6; ++bb) Hcc[36 * c + 6 * a + bb] += Ap[a] * Ap[bb] + Ap[6 + a] * Ap[6 + bb];
        for (int a = 0; a < 3; ++a)
            for (int bb = 0; bb < 3; ++bb) Hll[9 * j + 3 * a + bb] += Al[a] * Al[bb] + Al[3 + a] * Al[3 + bb];
        for (int a = 0; a < 5; ++a)
            for (int bb = 0; bb < 5; ++bb) HKK[5 * a + bb] += Ak[a] * Ak[bb] + Ak[5 + a] * Ak[5 + bb];
    }
    for (int i = 0; i < nc; ++i)
        for (int a = 0; a < 6; ++a) Hcc[36 * i + 7 * a] += L->Pw[6 * i + a] * L->Pw[6 * i + a];
    for (int j = 0; j < nl; ++j) {
        if (B->lm_ptr[j + 1] == B->lm_ptr[j]) { Hll[9 * j] = Hll[9 * j + 4] = Hll[9 * j + 8] = 1.0; continue; }
        for (int a = 0; a < 3; ++a) Hll[9 * j + 4 * a] += L->Lw[j] * L->Lw[j];
    }
    for (int a = 0; a < 5; ++a) HKK[6 * a] += L->Kw[a] * L->Kw[a];
    double* Dc = (double*)malloc(sizeof(double) * (size_t)n);
    double* Dl = (double*)malloc(sizeof(double) * (size_t)(nl3 > 0 ? nl3 : 1));
    for (int i = 0; i < nc; ++i)
        for (int a = 0; a < 6; ++a) Dc[6 * i + a] = clampd(Hcc[36 * i + 7 * a], 1e-6, 1e32);
    for (int a = 0; a < 5; ++a) Dc[6 * nc + a] = clampd(HKK[6 * a], 1e-6, 1e32);
    for (int j = 0; j < nl; ++j)
        for (int a = 0; a < 3; ++a) Dl[3 * j + a] = B->lm_ptr[j + 1] == B->lm_ptr[j] ? 0.0 : clampd(Hll[9 * j + 4 * a], 1e-6, 1e32);
    int ok = 1;
    for (int i = 0; i < nc; ++i) { /* block-Jacobi: Cholesky factors of the damped diagonal blocks */
        for (int a = 0; a < 6; ++a) Hcc[36 * i + 7 * a] += lambda * Dc[6 * i + a];
        ok &= chol_small(Hcc + 36 * i, 6);
    }
    for (int j = 0; j < nl; ++j) {
        for (int a = 0; a < 3; ++a) Hll[9 * j + 4 * a] += lambda * Dl[3 * j + a];
        ok &= chol_small(Hll + 9 * j, 3);
    }
    for (int a = 0; a < 5; ++a) HKK[6 * a] += lambda * Dc[6 * nc + a];
    ok &= chol_small(HKK, 5);
    int k = 0;
    if (ok) {
        vec_t x = {dc, dl}, r, z, p, q;
        double* buf = (double*)calloc((size_t)4 * (n + (nl3 > 0 ? nl3 : 1)), sizeof(double));
        r.c = buf; r.l = r.c + n; z.c = r.l + (nl3 > 0 ? nl3 : 1); z.l = z.c + n;
        p.c = z.l + (nl3 > 0 ? nl3 : 1); p.l = p.c + n; q.c = p.l + (nl3 > 0 ? nl3 : 1); q.l = q.c + n;
        memset(dc, 0, sizeof(double) * (size_t)n);
        memset(dl, 0, sizeof(double) * (size_t)nl3);
        gradient(B, L, r.c, r.l); /* r = b - A 0 = g */
#define PRECOND()                                                                                       \
        do {                                                                                            \
            for (int i = 0; i < nc; ++i) chol_small_solve(Hcc + 36 * i, 6, r.c + 6 * i, z.c + 6 * i);     \
            chol_small_solve(HKK, 5, r.c + 6 * nc, z.c + 6 * nc);                                       \
            for (int j = 0; j < nl; ++j) chol_small_solve(Hll + 9 * j, 3, r.l + 3 * j, z.l + 3 * j);      \
        } while (0)
        PRECOND();
        memcpy(p.c, z.c, sizeof(double) * (size_t)n);
        memcpy(p.l, z.l, sizeof(double) * (size_t)nl3);
        double gamma = dot2(r.c, r.l, z.c, z.l, n, nl3);
        const double thr = fmax(eps_abs, eps_rel * eps_rel * gamma);
        for (k = 1; k <= max_it && (gamma > thr || k <= min_it); ++k) {
            if (k % reset == 0) { /* (never reached with maxIterations 500 < reset 501; kept for the record) */
                hess_apply(B, L, lambda, Dc, Dl, x, q);
                gradient(B, L, r.c, r.l);
                for (int i = 0; i < n; ++i) r.c[i] -= q.c[i];
                for (int i = 0; i < nl3; ++i) r.l[i] -= q.l[i];
                PRECOND();
                memcpy(p.c, z.c, sizeof(double) * (size_t)n);
                memcpy(p.l, z.l, sizeof(double) * (size_t)nl3);
                gamma = dot2(r.c, r.l, z.c, z.l, n, nl3);
            }
            hess_apply(B, L, lambda, Dc, Dl, p, q);
            const double alpha = gamma / dot2(p.c, p.l, q.c, q.l, n, nl3);
            for (int i = 0; i < n; ++i) { dc[i] += alpha * p.c[i]; r.c[i] -= alpha * q.c[i]; }
            for (int i = 0; i < nl3; ++i) { dl[i] += alpha * p.l[i]; r.l[i] -= alpha * q.l[i]; }
            PRECOND();
            const double prev = gamma;
            gamma = dot2(r.c, r.l, z.c, z.l, n, nl3);
            const double beta = gamma / prev;
            for (int i = 0; i < n; ++i) p.c[i] = z.c[i] + beta * p.c[i];
            for (int i = 0; i < nl3; ++i) p.l[i] = z.l[i] + beta * p.l[i];
        }
#undef PRECOND
        --k;
        for (int j = 0; j < nl; ++j) /* landmarks without observations take no step */
            if (B->lm_ptr[j + 1] == B->lm_ptr[j]) dl[3 * j] = dl[3 * j + 1] = dl[3 * j + 2] = 0.0;
        free(buf);
    }
    if (iterations) *iterations = k;
    free(Hcc); free(Hll); free(Dc); free(Dl);
    return ok ? (k > 0 ? k : 1) : 0;
}

/* DoglegOptimizerImpl::ComputeDoglegPoint / ComputeBlend: x_d = cu * x_u + cn * x_n */
static void dogleg_point(double delta, double uu, double un, double nn, double* cu, double* cn) {
    const double DeltaSq = delta * delta;
    if (DeltaSq < uu) {
        *cu = sqrt(DeltaSq / uu);
        *cn = 0.0;
    } else if (DeltaSq < nn) {
        const double a = uu - 2. * un + nn, b = 2. * (un - uu), c = uu - DeltaSq;
        const double sq = sqrt(b * b - 4 * a * c);
        const double tau1 = (-b + sq) / (2. * a), tau2 = (-b - sq) / (2. * a);
        const double tau = (0.0 <= tau1 && tau1 <= 1.0) ? tau1 : tau2;
        *cu = 1. - tau;
        *cn = tau;
    } else {
        *cu = 0.0;
        *cn = 1.0;
    }
}

/* ---- exported: error at the initial values ---------------------------------------------------- */
double oracle_ba_error(const eacham_ba_problem* P) {
    ba_t B;
    if (ba_init(&B, P)) return NAN;
    const double e = graph_error(&B, &B.S);
    ba_free(&B);
    return e;
}

/* ---- exported: one damped step at the initial values (same contract as eacham_ba_debug_step) -- */
int oracle_ba_step(const eacham_ba_problem* P, double lambda, int mode, double* Sout, double* gout,
                   double* dc, double* dl, double* error, double* lin_change) {
    ba_t B;
    if (ba_init(&B, P)) return -1;
    lin_t L;
    lin_alloc(&L, &B.S);
    linearize(&B, &B.S, &L);
    memset(dl, 0, sizeof(double) * 3 * (size_t)P->n_points);
    /* mode 0 = Schur complement, 1 = dense over all variables, 2 = PCG + block-Jacobi (no Sout / gout) */
    const int ok = mode == 2 ? solve_step_pcg(&B, &L, lambda, dc, dl, 0) : solve_step(&B, &L, lambda, mode, dc, dl, Sout, gout);
    if (error) *error = graph_error(&B, &B.S);
    if (lin_change) *lin_change = ok ? linear_error(&B, &L, 0, 0) - linear_error(&B, &L, dc, dl) : NAN;
    lin_free(&L);
    ba_free(&B);
    return ok ? 0 : 1;
}

/* ---- exported: RefineBA's optimisation (LevenbergMarquardtOptimizer::optimize, Appendix A.4) ---- */
int oracle_ba_solve(const eacham_ba_problem* P, const eacham_ba_options* O, eacham_ba_result* R, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    ba_t B;
    if (ba_init(&B, P)) return -1;
    R->trace_len = 0;
    R->outer_iterations = R->inner_iterations = 0;
    R->final_lambda = 0.0;
    memcpy(R->cam_T_wc, P->cam_T_wc, sizeof(double) * 16 * (size_t)P->n_cams);
    memcpy(R->points, P->points, sizeof(double) * 3 * (size_t)P->n_points);
    memcpy(R->K, P->K, sizeof(double) * 4);
    if (B.n_landmarks_used < O->min_landmarks) { /* BundleAdjuster.cpp:166-169 */
        R->status = EACHAM_BA_SKIPPED;
        R->initial_error = R->final_error = NAN;
        ba_free(&B);
        return 0;
    }
    if (O->method != EACHAM_BA_LM && O->method != EACHAM_BA_DOGLEG) { ba_free(&B); return -4; }
    state_t* S = &B.S;
    const int n = 6 * S->nc + 5;
    state_t N = *S; /* tentative values */
    N.pose = (pose_t*)malloc(sizeof(pose_t) * (size_t)(S->nc > 0 ? S->nc : 1));
    N.pt = (double*)malloc(sizeof(double) * 3 * (size_t)(S->nl > 0 ? S->nl : 1));
    double* dc = (double*)calloc((size_t)n, sizeof(double));
    double* dl = (double*)calloc((size_t)3 * (S->nl > 0 ? S->nl : 1), sizeof(double));
    lin_t L;
    lin_alloc(&L, S);

    /* SetCeresDefaults + overrides (BundleAdjuster.cpp:184-190) */
    const double lambdaUpper = 1e32, lambdaLower = 1e-16, minModelFidelity = 1e-3;
    const double relTol = (double)O->max_tolerance, absTol = (double)O->max_tolerance, errorTol = 0.0;
    const double lambdaFactor = 2.0; /* SetCeresDefaults */
    double lambda = 1e-4, factor = lambdaFactor;
    int iterations = 0, inner = 0;
    double error = graph_error(&B, S);
    R->initial_error = error;
    double newErrorOuter = error, currentError = error;
    int indeterminate = 0;
    long pcg_total = 0;
    if (O->method == EACHAM_BA_DOGLEG) { /* DoglegOptimizer (BundleAdjuster.cpp:204-214): delta0 = config.delta */
        double delta = (double)O->delta;
        double* gc = (double*)calloc((size_t)n, sizeof(double));
        double* gl = (double*)calloc((size_t)3 * (S->nl > 0 ? S->nl : 1), sizeof(double));
        const int nl3 = 3 * S->nl;
        if (error > errorTol && iterations < O->max_iter) {
            do { /* NonlinearOptimizer::defaultOptimize */
                currentError = newErrorOuter;
                /* ---- DoglegOptimizer::iterate() ---- */
                linearize(&B, S, &L);
                gradient(&B, &L, gc, gl);                       /* g = A^T b = -gradientAtZero */
                const double gg = dot2(gc, gl, gc, gl, n, nl3);
                const double gHg = hessian_form(&B, &L, gc, gl, gc, gl);
                const double alpha = gg / gHg;                  /* dx_u = -(g.g / |R g|^2) grad = alpha * g */
                memset(dl, 0, sizeof(double) * (size_t)nl3);
                const int solved = solve_step(&B, &L, 0.0, 0, dc, dl, 0, 0); /* dx_n: the Gauss-Newton step */
                if (!solved) { indeterminate = 1; break; }      /* GTSAM throws IndeterminantLinearSystemException */
                const double uu = alpha * alpha * gg, un = alpha * dot2(gc, gl, dc, dl, n, nl3), nn = dot2(dc, dl, dc, dl, n, nl3);
                const double uHu = alpha * alpha * gHg, uHn = alpha * hessian_form(&B, &L, gc, gl, dc, dl),
                             nHn = hessian_form(&B, &L, dc, dl, dc, dl);
                const double gu = alpha * gg, gn = dot2(gc, gl, dc, dl, n, nl3);
                const double M_error = linear_error(&B, &L, 0, 0);
                /* DoglegOptimizerImpl::Iterate, mode ONE_STEP_PER_ITERATION */
                double f_new = error, cu = 0, cn = 0;
                int zero_step = 0;
                for (int stay = 1; stay;) {
                    dogleg_point(delta, uu, un, nn, &cu, &cn);
                    const double dnorm = sqrt(cu * cu * uu + 2 * cu * cn * un + cn * cn * nn);
                    {
                        double* xc = (double*)malloc(sizeof(double) * (size_t)n);
                        double* xl = (double*)malloc(sizeof(double) * (size_t)(nl3 > 0 ? nl3 : 1));
                        for (int k = 0; k < n; ++k) xc[k] = cu * alpha * gc[k] + cn * dc[k];
                        for (int k = 0; k < nl3; ++k) xl[k] = cu * alpha * gl[k] + cn * dl[k];
                        retract(S, xc, xl, &N);
                        free(xc); free(xl);
                    }
                    f_new = graph_error(&B, &N);
                    /* M(x) = M(0) - g.x + 1/2 x^T H x */
                    const double new_M_error = M_error - (cu * gu + cn * gn) + 0.5 * (cu * cu * uHu + 2 * cu * cn * uHn + cn * cn * nHn);
                    const double rho = (fabs(error - f_new) < 1e-15 || fabs(M_error - new_M_error) < 1e-15)
                                           ? 0.5 : (error - f_new) / (M_error - new_M_error);
                    int accepted = 1;
                    const double delta_used = delta;
                    if (rho >= 0.75) {
                        const double nd = 3.0 * dnorm;
                        delta = delta > nd ? delta : nd;
                        stay = 0;
                    } else if (rho >= 0.25) {
                        stay = 0;
                    } else if (rho >= 0.0) {
                        if (delta > 1e-5) delta = 0.5 * delta;
                        stay = 0;                               /* ONE_STEP_PER_ITERATION */
                    } else {                                    /* f increased (or NaN): shrink and retry */
                        if (delta > 1e-5) {
                            delta *= 0.5;
                            accepted = 0;
                        } else {
                            zero_step = 1;                      /* dx_d.setZero(); f_error unchanged */
                            stay = 0;
                            accepted = 0;
                        }
                    }
                    if (R->trace && R->trace_len < R->trace_cap) {
                        eacham_ba_trace_row* tr = &R->trace[R->trace_len++];
                        tr->lambda = delta_used; tr->new_error = f_new; tr->lin_change = M_error - new_M_error;
                        tr->accepted = accepted; tr->outer = iterations;
                    }
                    ++inner;
                }
                if (!zero_step) {
                    pose_t* tp = S->pose; S->pose = N.pose; N.pose = tp;
                    double* tq = S->pt; S->pt = N.pt; N.pt = tq;
                    memcpy(S->K, N.K, sizeof(S->K));
                    error = f_new;
                }
                ++iterations;
                newErrorOuter = error;
                /* checkConvergence */
                if (newErrorOuter <= errorTol) break;
                const double absDec = currentError - newErrorOuter, relDec = absDec / currentError;
                const int converged = (relTol != 0.0 && relDec <= relTol) || (absDec <= absTol);
                if (!(iterations < O->max_iter) || converged || !isfinite(currentError)) break;
            } while (1);
        }
        lambda = delta; /* reported in final_lambda */
        free(gc); free(gl);
    } else if (error > errorTol && iterations < O->max_iter) {
        do { /* NonlinearOptimizer::defaultOptimize */
            currentError = newErrorOuter;
            /* ---- iterate() ---- */
            linearize(&B, S, &L);
            for (;;) { /* while (!tryLambda()) */
                int success = 0, stop = 0;
                double newError = INFINITY, linChange = NAN, fidelity = 0.0;
                memset(dl, 0, sizeof(double) * 3 * (size_t)S->nl);
                /* use_preconditioner: the PCG + block-Jacobi the reference configures (BundleAdjuster.cpp:192-200)
                 * instead of the direct solve (the two are compared in tests/test_ba_oracle.py) */
                int pcg_it = 0;
                const int solved = O->use_preconditioner ? solve_step_pcg(&B, &L, lambda, dc, dl, &pcg_it)
                                                              : solve_step(&B, &L, lambda, 0, dc, dl, 0, 0);
                pcg_total += pcg_it;
                if (solved) {
                    const double oldLin = linear_error(&B, &L, 0, 0), newLin = linear_error(&B, &L, dc, dl);
                    linChange = oldLin - newLin;
                    if (linChange >= 0) {
                        retract(S, dc, dl, &N);
                        newError = graph_error(&B, &N);
                        const double cost = error - newError;
                        if (linChange > 2.220446049250313e-16 * oldLin) {
                            fidelity = cost / linChange;
                            success = fidelity > minModelFidelity;
                        }
                        if (fabs(cost) < relTol * error) stop = 1;
                    }
                }
                if (R->trace && R->trace_len < R->trace_cap) {
                    eacham_ba_trace_row* tr = &R->trace[R->trace_len++];
                    tr->lambda = lambda; tr->new_error = newError; tr->lin_change = linChange;
                    tr->accepted = success; tr->outer = iterations;
                }
                ++inner;
                if (success) { /* decreaseLambda */
                    double m = 1.0 - pow(2.0 * fidelity - 1.0, 3);
                    if (m < 1.0 / 3.0) m = 1.0 / 3.0;
                    lambda *= m;
                    /* LevenbergMarquardtState::decreaseLambda, non-fixed-factor branch: two readings of the
                     * line exist (EACHAM_BA_LM_FACTOR_* in eacham_hip.h); RESET = 2 * params.lambdaFactor */
                    factor = O->lm_factor_policy == EACHAM_BA_LM_FACTOR_DOUBLE ? 2.0 * factor : 2.0 * lambdaFactor;
                    if (lambda < lambdaLower) lambda = lambdaLower;
                    pose_t* tp = S->pose; S->pose = N.pose; N.pose = tp;
                    double* tq = S->pt; S->pt = N.pt; N.pt = tq;
                    memcpy(S->K, N.K, sizeof(S->K));
                    error = newError;
                    ++iterations;
                    break;
                } else if (!stop) { /* increaseLambda */
                    lambda *= factor;
                    factor *= 2.0;
                    if (lambda >= lambdaUpper) break; /* giving up */
                } else {
                    break;
                }
            }
            newErrorOuter = error;
            /* checkConvergence */
            if (newErrorOuter <= errorTol) break;
            const double absDec = currentError - newErrorOuter, relDec = absDec / currentError;
            const int converged = (relTol != 0.0 && relDec <= relTol) || (absDec <= absTol);
            if (!(iterations < O->max_iter) || converged || !isfinite(currentError)) break;
        } while (1);
    }
    R->status = indeterminate ? EACHAM_BA_INDETERMINATE : EACHAM_BA_DONE;
    R->final_error = graph_error(&B, S);
    R->final_lambda = lambda;
    R->outer_iterations = iterations;
    R->inner_iterations = inner;
    R->reserved = (int32_t)(pcg_total > 0x7fffffff ? 0x7fffffff : pcg_total); /* oracle only: PCG iterations in total */
    for (int i = 0; i < S->nc; ++i) pose_to_Twc(&S->pose[i], R->cam_T_wc + 16 * (size_t)i);
    memcpy(R->points, S->pt, sizeof(double) * 3 * (size_t)S->nl);
    R->K[0] = S->K[0]; R->K[1] = S->K[1]; R->K[2] = S->K[3]; R->K[3] = S->K[4]; /* fx fy px py (:224-227) */
    lin_free(&L);
    free(N.pose); free(N.pt); free(dc); free(dl);
    ba_free(&B);
    return 0;
}
