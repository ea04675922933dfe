/*
 * solve_oracle.c — CPU restatement of the MINIMAL SOLVERS inside the robust estimators eacham calls
 * (the scoring half is score_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY. PARITY UNPINNED: the solvers live in OpenCV 4.5.5 (conanfile.txt:3), which is not in the
 * reference tree; restated from its published sources (modules/calib3d/src/fundam.cpp HomographyEstimatorCallback::runKernel,
 * five-point.cpp EMEstimatorCallback::runKernel — Nister's five-point algorithm), anchored on the reference's call sites:
 *   cv::findHomography(pts1, pts2, cv::LMEDS, 4.0, mask2, 100, 0.999)              ReconstructionManager.cpp:75
 *   cv::findEssentialMat(pts1, pts2, focal, pp, cv::LMEDS, 0.99, 4.0, 1000, mask)  ReconstructionManager.cpp:57-61
 * OpenCV draws the minimal samples from its own RNG (cv::RNG(-1) state inside RANSACPointSetRegistrator / LMeDS...): the
 * sample INDICES are an input here, so what can be stated — and is tested — is "these correspondences -> these models";
 * end-to-end parity with findHomography / findEssentialMat cannot be pinned by anyone without that RNG stream.
 *   homography4  4 correspondences -> H (3x3 row-major, H[8] = 1): points normalised per set (centroid, mean absolute
 *                deviation per axis), LtL = sum of the 2 x 9 constraint rows' outer products, eigenvector of the smallest
 *                eigenvalue (cyclic Jacobi), denormalised, divided by H[8] — fundam.cpp's sequence.
 *   essential5   5 correspondences (pixels + K, normalised as findEssentialMat does) -> up to 10 E (3x3 row-major, unit
 *                Frobenius norm, ascending in the hidden variable z): null space of the 5 x 9 epipolar system (Householder),
 *                E = xX + yY + zZ + W, the ten cubic constraints det E = 0 and 2 E E^T E - tr(E E^T) E = 0 as a 10 x 20
 *                matrix in Nister's monomial order, Gauss-Jordan, the 3 x 3 polynomial matrix B(z), det B = degree-10
 *                polynomial, its real roots (Durand-Kerner on the monic polynomial, Newton polish), x and y from B(z).
 *                five-point.cpp finds the roots with cv::solvePoly and the null space with cv::SVD: same solution set.
 *   epnp         m >= 5 object points + pixels + K -> R | t: the minimal-set kernel (5 points) and the all-inlier refit of
 *                cv::solvePnPRansac(..., 10000, 4.0f, 0.999f, inliers, cv::SOLVEPNP_EPNP)   ReconstructionManager.cpp:227-228
 *                restated from the EPnP paper (Lepetit, Moreno-Noguer, Fua, IJCV 2009) and the structure of OpenCV's epnp.cpp:
 *                four control points (centroid + principal axes), barycentric coordinates, M^T M of the 2m x 12 projection
 *                system, its four smallest eigenvectors, the 6 x 10 control-point distance system, three linearised starts
 *                (4 / 3 / 5 unknown products) each polished by five Gauss-Newton steps, absolute orientation, and the start
 *                with the smallest reprojection error wins. Own choices where the result set is the same: Jacobi instead of
 *                cv::SVD, Householder least squares, Horn's quaternion form for the absolute orientation (epnp.cpp: SVD of
 *                the 3 x 3 correlation with a row flip for det < 0). COPLANAR sets (the smallest spread of the object points is
 *                <= 1e-12 of the largest) take the paper's planar form (section 3.4 there): THREE control points (centroid +
 *                the two in-plane axes), the 2m x 9 system — carried inside the same 12 x 12 arrays: the fourth control point
 *                gets barycentric coordinate 0 and its three diagonal entries of M^T M a value above every eigenvalue of the
 *                9 x 9 part, so that its unit vectors are never rotated and never among the four smallest —, the three distance
 *                equations of the control points (0,1), (0,2), (1,2), two linearised starts (N = 1: b00; N = 2: b00 b01 b11)
 *                polished by Gauss-Newton on their own 1 / 2 unknowns. epnp.cpp 4.5.5 itself has no planar branch as far as
 *                this restatement's author recalls (its control-point system is inverted with CV_SVD): unverifiable here.
 *                Collinear / coincident sets are reported degenerate (n_models = 0).
 * Products and sums are NOT contracted into FMAs.
 */
#pragma GCC optimize("fp-contract=off")
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* cyclic Jacobi on a symmetric n x n matrix (n <= 12): A is destroyed, V's COLUMNS are the eigenvectors, w the eigenvalues */
static void jacobi_eig(int n, double* A, double* V, double* w) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < n; ++p) {
            diag += A[p * n + p] * A[p * n + p];
            for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {  /* columns p, q */
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {  /* rows p, q */
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

/* The same eigenproblem in the ROUND-ROBIN (tournament) ordering, for the 9 x 9 and 12 x 12 systems: a sweep is n' - 1 rounds
 * (n' = n rounded up to even), a round rotates n' / 2 DISJOINT index pairs — position 0 holds index 0, position j >= 1 holds
 * 1 + ((j - 1 - round) mod (n' - 1)), pair i = the indices at positions i and n' - 1 - i, an index n of an odd system is a bye.
 * The rotations of a round do not touch each other's pivots, so all of them are formed from the matrix the round starts
 * with, then applied in three stages — every pair's column rotation, every pair's row rotation, every pair's eigenvector
 * columns — inside which no element is written twice: the device solver runs a stage on the lanes of a wave
 * (eacham_amd/csrc/solve.hip, jacobi_wave) and this restatement defines the same arithmetic, element by element. The cyclic
 * order above needs 66 dependent rotations per sweep of a 12 x 12 system, this one 11 rounds. */
static void jacobi_eig_rr(int n, double* A, double* V, double* w) {
    const int np = n + (n & 1), half = np / 2;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < n; ++p) {
            diag += A[p * n + p] * A[p * n + p];
            for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int round = 0; round < np - 1; ++round) {
            int P[8], Q[8], on[8];
            double C[8], S[8];
            for (int i = 0; i < half; ++i) {
                const int j1 = i, j2 = np - 1 - i;
                const int a = j1 == 0 ? 0 : 1 + ((j1 - 1 - round) % (np - 1) + (np - 1)) % (np - 1);
                const int b = 1 + ((j2 - 1 - round) % (np - 1) + (np - 1)) % (np - 1);
                P[i] = a < b ? a : b, Q[i] = a < b ? b : a;
                on[i] = 0;
                if (Q[i] >= n) continue;  /* the bye of an odd system */
                const double apq = A[P[i] * n + Q[i]];
                if (apq == 0.0) continue;
                const double theta = (A[Q[i] * n + Q[i]] - A[P[i] * n + P[i]]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                C[i] = 1.0 / sqrt(t * t + 1.0), S[i] = t * C[i];
                on[i] = 1;
            }
            for (int i = 0; i < half; ++i) {  /* columns p, q of every pair */
                if (!on[i]) continue;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + P[i]], akq = A[k * n + Q[i]];
                    A[k * n + P[i]] = C[i] * akp - S[i] * akq;
                    A[k * n + Q[i]] = S[i] * akp + C[i] * akq;
                }
            }
            for (int i = 0; i < half; ++i) {  /* rows p, q of every pair */
                if (!on[i]) continue;
                for (int k = 0; k < n; ++k) {
                    const double apk = A[P[i] * n + k], aqk = A[Q[i] * n + k];
                    A[P[i] * n + k] = C[i] * apk - S[i] * aqk;
                    A[Q[i] * n + k] = S[i] * apk + C[i] * aqk;
                }
            }
            for (int i = 0; i < half; ++i) {  /* the eigenvector columns */
                if (!on[i]) continue;
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + P[i]], vkq = V[k * n + Q[i]];
                    V[k * n + P[i]] = C[i] * vkp - S[i] * vkq;
                    V[k * n + Q[i]] = S[i] * vkp + C[i] * vkq;
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

/* a: 4 x 2 source points, b: 4 x 2 destination points; H: 9 doubles. Returns 1, or 0 for a degenerate sample. */
int oracle_homography4(const double* a, const double* b, double* H) {
    const int count = 4;
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < count; ++i) {
        cM[0] += a[2 * i]; cM[1] += a[2 * i + 1];
        cm[0] += b[2 * i]; cm[1] += b[2 * i + 1];
    }
    for (int k = 0; k < 2; ++k) cM[k] /= count, cm[k] /= count;
    for (int i = 0; i < count; ++i)
        for (int k = 0; k < 2; ++k) {
            sM[k] += fabs(a[2 * i + k] - cM[k]);
            sm[k] += fabs(b[2 * i + k] - cm[k]);
        }
    for (int k = 0; k < 2; ++k)
        if (fabs(sM[k]) < 2.220446049250313e-16 || fabs(sm[k]) < 2.220446049250313e-16) return 0;
    for (int k = 0; k < 2; ++k) sM[k] = count / sM[k], sm[k] = count / sm[k];
    double LtL[81];
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < count; ++i) {
        const double x = (b[2 * i] - cm[0]) * sm[0], y = (b[2 * i + 1] - cm[1]) * sm[1];
        const double X = (a[2 * i] - cM[0]) * sM[0], Y = (a[2 * i + 1] - cM[1]) * sM[1];
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; ++j)
            for (int k = j; k < 9; ++k) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; ++j)
        for (int k = 0; k < j; ++k) LtL[j * 9 + k] = LtL[k * 9 + j];
    double V[81], w[9];
    jacobi_eig_rr(9, LtL, V, w);
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (w[i] < w[best]) best = i;
    double H0[9];
    for (int k = 0; k < 9; ++k) H0[k] = V[k * 9 + best];
    /* H = invHnorm * H0 * Hnorm2, invHnorm = [1/sm.x 0 cm.x; 0 1/sm.y cm.y; 0 0 1], Hnorm2 = [sM.x 0 -cM.x sM.x; 0 sM.y -cM.y sM.y; 0 0 1] */
    const double inv[9] = {1.0 / sm[0], 0, cm[0], 0, 1.0 / sm[1], cm[1], 0, 0, 1};
    const double n2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double T[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[3 * r + c] = inv[3 * r] * H0[c] + inv[3 * r + 1] * H0[3 + c] + inv[3 * r + 2] * H0[6 + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[3 * r + c] = T[3 * r] * n2[c] + T[3 * r + 1] * n2[3 + c] + T[3 * r + 2] * n2[6 + c];
    if (!(fabs(H[8]) > 0.0)) return 0;
    const double s = 1.0 / H[8];
    for (int k = 0; k < 9; ++k) H[k] *= s;
    return 1;
}

/* ---- five-point ---------------------------------------------------------------------------------------------- */
/* column of the monomial x^ex y^ey z^ez (total degree <= 3) in Nister's elimination order */
static int mono_col(int ex, int ey, int ez) {
    static const int8_t order[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                                        {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
    for (int k = 0; k < 20; ++k)
        if (order[k][0] == ex && order[k][1] == ey && order[k][2] == ez) return k;
    return -1;
}
/* row += s * l1 l2 l3, each l a linear form {x, y, z, 1} */
static void mul3acc(const double* l1, const double* l2, const double* l3, double s, double* row) {
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            for (int c = 0; c < 4; ++c) {
                const int ex = (a == 0) + (b == 0) + (c == 0), ey = (a == 1) + (b == 1) + (c == 1), ez = (a == 2) + (b == 2) + (c == 2);
                row[mono_col(ex, ey, ez)] += s * (l1[a] * l2[b]) * l3[c];
            }
}
static void poly_mul(const double* p, int dp, const double* q, int dq, double* out /* dp + dq + 1 */) {
    for (int k = 0; k <= dp + dq; ++k) out[k] = 0.0;
    for (int i = 0; i <= dp; ++i)
        for (int j = 0; j <= dq; ++j) out[i + j] += p[i] * q[j];
}

/* real roots of c[0] + c[1] z + ... + c[10] z^10 (ascending), sorted ascending; returns their number */
static int real_roots10(const double* c, double* roots) {
    int deg = 10;
    double cmax = 0.0;
    for (int k = 0; k <= 10; ++k) cmax = fmax(cmax, fabs(c[k]));
    if (!(cmax > 0.0)) return 0;
    while (deg > 0 && fabs(c[deg]) <= 1e-14 * cmax) --deg;
    if (deg == 0) return 0;
    double m[11];  /* monic */
    for (int k = 0; k <= deg; ++k) m[k] = c[k] / c[deg];
    double bound = 0.0;
    for (int k = 0; k < deg; ++k) bound = fmax(bound, fabs(m[k]));
    bound += 1.0;
    /* Durand-Kerner from the powers of 0.4 + 0.9 i scaled to the geometric mean of the root magnitudes |m0|^(1/deg)
     * (Newton's iteration for the deg-th root: only + - * /, so that every build computes the same bits) */
    double zr[10], zi[10];
    {
        double r0 = 1.0;
        const double a0 = fabs(m[0]);
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
            zr[k] = r0 * cr;
            zi[k] = r0 * ci;
            const double tr = cr * 0.4 - ci * 0.9, ti = cr * 0.9 + ci * 0.4;
            cr = tr, ci = ti;
        }
    }
    /* The simultaneous (Weierstrass) form: every iterate of a sweep is corrected from the iterates the sweep STARTED with, so
     * the ten corrections are independent of each other — the device solver computes them on ten lanes of a wave
     * (eacham_amd/csrc/solve.hip) and this restatement defines the same arithmetic. (Until round 4 both used the
     * Gauss-Seidel form, a correction seeing the corrections before it: one more dependency per root, no better a result.)
     * The sweeps stop at a correction of 1e-11 of the root bound, at most 200 of them: the real roots are polished by Newton's
     * iteration on the real polynomial right below (quadratic: one step from 1e-11 is machine precision), and clustered roots,
     * which converge linearly and never reach 1e-15 — 3.6 % of the samples of a two-view scene ran the old limit of 600 sweeps
     * to its end — are not solutions anybody wants to wait for. */
    for (int it = 0; it < 200; ++it) {
        double change = 0.0, nzr[10], nzi[10];
        for (int k = 0; k < deg; ++k) {
            double pr = 1.0, pi = 0.0;  /* Horner on the monic polynomial */
            for (int j = deg - 1; j >= 0; --j) {
                const double tr = pr * zr[k] - pi * zi[k] + m[j], ti = pr * zi[k] + pi * zr[k];
                pr = tr, pi = ti;
            }
            double dr = 1.0, di = 0.0;
            for (int j = 0; j < deg; ++j)
                if (j != k) {
                    const double ar = zr[k] - zr[j], ai = zi[k] - zi[j];
                    const double tr = dr * ar - di * ai, ti = dr * ai + di * ar;
                    dr = tr, di = ti;
                }
            const double den = dr * dr + di * di;
            nzr[k] = zr[k], nzi[k] = zi[k];
            if (!(den > 0.0)) continue;
            const double qr = (pr * dr + pi * di) / den, qi = (pi * dr - pr * di) / den;
            nzr[k] = zr[k] - qr;
            nzi[k] = zi[k] - qi;
            change = fmax(change, fabs(qr) + fabs(qi));
        }
        for (int k = 0; k < deg; ++k) zr[k] = nzr[k], zi[k] = nzi[k];
        if (change <= 1e-11 * bound) break;
    }
    int n = 0;
    for (int k = 0; k < deg; ++k) {
        if (fabs(zi[k]) > 1e-7 * (1.0 + fabs(zr[k]))) continue;
        double z = zr[k];
        for (int it = 0; it < 4; ++it) {  /* Newton polish on the real polynomial */
            double p = c[deg], d = 0.0;
            for (int j = deg - 1; j >= 0; --j) {
                d = d * z + p;
                p = p * z + c[j];
            }
            if (!(fabs(d) > 0.0)) break;
            z -= p / d;
        }
        roots[n++] = z;
    }
    for (int i = 1; i < n; ++i) {  /* insertion sort */
        const double v = roots[i];
        int j = i - 1;
        while (j >= 0 && roots[j] > v) roots[j + 1] = roots[j], --j;
        roots[j + 1] = v;
    }
    return n;
}

/* p1, p2: 5 x 2 pixels of view 1 / view 2; K = fx fy cx cy (NULL: already normalised). E: 10 x 9. Returns the number of models. */
int oracle_essential5(const double* p1, const double* p2, const double* K, double* E) {
    double Q[9][5];  /* Q^T: column i = the constraint row of correspondence i */
    for (int i = 0; i < 5; ++i) {
        double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        if (K) {
            x1 = (x1 - K[2]) / K[0]; y1 = (y1 - K[3]) / K[1];
            x2 = (x2 - K[2]) / K[0]; y2 = (y2 - K[3]) / K[1];
        }
        const double row[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        for (int k = 0; k < 9; ++k) Q[k][i] = row[k];
    }
    /* Householder QR of Q^T (9 x 5): the last four columns of the orthogonal factor span the null space of Q */
    double P[9][9];
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) P[r][c] = r == c ? 1.0 : 0.0;
    for (int k = 0; k < 5; ++k) {
        double norm = 0.0;
        for (int r = k; r < 9; ++r) norm += Q[r][k] * Q[r][k];
        norm = sqrt(norm);
        if (!(norm > 0.0)) return 0;
        double v[9];
        for (int r = 0; r < 9; ++r) v[r] = r < k ? 0.0 : Q[r][k];
        v[k] += Q[k][k] >= 0.0 ? norm : -norm;
        double vv = 0.0;
        for (int r = k; r < 9; ++r) vv += v[r] * v[r];
        if (!(vv > 0.0)) return 0;
        for (int c = k; c < 5; ++c) {  /* Q <- (I - 2 v v^T / vv) Q */
            double d = 0.0;
            for (int r = k; r < 9; ++r) d += v[r] * Q[r][c];
            d = 2.0 * d / vv;
            for (int r = k; r < 9; ++r) Q[r][c] -= d * v[r];
        }
        for (int r = 0; r < 9; ++r) {  /* P <- P (I - 2 v v^T / vv) */
            double d = 0.0;
            for (int c = k; c < 9; ++c) d += P[r][c] * v[c];
            d = 2.0 * d / vv;
            for (int c = k; c < 9; ++c) P[r][c] -= d * v[c];
        }
    }
    double lin[9][4];  /* entry e of E as a linear form in (x, y, z, 1) */
    for (int e = 0; e < 9; ++e)
        for (int b = 0; b < 4; ++b) lin[e][b] = P[e][5 + b];
    double A[10][20];
    memset(A, 0, sizeof(A));
    {   /* det E */
        static const int perm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
        for (int p = 0; p < 6; ++p)
            mul3acc(lin[perm[p][0]], lin[3 + perm[p][1]], lin[6 + perm[p][2]], p < 3 ? 1.0 : -1.0, A[0]);
    }
    for (int i = 0; i < 3; ++i)      /* 2 E E^T E - tr(E E^T) E */
        for (int j = 0; j < 3; ++j) {
            double* row = A[1 + 3 * i + j];
            for (int k = 0; k < 3; ++k)
                for (int l = 0; l < 3; ++l) {
                    mul3acc(lin[3 * i + l], lin[3 * k + l], lin[3 * k + j], 2.0, row);
                    mul3acc(lin[3 * k + l], lin[3 * k + l], lin[3 * i + j], -1.0, row);
                }
        }
    double A0[10][20];  /* the constraints as assembled: the polish below evaluates them */
    for (int r = 0; r < 10; ++r)
        for (int c = 0; c < 20; ++c) A0[r][c] = A[r][c];
    for (int col = 0; col < 10; ++col) {  /* Gauss-Jordan, partial pivoting */
        int piv = col;
        for (int r = col + 1; r < 10; ++r)
            if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
        if (!(fabs(A[piv][col]) > 1e-300)) return 0;
        if (piv != col)
            for (int c = 0; c < 20; ++c) {
                const double t = A[piv][c];
                A[piv][c] = A[col][c];
                A[col][c] = t;
            }
        const double inv = 1.0 / A[col][col];
        for (int c = 0; c < 20; ++c) A[col][c] *= inv;
        for (int r = 0; r < 10; ++r)
            if (r != col) {
                const double f = A[r][col];
                if (f != 0.0)
                    for (int c = 0; c < 20; ++c) A[r][c] -= f * A[col][c];
            }
    }
    /* B(z): rows k = e - z f, l = g - z h, m = i - z j; entries = polynomials in z (ascending), degrees 3, 3, 4 */
    double B[3][3][5];
    for (int r = 0; r < 3; ++r) {
        const double* e = &A[4 + 2 * r][10];
        const double* f = &A[5 + 2 * r][10];
        B[r][0][0] = e[2];  B[r][0][1] = e[1] - f[2];  B[r][0][2] = e[0] - f[1];  B[r][0][3] = -f[0];  B[r][0][4] = 0.0;
        B[r][1][0] = e[5];  B[r][1][1] = e[4] - f[5];  B[r][1][2] = e[3] - f[4];  B[r][1][3] = -f[3];  B[r][1][4] = 0.0;
        B[r][2][0] = e[9];  B[r][2][1] = e[8] - f[9];  B[r][2][2] = e[7] - f[8];  B[r][2][3] = e[6] - f[7];  B[r][2][4] = -f[6];
    }
    double poly[11];
    for (int k = 0; k <= 10; ++k) poly[k] = 0.0;
    {
        static const int cyc[3][2] = {{1, 2}, {2, 0}, {0, 1}};  /* cofactor expansion along row 0: columns (c1, c2) of rows 1, 2 */
        for (int c0 = 0; c0 < 3; ++c0) {
            const int c1 = cyc[c0][0], c2 = cyc[c0][1];
            const int d1 = c1 == 2 ? 4 : 3, d2 = c2 == 2 ? 4 : 3, d0 = c0 == 2 ? 4 : 3;
            double m1[9], m2[9], minor[9], term[13];
            poly_mul(B[1][c1], d1, B[2][c2], d2, m1);
            poly_mul(B[1][c2], d2, B[2][c1], d1, m2);
            for (int k = 0; k <= d1 + d2; ++k) minor[k] = m1[k] - m2[k];
            poly_mul(B[0][c0], d0, minor, d1 + d2, term);
            for (int k = 0; k <= d0 + d1 + d2 && k <= 10; ++k) poly[k] += term[k];
        }
    }
    double roots[10];
    const int nr = real_roots10(poly, roots);
    int n = 0;
    for (int r = 0; r < nr; ++r) {
        const double z = roots[r];
        double b[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double v = B[i][j][4];
                for (int k = 3; k >= 0; --k) v = v * z + B[i][j][k];
                b[i][j] = v;
            }
        /* [x y 1]^T spans the null space of b: two of its rows, the pair with the largest 2 x 2 determinant */
        static const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        int bp = 0;
        double bd = 0.0;
        for (int p = 0; p < 3; ++p) {
            const double d = b[pr[p][0]][0] * b[pr[p][1]][1] - b[pr[p][0]][1] * b[pr[p][1]][0];
            if (fabs(d) > fabs(bd)) bd = d, bp = p;
        }
        if (!(fabs(bd) > 0.0)) continue;
        const int r0 = pr[bp][0], r1 = pr[bp][1];
        double x = (b[r0][1] * b[r1][2] - b[r0][2] * b[r1][1]) / bd;
        double y = (b[r0][2] * b[r1][0] - b[r0][0] * b[r1][2]) / bd;
        double zz = z;
        /* The root of a degree-10 polynomial carries the conditioning of the whole elimination (1e-4 seen on samples of a
         * short baseline): three Gauss-Newton steps on the ten constraints themselves, in (x, y, z), bring the solution
         * back to the accuracy of the input. (five-point.cpp returns the unpolished root.) */
        for (int it = 0; it < 3; ++it) {
            const double px[4] = {1.0, x, x * x, x * x * x}, py[4] = {1.0, y, y * y, y * y * y}, pz[4] = {1.0, zz, zz * zz, zz * zz * zz};
            double JtJ[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Jtr[3] = {0, 0, 0};
            for (int row = 0; row < 10; ++row) {
                double rv = 0.0, g[3] = {0, 0, 0};
                for (int ex = 0; ex <= 3; ++ex)
                    for (int ey = 0; ex + ey <= 3; ++ey)
                        for (int ez = 0; ex + ey + ez <= 3; ++ez) {
                            const double cf = A0[row][mono_col(ex, ey, ez)];
                            rv += cf * (px[ex] * py[ey]) * pz[ez];
                            if (ex) g[0] += cf * (ex * px[ex - 1] * py[ey]) * pz[ez];
                            if (ey) g[1] += cf * (px[ex] * (ey * py[ey - 1])) * pz[ez];
                            if (ez) g[2] += cf * (px[ex] * py[ey]) * (ez * pz[ez - 1]);
                        }
                for (int u = 0; u < 3; ++u) {
                    Jtr[u] += g[u] * rv;
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
        double Ev[9], nrm = 0.0;
        for (int e = 0; e < 9; ++e) {
            Ev[e] = lin[e][0] * x + lin[e][1] * y + lin[e][2] * zz + lin[e][3];
            nrm += Ev[e] * Ev[e];
        }
        nrm = sqrt(nrm);
        if (!(nrm > 0.0) || !(nrm < 1e300)) continue;
        for (int e = 0; e < 9; ++e) E[9 * n + e] = Ev[e] / nrm;
        ++n;
    }
    return n;
}

/* least squares min |A x - b| for an r x c system (c <= r <= 6, c <= 5), Householder QR on a copy; returns 0 if a column collapses */
static int lsq_small(int r, int c, const double* A, const double* b, double* x) {
    double Q[6 * 6];  /* the r x (c + 1) working array [A | b] */
    for (int i = 0; i < r; ++i) {
        for (int j = 0; j < c; ++j) Q[i * 6 + j] = A[i * c + j];
        Q[i * 6 + c] = b[i];
    }
    for (int j = 0; j < c; ++j) {
        double nrm = 0.0;
        for (int i = j; i < r; ++i) nrm += Q[i * 6 + j] * Q[i * 6 + j];
        nrm = sqrt(nrm);
        if (!(nrm > 0.0)) return 0;
        const double alpha = Q[j * 6 + j] > 0.0 ? -nrm : nrm;
        double v[6];
        for (int i = j; i < r; ++i) v[i] = Q[i * 6 + j];
        v[j] -= alpha;
        double vv = 0.0;
        for (int i = j; i < r; ++i) vv += v[i] * v[i];
        if (!(vv > 0.0)) return 0;
        for (int k = j; k <= c; ++k) {
            double d = 0.0;
            for (int i = j; i < r; ++i) d += v[i] * Q[i * 6 + k];
            d = 2.0 * d / vv;
            for (int i = j; i < r; ++i) Q[i * 6 + k] -= d * v[i];
        }
        Q[j * 6 + j] = alpha;
    }
    for (int j = c - 1; j >= 0; --j) {
        double s = Q[j * 6 + c];
        for (int k = j + 1; k < c; ++k) s -= Q[j * 6 + k] * x[k];
        x[j] = s / Q[j * 6 + j];
    }
    for (int j = 0; j < c; ++j)
        if (!(fabs(x[j]) < 1e300)) return 0;
    return 1;
}

/* EPnP (Lepetit, Moreno-Noguer, Fua 2009) on the m >= 4 points idx[0..m) of obj (n x 3) / img (n x 2), K = fx fy cx cy.
 * Rt = R (row-major) | t of x_cam = R X + t. Every pass over the points recomputes the barycentric coordinates, so the
 * working set does not grow with m (the RANSAC kernel calls it with m = 5, the final refit with all inliers).
 * Coplanar point sets take the three-control-point form (see the header). Returns 1, or 0 for a degenerate point set
 * (collinear / coincident points). */
int oracle_epnp(int m, const int* idx, const double* obj, const double* img, const double* K, double* Rt) {
    if (m < 4) return 0;
    /* Sums over the points: sequential for m <= 64; beyond that W = 64 strided partial sums (points l, l + 64, ... in
     * partial l), added in the order of l — the order a 64-lane wave produces, and for m <= 64 the same bits as the
     * sequential sum (every partial then holds at most one term). */
    const int W = m > 64 ? 64 : 1;
    const double fu = K[0], fv = K[1], uc = K[2], vc = K[3];
    /* control points: centroid + principal axes scaled by the spread along them */
    double c0[3] = {0, 0, 0};
    for (int l = 0; l < W; ++l) {
        double part[3] = {0, 0, 0};
        for (int k = l; k < m; k += W)
            for (int e = 0; e < 3; ++e) part[e] += obj[3 * (size_t)idx[k] + e];
        for (int e = 0; e < 3; ++e) c0[e] = l ? c0[e] + part[e] : part[e];
    }
    for (int e = 0; e < 3; ++e) c0[e] /= (double)m;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, V3[9], w3[3];
    for (int l = 0; l < W; ++l) {
        double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = l; k < m; k += W) {
            double d[3];
            for (int e = 0; e < 3; ++e) d[e] = obj[3 * (size_t)idx[k] + e] - c0[e];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) part[3 * i + j] += d[i] * d[j];
        }
        for (int i = 0; i < 9; ++i) C[i] = l ? C[i] + part[i] : part[i];
    }
    jacobi_eig(3, C, V3, w3);
    double wmax = w3[0] > w3[1] ? w3[0] : w3[1];
    wmax = wmax > w3[2] ? wmax : w3[2];
    double ax[3][3], sc[3];  /* ax[k] = unit axis k, sc[k] = its length: control point k+1 = c0 + sc[k] ax[k] */
    if (!(wmax > 0.0)) return 0;
    int kmin = 0;            /* the axis of the smallest spread (ties: the lower index) */
    for (int k = 1; k < 3; ++k)
        if (w3[k] < w3[kmin]) kmin = k;
    for (int k = 0; k < 3; ++k)
        if (k != kmin && !(w3[k] > 1e-12 * wmax)) return 0;  /* collinear / coincident points */
    const int planar = !(w3[kmin] > 1e-12 * wmax);
    /* a point set with volume keeps the axes as the eigenproblem leaves them; a plane's flat axis goes last and gets length 0:
     * control point 3 then coincides with the centroid and carries barycentric coordinate 0 */
    const int perm[3] = {planar ? (kmin == 0 ? 1 : 0) : 0, planar ? (kmin == 2 ? 1 : 2) : 1, planar ? kmin : 2};
    for (int k = 0; k < 3; ++k) {
        sc[k] = (planar && k == 2) ? 0.0 : sqrt(w3[perm[k]] / (double)m);
        for (int e = 0; e < 3; ++e) ax[k][e] = V3[3 * e + perm[k]];
    }
#define EPNP_ALPHAS(i, al)                                                                        \
    {                                                                                             \
        double d_[3];                                                                             \
        for (int e_ = 0; e_ < 3; ++e_) d_[e_] = obj[3 * (size_t)(i) + e_] - c0[e_];               \
        for (int k_ = 0; k_ < 3; ++k_)                                                            \
            (al)[k_ + 1] = (planar && k_ == 2) ? 0.0 : (ax[k_][0] * d_[0] + ax[k_][1] * d_[1] + ax[k_][2] * d_[2]) / sc[k_]; \
        (al)[0] = 1.0 - (al)[1] - (al)[2] - (al)[3];                                              \
    }
    /* M^T M of the 2m x 12 projection system  sum_j alpha_j (fu Xc_j + (uc - u) Zc_j) = 0, same with v */
    double MtM[144], V[144], w[12];
    for (int i = 0; i < 144; ++i) MtM[i] = 0.0;
    for (int l = 0; l < W; ++l) {
        double part[144];
        for (int i = 0; i < 144; ++i) part[i] = 0.0;
        for (int k = l; k < m; k += W) {
            double al[4], r1[12], r2[12];
            EPNP_ALPHAS(idx[k], al);
            const double du = uc - img[2 * (size_t)idx[k]], dv = vc - img[2 * (size_t)idx[k] + 1];
            for (int j = 0; j < 4; ++j) {
                r1[3 * j] = al[j] * fu, r1[3 * j + 1] = 0.0, r1[3 * j + 2] = al[j] * du;
                r2[3 * j] = 0.0, r2[3 * j + 1] = al[j] * fv, r2[3 * j + 2] = al[j] * dv;
            }
            for (int i = 0; i < 12; ++i)
                for (int j = i; j < 12; ++j) part[12 * i + j] += r1[i] * r1[j] + r2[i] * r2[j];
        }
        for (int i = 0; i < 144; ++i) MtM[i] = l ? MtM[i] + part[i] : part[i];
    }
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < i; ++j) MtM[12 * i + j] = MtM[12 * j + i];
    if (planar) {  /* rows / columns 9..11 are exact zeros: their diagonal goes above every eigenvalue of the 9 x 9 part */
        double tr = 0.0;
        for (int i = 0; i < 9; ++i) tr += MtM[13 * i];
        for (int i = 9; i < 12; ++i) MtM[13 * i] = 2.0 * tr + 1.0;
    }
    jacobi_eig_rr(12, MtM, V, w);
    int ord[4];  /* the four smallest eigenvalues, ascending (ties: lower index first) */
    for (int k = 0; k < 4; ++k) {
        int best = -1;
        for (int i = 0; i < 12; ++i) {
            int used = 0;
            for (int q = 0; q < k; ++q) used |= ord[q] == i;
            if (!used && (best < 0 || w[i] < w[best])) best = i;
        }
        ord[k] = best;
    }
    double ev[4][12];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 12; ++i) ev[k][i] = V[12 * i + ord[k]];
    /* the six control-point distance constraints, quadratic in beta: L (6 x 10) over [b00 b01 b11 b02 b12 b22 b03 b13 b23 b33] */
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double cw[4][3];
    for (int e = 0; e < 3; ++e) {
        cw[0][e] = c0[e];
        for (int k = 0; k < 3; ++k) cw[k + 1][e] = c0[e] + sc[k] * ax[k][e];
    }
    double L[6][10], rho[6];
    for (int p = 0; p < 6; ++p) {
        double dv[4][3];
        for (int k = 0; k < 4; ++k)
            for (int e = 0; e < 3; ++e) dv[k][e] = ev[k][3 * pa[p] + e] - ev[k][3 * pb[p] + e];
        int col = 0;
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i <= j; ++i) {
                const double d = dv[i][0] * dv[j][0] + dv[i][1] * dv[j][1] + dv[i][2] * dv[j][2];
                L[p][col++] = i == j ? d : 2.0 * d;
            }
        rho[p] = 0.0;
        for (int e = 0; e < 3; ++e) rho[p] += (cw[pa[p]][e] - cw[pb[p]][e]) * (cw[pa[p]][e] - cw[pb[p]][e]);
    }
    double best_err = -1.0;
    for (int variant = 0; variant < 3; ++variant) {
        /* linearised start: the products b_i b_j that involve only the first 1 / 2 / 3 null vectors' leading terms */
        const int ncol[3] = {4, 3, 5};
        const int cols[3][5] = {{0, 1, 3, 6, 0}, {0, 1, 2, 0, 0}, {0, 1, 2, 3, 4}};
        double A[30], x[5], beta[4] = {0, 0, 0, 0};
        if (planar) {
            /* three control points: the distance equations of the pairs (0,1), (0,2), (1,2) = rows 0, 1, 3; start 0 takes the first
             * null vector alone (x = b00), start 1 the first two (x = b00 b01 b11, a square system); there is no third start.
             * Gauss-Newton runs on the unknowns of the start (1 or 2 betas against three equations). */
            if (variant == 2) continue;
            const int rows3[3] = {0, 1, 3}, nb = variant + 1, nc3 = variant == 0 ? 1 : 3;
            double rho3[3];
            for (int p = 0; p < 3; ++p) {
                rho3[p] = rho[rows3[p]];
                for (int j = 0; j < nc3; ++j) A[p * nc3 + j] = L[rows3[p]][j];
            }
            if (!lsq_small(3, nc3, A, rho3, x)) continue;
            const double s = x[0] < 0.0 ? -1.0 : 1.0;
            beta[0] = sqrt(s * x[0]);
            if (variant == 1) {
                beta[1] = s * x[2] > 0.0 ? sqrt(s * x[2]) : 0.0;
                if (x[1] < 0.0) beta[0] = -beta[0];
            }
            if (!(beta[0] != 0.0)) continue;
            for (int it = 0; it < 5; ++it) {
                double J[6], r[3], dx[2];
                for (int p = 0; p < 3; ++p) {
                    const double* l = L[rows3[p]];
                    J[nb * p] = 2.0 * l[0] * beta[0] + l[1] * beta[1];
                    if (nb == 2) J[nb * p + 1] = l[1] * beta[0] + 2.0 * l[2] * beta[1];
                    r[p] = rho3[p] - (l[0] * beta[0] * beta[0] + l[1] * beta[0] * beta[1] + l[2] * beta[1] * beta[1]);
                }
                if (!lsq_small(3, nb, J, r, dx)) break;
                for (int k = 0; k < nb; ++k) beta[k] += dx[k];
            }
        } else {
            for (int p = 0; p < 6; ++p)
                for (int j = 0; j < ncol[variant]; ++j) A[p * ncol[variant] + j] = L[p][cols[variant][j]];
            if (!lsq_small(6, ncol[variant], A, rho, x)) continue;
            if (variant == 0) {  /* x = b00 b01 b02 b03 */
                const double s = x[0] < 0.0 ? -1.0 : 1.0;
                beta[0] = sqrt(s * x[0]);
                if (!(beta[0] > 0.0)) continue;
                for (int k = 1; k < 4; ++k) beta[k] = s * x[k] / beta[0];
            } else {             /* x = b00 b01 b11 (b02 b12) */
                const double s = x[0] < 0.0 ? -1.0 : 1.0;
                beta[0] = sqrt(s * x[0]);
                beta[1] = s * x[2] > 0.0 ? sqrt(s * x[2]) : 0.0;
                if (x[1] < 0.0) beta[0] = -beta[0];
                if (!(beta[0] != 0.0)) continue;
                if (variant == 2) beta[2] = x[3] / beta[0];
            }
            for (int it = 0; it < 5; ++it) {  /* Gauss-Newton on the six distance equations */
                double J[24], r[6], dx[4];
                for (int p = 0; p < 6; ++p) {
                    const double* l = L[p];
                    J[4 * p + 0] = 2.0 * l[0] * beta[0] + l[1] * beta[1] + l[3] * beta[2] + l[6] * beta[3];
                    J[4 * p + 1] = l[1] * beta[0] + 2.0 * l[2] * beta[1] + l[4] * beta[2] + l[7] * beta[3];
                    J[4 * p + 2] = l[3] * beta[0] + l[4] * beta[1] + 2.0 * l[5] * beta[2] + l[8] * beta[3];
                    J[4 * p + 3] = l[6] * beta[0] + l[7] * beta[1] + l[8] * beta[2] + 2.0 * l[9] * beta[3];
                    r[p] = rho[p] - (l[0] * beta[0] * beta[0] + l[1] * beta[0] * beta[1] + l[2] * beta[1] * beta[1] + l[3] * beta[0] * beta[2] +
                                     l[4] * beta[1] * beta[2] + l[5] * beta[2] * beta[2] + l[6] * beta[0] * beta[3] + l[7] * beta[1] * beta[3] +
                                     l[8] * beta[2] * beta[3] + l[9] * beta[3] * beta[3]);
                }
                if (!lsq_small(6, 4, J, r, dx)) break;
                for (int k = 0; k < 4; ++k) beta[k] += dx[k];
            }
        }
        /* control points in the camera frame, sign from the first point's depth */
        double cc[4][3];
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 3; ++e) cc[j][e] = beta[0] * ev[0][3 * j + e] + beta[1] * ev[1][3 * j + e] + beta[2] * ev[2][3 * j + e] + beta[3] * ev[3][3 * j + e];
        {
            double al[4];
            EPNP_ALPHAS(idx[0], al);
            const double z0 = al[0] * cc[0][2] + al[1] * cc[1][2] + al[2] * cc[2][2] + al[3] * cc[3][2];
            if (z0 < 0.0)
                for (int j = 0; j < 4; ++j)
                    for (int e = 0; e < 3; ++e) cc[j][e] = -cc[j][e];
        }
        /* absolute orientation world -> camera (Horn's quaternion form): S = sum pc (pw - c0)^T */
        double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pcm[3] = {0, 0, 0};
        for (int l = 0; l < W; ++l) {
            double Sp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pp[3] = {0, 0, 0};
            for (int k = l; k < m; k += W) {
                double al[4], pc[3];
                EPNP_ALPHAS(idx[k], al);
                for (int e = 0; e < 3; ++e) {
                    pc[e] = al[0] * cc[0][e] + al[1] * cc[1][e] + al[2] * cc[2][e] + al[3] * cc[3][e];
                    pp[e] += pc[e];
                }
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) Sp[3 * i + j] += pc[i] * (obj[3 * (size_t)idx[k] + j] - c0[j]);
            }
            for (int i = 0; i < 9; ++i) S[i] = l ? S[i] + Sp[i] : Sp[i];
            for (int e = 0; e < 3; ++e) pcm[e] = l ? pcm[e] + pp[e] : pp[e];
        }
        for (int e = 0; e < 3; ++e) pcm[e] /= (double)m;
        /* S[i][j] = sum camera_i world_j; the rotation maximising tr(R^T S) is the top eigenvector of Horn's 4 x 4 matrix
         * written for the map world -> camera (its "left" set is the world points: Sxy = sum world_x camera_y = S[y][x]) */
        const double Sxx = S[0], Sxy = S[3], Sxz = S[6], Syx = S[1], Syy = S[4], Syz = S[7], Szx = S[2], Szy = S[5], Szz = S[8];
        double N[16] = {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx,
                        Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz,
                        Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy,
                        Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz};
        double V4[16], w4[4];
        jacobi_eig(4, N, V4, w4);
        int top = 0;
        for (int k = 1; k < 4; ++k)
            if (w4[k] > w4[top]) top = k;
        const double q0 = V4[top], qx = V4[4 + top], qy = V4[8 + top], qz = V4[12 + top];
        double cand[12];
        cand[0] = q0 * q0 + qx * qx - qy * qy - qz * qz, cand[1] = 2.0 * (qx * qy - q0 * qz), cand[2] = 2.0 * (qx * qz + q0 * qy);
        cand[3] = 2.0 * (qy * qx + q0 * qz), cand[4] = q0 * q0 - qx * qx + qy * qy - qz * qz, cand[5] = 2.0 * (qy * qz - q0 * qx);
        cand[6] = 2.0 * (qz * qx - q0 * qy), cand[7] = 2.0 * (qz * qy + q0 * qx), cand[8] = q0 * q0 - qx * qx - qy * qy + qz * qz;
        for (int i = 0; i < 3; ++i) cand[9 + i] = pcm[i] - (cand[3 * i] * c0[0] + cand[3 * i + 1] * c0[1] + cand[3 * i + 2] * c0[2]);
        double err = 0.0;
        for (int l = 0; l < W; ++l) {
            double ep = 0.0;
            for (int k = l; k < m; k += W) {
                const double* X = obj + 3 * (size_t)idx[k];
                const double xc = cand[0] * X[0] + cand[1] * X[1] + cand[2] * X[2] + cand[9];
                const double yc = cand[3] * X[0] + cand[4] * X[1] + cand[5] * X[2] + cand[10];
                const double zc = cand[6] * X[0] + cand[7] * X[1] + cand[8] * X[2] + cand[11];
                const double eu = uc + fu * xc / zc - img[2 * (size_t)idx[k]], evv = vc + fv * yc / zc - img[2 * (size_t)idx[k] + 1];
                ep += sqrt(eu * eu + evv * evv);
            }
            err = l ? err + ep : ep;
        }
        if (!(err < 1e300)) continue;
        if (best_err < 0.0 || err < best_err) {
            best_err = err;
            for (int i = 0; i < 12; ++i) Rt[i] = cand[i];
        }
    }
#undef EPNP_ALPHAS
    return best_err >= 0.0 ? 1 : 0;
}

/* batch form: n_samples index lists of sample_size points each -> one pose (R row-major | t) per sample, n_models[s] = 0 / 1 */
void oracle_solve_pnp(const double* obj, const double* img, const double* K, int sample_size, int n_samples, const int32_t* idx,
                      double* models, int32_t* n_models) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int s = 0; s < n_samples; ++s) {
        double* out = models + 12 * (size_t)s;
        for (int k = 0; k < 12; ++k) out[k] = 0.0;
        n_models[s] = oracle_epnp(sample_size, idx + (size_t)s * sample_size, obj, img, K, out);
        if (!n_models[s])
            for (int k = 0; k < 12; ++k) out[k] = 0.0;
    }
}

/* batch form with caller-supplied sample indices: kind 0 = homography4 (a -> b), kind 1 = essential5 (a = view 1, b = view 2).
 * models: n_samples x max_models x 9 (max_models = 1 / 10), n_models per sample. */
void oracle_solve_minimal(int kind, const double* a, const double* b, const double* K, int n_samples, const int32_t* idx,
                          double* models, int32_t* n_models) {
    const int m = kind == 0 ? 4 : 5, maxm = kind == 0 ? 1 : 10;
#pragma omp parallel for schedule(dynamic, 16)
    for (int s = 0; s < n_samples; ++s) {
        double pa[10], pb[10];
        for (int k = 0; k < m; ++k) {
            pa[2 * k] = a[2 * idx[s * m + k]]; pa[2 * k + 1] = a[2 * idx[s * m + k] + 1];
            pb[2 * k] = b[2 * idx[s * m + k]]; pb[2 * k + 1] = b[2 * idx[s * m + k] + 1];
        }
        double* out = models + (size_t)s * maxm * 9;
        for (int k = 0; k < maxm * 9; ++k) out[k] = 0.0;
        n_models[s] = kind == 0 ? oracle_homography4(pa, pb, out) : oracle_essential5(pa, pb, K, out);
    }
}
