// ba_window.hpp — the structure of the DENSE form of the Schur stage (round 5), pure C++ (held against the oracle and the pair-list
// form by tests/test_ba_gpu.py::test_the_dense_form_of_a_local_window_solves_the_same_system, which also walks the structure).
//
// What it serves: the per-frame RefineBA of a local window (apps/sfm/main.cpp:207 -> modules/sfm/reconstruction/BundleAdjuster.cpp:
// 123-145: the current frame and its factor neighbours, ~20 cameras, ~10 k observations). There the reduced camera system has
// (nc + 1)(nc + 2) / 2 <= 325 blocks of 6x6 — the calibration columns + right-hand side act as one more camera, as in ba_groups.hpp —
// few enough that a workgroup which owns a group of landmarks can form its share of EVERY block and write one dense partial:
// nothing is enumerated, sorted or uploaded per block. Which landmarks of the group contribute to block (c1, c2) is the AND of two
// 64-bit masks the kernel builds itself (a group has at most 64 landmarks). The structure is the landmark-ordered rows and nothing
// else, a few microseconds of host loops, where the pair lists of rounds 1-4 and the entry lists of ba_groups.hpp cost 0.1 /
// 0.3-0.6 ms per window — more than what the Levenberg-Marquardt tries of a window save.
//
// Definition:
//  1. landmarks with at least one observation, in index order; landmark k owns m_k + 1 consecutive ROWS: its observations in
//     ascending CAMERA order (a landmark seen twice by one camera: not built, the pair lists serve), then its own row;
//  2. groups are filled greedily in that order: a group holds <= rows rows and <= rows / 4 landmarks (a landmark costs max(m + 1, 4));
//  3. per-group arrays are padded to those bounds (row r of group g at g rows + r, landmark t at g rows / 4 + t), as in ba_groups.hpp:
//     rowinfo = {camera (nc: a landmark's own row, -1: no row), landmark index inside the group}, uv, lmid (-1 beyond the group's
//     landmarks), lmrow = local row of the landmark's own row (its first row = the previous landmark's own row + 1).
// A partial (ba_schur_dense -> ba_assemble_dense) is win_stride(nc) doubles: block (c1 <= c2) at 36 (c2 (c2 + 1) / 2 + c1), element
// (a, b) = a of c1, b of c2; then per camera 12 doubles (the Hessian's own diagonal of the camera block and its gradient: what the
// damping and the linearised cost change need apart from the Schur terms); then WIN_KK = 50: the upper triangle of the calibration
// Hessian and its gradient (20), the calibration block's Schur sums (30) — block (nc, nc) of the partial itself stays unused.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "ba_groups.hpp"

namespace eacham {

constexpr int WIN_NC_MAX = 24;       // cameras: 325 blocks = 94 KB per partial
constexpr int WIN_ROW = 42;          // doubles per row in LDS: Et (6x3) | Ap^T (6x2) | Q^T (6x2)
constexpr int WIN_ROWS_MAX = 256;    // one row per thread
constexpr int WIN_GROUPS = 64;       // groups aimed at
constexpr size_t WIN_PART_MAX = (size_t)12 << 20;  // bytes of partials beyond which the form is not used

GRP_HD inline int win_nblk(int nc) { return (nc + 1) * (nc + 2) / 2; }
GRP_HD inline int win_stride(int nc) { return (win_nblk(nc) * 36 + nc * 12 + 50 + 1) & ~1; }
constexpr int WIN_KK = 50;           // per-group calibration part of a partial: Hessian upper triangle (15) + gradient (5) | sum EKt EKt^T (25) + sum EKt gt (5)
// visits (block, landmark) a group can have: a landmark with m + 1 rows has (m + 1)(m + 2) / 2 - 1 <= (m + 1)(nc + 2) / 2 of them
GRP_HD inline int win_visits_max(int nc, int rows) { return rows * (nc + 2) / 2; }
constexpr int WIN_TPB = 512;          // threads of ba_schur_dense: eight waves, two per SIMD (a lone wave issues an fp64 operation every ~8 cycles)
constexpr int WIN_LANE_GROUPS = 21 * (WIN_TPB / 64);  // lane groups of phase B (three lanes each, 21 per wave)
constexpr int WIN_PIECE = 8;         // visits of a long block one lane group sums at most ...
constexpr int WIN_PIECES = 4;        // ... in at most this many pieces, added up inside the workgroup
// LDS of ba_schur_dense: rows | Linv (6) + point (3) per landmark | a landmark mask per camera | the row of (camera, landmark) |
// the visit list (u32) | first visit of a lane group (u16), pieces of a long block (u8) | the pieces' sums (42 doubles each) |
// block-sum scratch
GRP_HD inline size_t win_lds_bytes(int nc, int rows) {
    return sizeof(double) * ((size_t)WIN_ROW * rows + (size_t)9 * (rows / 4) + (size_t)(nc + 1) + (size_t)((nc + 1) * (rows / 4) + 7) / 8 +
                             (size_t)(win_visits_max(nc, rows) + 1) / 2 + (size_t)(2 * (WIN_LANE_GROUPS + 2) + 2 * nc + 7) / 8 +
                             (size_t)42 * WIN_PIECES * 2 * nc + 4 * (WIN_TPB / 64) * WIN_KK + 8);
}
// rows per group: ~WIN_GROUPS groups, what LDS allows, one row per thread
inline int win_rows_for(long long total_rows, int nc) {
    int r = (int)((total_rows + WIN_GROUPS - 1) / WIN_GROUPS);
    r = std::max(64, (r + 3) & ~3);
    r = std::min(r, WIN_ROWS_MAX);
    while (r >= 64 && win_lds_bytes(nc, r) > (size_t)160 * 1024) r -= 4;
    return r >= 64 ? r : 0;
}

struct BaWinGroup { int nlm, nrows; };
struct BaWin {
    int rows = 0;          // rows per group (0: not built)
    int n_rows = 0;        // rows in use
    std::vector<BaWinGroup> groups;
    std::vector<GrpI2> rowinfo;   // [groups x rows]
    std::vector<double> uv;       // [2 x groups x rows]
    std::vector<int> lmid, lmrow; // [groups x rows / 4]
};

inline bool build_window(int nc, int nl, const int* lm_ptr, const unsigned* obs_cam, const double* obs_uv, BaWin& out, int rows_override = 0) {
    out = BaWin();
    if (nc < 1 || nc > WIN_NC_MAX) return false;
    long long total_rows = 0;
    for (int j = 0; j < nl; ++j)
        if (lm_ptr[j + 1] > lm_ptr[j]) total_rows += lm_ptr[j + 1] - lm_ptr[j] + 1;
    if (total_rows == 0) return false;
    const int rows = rows_override > 0 ? rows_override : win_rows_for(total_rows, nc);
    if (rows < 64 || rows % 4 != 0 || rows > WIN_ROWS_MAX || win_lds_bytes(nc, rows) > (size_t)160 * 1024) return false;
    const int lmax = rows / 4;
    // groups, greedily
    std::vector<int> first_lm;  // first landmark (index into `used`) of every group
    std::vector<int> used;
    used.reserve(nl);
    {
        int r = 0, t = 0;
        for (int j = 0; j < nl; ++j) {
            const int m = lm_ptr[j + 1] - lm_ptr[j];
            if (m == 0) continue;
            if (m + 1 > rows) return false;
            if (used.empty() || r + m + 1 > rows || t + 1 > lmax) {
                first_lm.push_back((int)used.size());
                r = 0, t = 0;
            }
            used.push_back(j);
            r += m + 1, ++t;
        }
    }
    const int ng = (int)first_lm.size();
    if ((size_t)ng * win_stride(nc) * sizeof(double) > WIN_PART_MAX) return false;
    first_lm.push_back((int)used.size());
    out.groups.assign(ng, BaWinGroup{0, 0});
    out.rowinfo.assign((size_t)ng * rows, GrpI2{-1, -1});
    out.uv.assign(2 * (size_t)ng * rows, 0.0);
    out.lmid.assign((size_t)ng * lmax, -1);
    out.lmrow.assign((size_t)ng * lmax, 0);
    int idx[WIN_NC_MAX + 1];
    for (int g = 0; g < ng; ++g) {
        int r = 0;
        for (int k = first_lm[g]; k < first_lm[g + 1]; ++k) {
            const int t = k - first_lm[g], j = used[k], a0 = lm_ptr[j], m = lm_ptr[j + 1] - a0;
            if (m > WIN_NC_MAX) return (out = BaWin(), false);  // (more observations than cameras: some camera twice)
            for (int i = 0; i < m; ++i) idx[i] = i;
            std::sort(idx, idx + m, [&](int a, int b) { return obs_cam[a0 + a] < obs_cam[a0 + b]; });
            const size_t base = (size_t)g * rows + r;
            for (int i = 0; i < m; ++i) {
                const int o = a0 + idx[i];
                if (i > 0 && obs_cam[o] == obs_cam[a0 + idx[i - 1]]) return (out = BaWin(), false);
                out.rowinfo[base + i] = GrpI2{(int)obs_cam[o], t};
                out.uv[2 * (base + i)] = obs_uv[2 * (size_t)o];
                out.uv[2 * (base + i) + 1] = obs_uv[2 * (size_t)o + 1];
            }
            out.rowinfo[base + m] = GrpI2{nc, t};
            out.lmid[(size_t)g * lmax + t] = j;
            out.lmrow[(size_t)g * lmax + t] = r + m;
            r += m + 1;
        }
        out.groups[g] = BaWinGroup{first_lm[g + 1] - first_lm[g], r};
        out.n_rows += r;
    }
    out.rows = rows;
    return true;
}

}  // namespace eacham
