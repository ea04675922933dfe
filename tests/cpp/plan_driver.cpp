// plan_driver.cpp — CPU check of eacham_amd/csrc/ba_plan.hpp (the host-side analysis of the sparse reduced-system solve).
// Reads   nc  n_edges  ordering  seed   and n_edges camera pairs from stdin, builds the plan, checks its invariants, then
// EXECUTES the schedule on a random SPD matrix of that block pattern in plain double arithmetic — leaf factors, level
// updates in item order, the raw-tile back-substitution with the right-hand side carried as row 63 of the root panel —
// exactly the data flow of sp_diag / sp_level / sp_backsolve in ba.hip, and compares the step with a dense Cholesky
// solve. Prints one JSON line. Test infrastructure (tests/test_ba_plan.py); nothing here runs on the product path.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#include "../../eacham_amd/csrc/ba_plan.hpp"

using namespace eacham;
typedef std::vector<double> Mat;  // 64 x 64 row-major

static bool chol_inverse(const Mat& A, Mat& W) {  // W = L^-1, A = L L^T (lower triangle of A read)
    const int n = 64;
    Mat L(n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0.0)) return false;
        L[j * n + j] = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double v = A[i * n + j];
            for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v / L[j * n + j];
        }
    }
    W.assign(n * n, 0.0);
    for (int c = 0; c < n; ++c)
        for (int i = c; i < n; ++i) {
            double v = i == c ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) v -= L[i * n + k] * W[k * n + c];
            W[i * n + c] = v / L[i * n + i];
        }
    return true;
}

int main() {
    int nc, ne, ordering;
    unsigned seed;
    if (scanf("%d %d %d %u", &nc, &ne, &ordering, &seed) != 4) return 2;
    std::vector<std::pair<int, int>> edges(ne);
    for (auto& e : edges)
        if (scanf("%d %d", &e.first, &e.second) != 2) return 2;
    BaPlan P;
    build_ba_plan(nc, edges, ordering, P);
    const int np = P.npan, n = 6 * nc + 5, root = np - 1;
    // ---- invariants ----
    int bad = 0;
    {
        std::vector<char> seen((size_t)np * 64, 0);
        for (int c = 0; c < nc; ++c)
            for (int a = 0; a < 6; ++a) {
                const int q = P.pos[c] + a;
                if (q < 0 || q >= np * 64 || seen[q] || P.col_dest[q] != 6 * c + a) ++bad;
                else seen[q] = 1;
            }
        if (P.posK / 64 != root || P.posK % 64 + 5 > 63 || P.rhs_row != np * 64 - 1) ++bad;
        for (const auto& e : edges) {  // every camera block has its tile(s)
            for (int a : {0, 5})
                for (int b : {0, 5}) {
                    int r = P.pos[e.first] + a, q = P.pos[e.second] + b;
                    if (r < q) std::swap(r, q);
                    if (P.tile(r / 64, q / 64) < 0) ++bad;
                }
        }
        for (int J = 0; J < root; ++J)
            if (P.tile(root, J) < 0 || P.parent[J] < 0 || P.level[P.parent[J]] <= P.level[J]) ++bad;
        // every (J, I1 >= I2) of the structure is applied exactly once, in a launch of its window: not before its source
        // exists (level(J)), not after the launch before its target's column is read or factorised (level(I2) - 1), and a
        // target's sources in ascending (level, panel) order over the launches
        std::set<std::vector<int>> done;
        int nnz_off = 0;
        for (int J = 0; J < np; ++J) nnz_off += (int)P.strct[J].size();
        if (P.ntiles != np + nnz_off + P.n_shadow) ++bad;
        for (size_t l = 0; l < P.launches.size(); ++l)
            for (int i = P.launches[l].first; i < P.launches[l].first + P.launches[l].count; ++i) {
                const BaPlanItem& it = P.items[i];
                if (((it.flags & 2) != 0) != (i - P.launches[l].first < P.launches[l].n_first)) ++bad;  // factorising items first
                const bool main_acc = it.tgt == P.tile(it.panel, it.col);
                if (((it.flags & 2) != 0) != (main_acc && it.panel == it.col && (int)l == P.level[it.col] - 1)) ++bad;
                if (!main_acc && (it.tgt < np + nnz_off || (int)l >= P.level[it.col] - 1)) ++bad;  // a shadow works before the deadline only
                if (it.nshadow && (!main_acc || (int)l != P.level[it.col] - 1)) ++bad;               // and is folded in at the deadline
                for (int s = it.src0; s < it.src0 + it.nsrc; ++s) {
                    if (P.level[P.srcs[s].J] > (int)l || (int)l > P.level[it.col] - 1) ++bad;
                    if (P.srcs[s].tile_i != P.tile(it.panel, P.srcs[s].J) || P.srcs[s].tile_j != P.tile(it.col, P.srcs[s].J)) ++bad;
                    if (!done.insert({P.srcs[s].J, P.tile(it.panel, it.col)}).second) ++bad;
                }
            }
        if ((long long)done.size() != P.tile_updates) ++bad;
        long long want = 0;
        for (int J = 0; J < np; ++J) want += (long long)P.strct[J].size() * (P.strct[J].size() + 1) / 2;
        if (want != P.tile_updates) ++bad;
    }
    // ---- a random SPD matrix of the pattern, in the caller's order ----
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> N01;
    std::vector<double> S((size_t)n * n, 0.0), g(n);
    auto add_block = [&](int r0, int nr_, int c0, int ncol) {
        for (int a = 0; a < nr_; ++a)
            for (int b = 0; b < ncol; ++b) {
                const double v = 0.3 * N01(rng);
                S[(size_t)(r0 + a) * n + c0 + b] += v;
                S[(size_t)(c0 + b) * n + r0 + a] += v;
            }
    };
    for (const auto& e : edges) add_block(6 * e.first, 6, 6 * e.second, 6);
    for (int c = 0; c < nc; ++c) add_block(6 * c, 6, 6 * nc, 5);
    for (int i = 0; i < n; ++i) {  // diagonally dominant
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += std::fabs(S[(size_t)i * n + j]);
        S[(size_t)i * n + i] = s + 1.0 + std::fabs(N01(rng));
    }
    for (auto& v : g) v = N01(rng);
    // dense reference solve
    std::vector<double> x(n);
    {
        std::vector<double> L(S);
        for (int j = 0; j < n; ++j) {
            double d = L[(size_t)j * n + j];
            for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
            L[(size_t)j * n + j] = std::sqrt(d);
            for (int i = j + 1; i < n; ++i) {
                double v = L[(size_t)i * n + j];
                for (int k = 0; k < j; ++k) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
                L[(size_t)i * n + j] = v / L[(size_t)j * n + j];
            }
        }
        std::vector<double> y(g);
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < i; ++k) y[i] -= L[(size_t)i * n + k] * y[k];
            y[i] /= L[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            for (int k = i + 1; k < n; ++k) y[i] -= L[(size_t)k * n + i] * y[k];
            y[i] /= L[(size_t)i * n + i];
        }
        x = y;
    }
    // ---- assemble into tiles as ba_assemble does ----
    std::vector<Mat> T(P.ntiles, Mat(64 * 64, 0.0));
    auto store = [&](int r, int q, double v) {
        const int I = r >> 6, J = q >> 6;
        if (I < J) return;
        const int t = P.tile(I, J);
        if (t < 0) { ++bad; return; }
        T[t][(r & 63) * 64 + (q & 63)] = v;
    };
    std::vector<int> colpos(n);
    for (int c = 0; c < nc; ++c)
        for (int a = 0; a < 6; ++a) colpos[6 * c + a] = P.pos[c] + a;
    for (int k = 0; k < 5; ++k) colpos[6 * nc + k] = P.posK + k;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (S[(size_t)i * n + j] != 0.0) store(colpos[i], colpos[j], S[(size_t)i * n + j]);
    for (int i = 0; i < n; ++i) store(P.rhs_row, colpos[i], g[i]);
    for (int q : P.pad_cols) store(q, q, 1.0);
    store(P.rhs_row, P.rhs_row, 1e100);
    // ---- execute the schedule ----
    std::vector<Mat> W(np);
    bool ok = true;
    for (int J : P.leaves) ok = chol_inverse(T[P.diag_tile[J]], W[J]) && ok;
    auto strip = [&](const Mat& A, const Mat& Wj, Mat& Lm) {  // L = A W^T
        Lm.assign(64 * 64, 0.0);
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double s = 0.0;
                for (int k = 0; k <= j; ++k) s += A[i * 64 + k] * Wj[j * 64 + k];
                Lm[i * 64 + j] = s;
            }
    };
    for (const auto& la : P.launches)
        for (int i = la.first; i < la.first + la.count; ++i) {
            const BaPlanItem& it = P.items[i];
            Mat& tgt = T[it.tgt];
            for (int k = 0; k < it.nshadow; ++k)
                for (int e = 0; e < 64 * 64; ++e) tgt[e] += T[it.shadow0 + k][e];
            for (int s = it.src0; s < it.src0 + it.nsrc; ++s) {
                const BaPlanSrc& sr = P.srcs[s];
                if (W[sr.J].empty()) { ++bad; continue; }  // a source whose factor does not exist yet
                Mat Li, Lj;
                strip(T[sr.tile_i], W[sr.J], Li);
                strip(T[sr.tile_j], W[sr.J], Lj);
                for (int a = 0; a < 64; ++a)
                    for (int b = 0; b < 64; ++b) {
                        double v = 0.0;
                        for (int k = 0; k < 64; ++k) v += Li[a * 64 + k] * Lj[b * 64 + k];
                        tgt[a * 64 + b] -= v;
                    }
            }
            if (it.flags & 2) ok = chol_inverse(tgt, W[it.panel]) && ok;
        }
    for (int J = 0; J < np; ++J)
        if (W[J].empty()) ++bad;
    // ---- back-substitution on the raw tiles ----
    std::vector<double> z((size_t)np * 64, 0.0);
    std::vector<char> have(np, 0);
    if (!bad && ok) {
        for (int b = 0; b < np; ++b) {
            const int J = P.bs_order[b];
            double* zj = &z[(size_t)J * 64];
            if (J == root) {
                for (int i = 0; i < 64; ++i) zj[i] = -W[J][63 * 64 + i] / W[J][63 * 64 + 63];
            } else {
                if (!have[P.parent[J]]) ++bad;
                double t[64], u[64];
                for (int j = 0; j < 64; ++j) t[j] = 0.0;
                for (int e = P.bs_ptr[b]; e < P.bs_ptr[b + 1]; ++e) {
                    const Mat& A = T[P.bs_ent[e].first];
                    const double* zi = &z[(size_t)P.bs_ent[e].second * 64];
                    if (!have[P.bs_ent[e].second]) ++bad;
                    for (int r = 0; r < 64; ++r)
                        for (int j = 0; j < 64; ++j) t[j] -= A[r * 64 + j] * zi[r];
                }
                for (int i = 0; i < 64; ++i) {
                    u[i] = 0.0;
                    for (int k = 0; k <= i; ++k) u[i] += W[J][i * 64 + k] * t[k];
                }
                for (int j = 0; j < 64; ++j) {
                    zj[j] = 0.0;
                    for (int i = j; i < 64; ++i) zj[j] += W[J][i * 64 + j] * u[i];
                }
            }
            have[J] = 1;
        }
    }
    double err = 0.0, ref = 0.0;
    for (int q = 0; q < np * 64; ++q)
        if (P.col_dest[q] >= 0) {
            err = std::max(err, std::fabs(z[q] - x[P.col_dest[q]]));
            ref = std::max(ref, std::fabs(x[P.col_dest[q]]));
        }
    int max_final_src = 0, max_src = 0;
    for (const auto& it : P.items) {
        if (it.flags & 2) max_final_src = std::max(max_final_src, it.nsrc);
        max_src = std::max(max_src, it.nsrc);
    }
    printf("{\"nc\": %d, \"npan\": %d, \"ntiles\": %d, \"levels\": %d, \"ordering\": %d, \"nd_leaf\": %d, \"est_us\": %.1f, "
           "\"tile_updates\": %lld, \"shadows\": %d, \"launch_items\": [",
           nc, np, P.ntiles, P.n_levels, P.ordering, P.nd_leaf, P.est_us, P.tile_updates, P.n_shadow);
    for (size_t l = 0; l < P.launches.size(); ++l) printf("%s%d", l ? ", " : "", P.launches[l].count);
    printf("], \"max_final_src\": %d, \"max_src\": %d, \"bad\": %d, \"spd\": %d, \"rel_err\": %.3e}\n", max_final_src, max_src, bad, ok ? 1 : 0, ref > 0 ? err / ref : err);
    return bad ? 1 : 0;
}
