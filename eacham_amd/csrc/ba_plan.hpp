// ba_plan.hpp — host-side analysis of the reduced camera system: fill-reducing ordering, panel layout, symbolic
// factorisation, elimination tree and the level schedule the device executes (ba.hip: sp_diag / sp_level /
// sp_backsolve). Pure C++ (no HIP): the CPU suite compiles and tests it on its own (tests/cpp/plan_driver.cpp).
//
// What it replaces: GTSAM's MULTIFRONTAL_CHOLESKY with COLAMD ordering, which the reference selects through
// LevenbergMarquardtParams::SetCeresDefaults (modules/sfm/reconstruction/BundleAdjuster.cpp:182-190). GTSAM orders and
// eliminates ALL variables; here the landmarks are eliminated first in closed form (3x3 blocks, ba.hip) and what is
// analysed is the camera graph of the Schur complement S (n = 6 nc + 5): cameras c, c' are adjacent iff they share a
// landmark; the five calibration columns K and the right-hand side are dense and go last.
//
// Layout. The permuted cameras are laid out in 64-column PANELS (the unit the device factorises per launch: two
// 32x32 diagonal blocks in one workgroup). An ordering is a list of NODES (sets of cameras: the leaves and separators of
// a nested dissection, or one node for a band ordering); every node starts on a panel boundary and its cameras are
// packed without gaps (a camera's six columns may straddle two panels of its node), the columns left over in a
// node's last panel are identity padding. K follows the last node; the right-hand side is carried as ROW 63 of the
// last panel (the root of the elimination tree, a neighbour of every panel because K is dense), with 1e100 on its
// diagonal: the factorisation then performs the forward substitution by itself and the back-substitution starts from
// z_root = -W[63][:] / W[63][63] (W = inverse of the root's 64x64 factor).
// Storage is TILES of 64 x 64 doubles, only those of the symbolic pattern of L (lower triangle incl. fill).
//
// Schedule. level(J) = 1 + max level(children of J) in the panel elimination tree (leaves 0). Launch 0 factorises the
// diagonal tiles of all leaves; launch l + 1 applies the updates A[I1][I2] -= L[I1][J] L[I2][J]^T of every panel J of
// level l — one workgroup per TARGET tile, its sources in ascending J: a fixed summation order, no atomics — and the
// workgroup of a diagonal target (P, P) with level(P) = l + 1 then factorises it. The dependent chain of the solve is
// the HEIGHT of the tree, not n / 64, and a launch touches only tiles of the pattern.
#pragma once

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <exception>
#include <mutex>
#include <system_error>
#include <numeric>
#include <utility>
#include <thread>
#include <vector>

namespace eacham {

constexpr int PLAN_PANEL = 64;   // columns per panel
constexpr int PLAN_CAM = 6;      // columns per camera
constexpr int PLAN_K = 5;        // calibration columns

enum BaOrdering { BA_ORDER_AUTO = 0, BA_ORDER_NATURAL = 1, BA_ORDER_RCM = 2, BA_ORDER_ND = 3 };

struct BaPlanItem {   // one workgroup of a level launch
    int tgt;          // tile index of the target A[I1][I2]
    int nsrc, src0;   // its sources of this level: srcs[src0 .. src0 + nsrc)
    int flags;        // bit 0: diagonal target (I1 == I2); bit 1: the target's panel is complete after this launch: factorise it
    int panel;        // I1 (the panel whose W the factorising item writes)
    int col;          // I2
    int shadow0, nshadow;  // tiles [shadow0, shadow0 + nshadow) are added to the target first (see plan_from_nodes)
};
struct BaPlanSrc {
    int tile_i, tile_j;  // raw strips A'[I1][J], A'[I2][J]
    int J;               // source panel (its W operands)
    int pad;
};

struct BaPlan {
    int nc = 0, npan = 0, ntiles = 0, n_levels = 0;   // ntiles: tiles of the pattern + shadow accumulators
    int n_shadow = 0;
    int posK = 0, rhs_row = 0;
    int ordering = BA_ORDER_NATURAL;   // the candidate chosen
    int nd_leaf = 0;                   // its leaf size (ND)
    double est_us = 0.0;               // cost model of the factorisation + back-substitution
    long long tile_updates = 0;        // rank-64 tile updates of the factorisation (dense: sum over steps of nt (nt + 1) / 2)
    std::vector<int> pos;              // [nc] first padded column of camera c
    std::vector<int> tile_map;         // [npan (npan + 1) / 2] (I >= J) -> tile index or -1
    std::vector<int> diag_tile;        // [npan]
    std::vector<int> parent, level;    // [npan]
    std::vector<std::vector<int>> strct;  // [npan] row panels I > J of column J in L, ascending
    std::vector<int> leaves;           // panels of level 0 (launch 0)
    std::vector<BaPlanItem> items;
    std::vector<BaPlanSrc> srcs;
    struct Launch { int first, count, n_first; };  // items [first, first + count), the first n_first of them factorise
    std::vector<Launch> launches;      // one per level l = 0 .. n_levels - 2
    std::vector<int> bs_order;         // back-substitution: workgroup b solves panel bs_order[b] (root first)
    std::vector<int> bs_ptr;           // [npan + 1] into bs_ent, indexed by workgroup
    std::vector<std::pair<int, int>> bs_ent;  // (tile of A'[I][J], I)
    std::vector<int> col_dest;         // [64 npan] index into delta_c (6 c + a, 6 nc + k) or -1
    std::vector<int> pad_cols;         // padded columns that carry an identity diagonal (the rhs row excluded)

    int tile(int I, int J) const { return tile_map[(size_t)I * (I + 1) / 2 + J]; }
};

namespace plan_detail {

typedef std::vector<std::vector<int>> Adj;

// breadth-first levels of the subgraph induced by the nodes with mark[v] == tag, from `start`; returns the visit order
inline std::vector<int> bfs(const Adj& adj, const std::vector<int>& mark, int tag, int start, std::vector<int>& lev) {
    std::vector<int> order;
    order.push_back(start);
    lev[start] = 0;
    for (size_t h = 0; h < order.size(); ++h) {
        const int u = order[h];
        for (int v : adj[u])
            if (mark[v] == tag && lev[v] < 0) {
                lev[v] = lev[u] + 1;
                order.push_back(v);
            }
    }
    return order;
}

// George-Liu pseudo-peripheral node of the component of `seed`
inline int pseudo_peripheral(const Adj& adj, const std::vector<int>& mark, int tag, int seed, std::vector<int>& lev) {
    int s = seed, best = -1;
    for (int it = 0; it < 6; ++it) {
        std::vector<int> order = bfs(adj, mark, tag, s, lev);
        const int ecc = lev[order.back()];
        int cand = -1;
        size_t cdeg = 0;
        for (int u : order)
            if (lev[u] == ecc && (cand < 0 || adj[u].size() < cdeg)) cand = u, cdeg = adj[u].size();
        for (int u : order) lev[u] = -1;
        if (ecc <= best) break;
        best = ecc;
        s = cand;
    }
    return s;
}

// Nested dissection by level-structure separators. The bisection of a node set does not depend on the leaf size the
// recursion stops at, so the recursion is recorded ONCE as a tree (down to the smallest leaf size of interest) and the
// node list of any leaf size is read off it (emit_nodes): the candidates of build_ba_plan share two dissections instead
// of running eight.
struct DTree {
    enum Kind { LEAF, TERMINAL, COMPONENTS, SPLIT };
    Kind kind = LEAF;
    std::vector<int> nodes;  // the set as it was handed to this level (a leaf is emitted in this order)
    std::vector<int> order;  // TERMINAL: no separator exists (a nearly complete graph) — emitted in level-structure order
    std::vector<int> sep;    // SPLIT: the separator, eliminated after both sides
    std::vector<DTree> kids;
};

// `par` > 0: the two sides of a split of some size are dissected by two threads, `par` levels deep. The sides are not adjacent
// (that is what the separator is for) and a recursion reads mark / lev of its own nodes and their neighbours only, so the two
// threads touch disjoint elements; the second side draws its tags from a range of its own.
inline void dissect_tree(const Adj& adj, std::vector<int>& mark, int& next_tag, const std::vector<int>& nodes, int min_leaf,
                         DTree& T, std::vector<int>& lev, int variants = 0, int par = 0) {
    T.nodes = nodes;
    if ((int)nodes.size() <= min_leaf) {
        T.kind = DTree::LEAF;
        return;
    }
    const int tag = next_tag++;
    for (int u : nodes) mark[u] = tag;
    // connected components first
    {
        std::vector<int> first = bfs(adj, mark, tag, nodes[0], lev);
        for (int u : first) lev[u] = -1;
        if (first.size() < nodes.size()) {
            std::vector<std::vector<int>> comps;
            std::vector<char> seen(adj.size(), 0);
            for (int s : nodes) {
                if (seen[s]) continue;
                std::vector<int> comp = bfs(adj, mark, tag, s, lev);
                for (int u : comp) lev[u] = -1, seen[u] = 1;
                comps.push_back(std::move(comp));
            }
            T.kind = DTree::COMPONENTS;
            T.kids.resize(comps.size());
            for (size_t k = 0; k < comps.size(); ++k) dissect_tree(adj, mark, next_tag, comps[k], min_leaf, T.kids[k], lev, variants, 0);
            return;
        }
    }
    // candidate roots of the level structure: pseudo-peripheral nodes reached from a few spread-out seeds; the level whose
    // removal leaves the smallest larger side + separator wins
    std::vector<int> seeds;
    {
        int s0 = nodes[0];
        for (int u : nodes)
            if (adj[u].size() < adj[s0].size()) s0 = u;
        seeds.push_back(s0);
        for (int k = 1; k <= variants; ++k) seeds.push_back(nodes[(size_t)k * nodes.size() / (variants + 1)]);
    }
    std::vector<int> order;
    int best_l = -1, depth = 0;
    long long best_score = -1;
    {
        std::vector<int> tried;
        for (int seed : seeds) {
            const int root = pseudo_peripheral(adj, mark, tag, seed, lev);
            if (std::find(tried.begin(), tried.end(), root) != tried.end()) continue;
            tried.push_back(root);
            std::vector<int> ord = bfs(adj, mark, tag, root, lev);
            const int dep = lev[ord.back()];
            if (dep >= 2) {
                std::vector<int> cnt(dep + 1, 0);
                for (int u : ord) cnt[lev[u]]++;
                long long before = 0, total = (long long)ord.size();
                for (int l = 0; l <= dep; ++l) {
                    if (l >= 1 && l < dep) {
                        const long long a = before, b = total - before - cnt[l];
                        auto pan = [](long long cams) { return (cams * PLAN_CAM + PLAN_PANEL - 1) / PLAN_PANEL; };
                        const long long score = 1000 * (pan(std::max(a, b)) + pan(cnt[l])) + std::max(a, b) + cnt[l];  // height in panels first
                        if (best_score < 0 || score < best_score) best_score = score, best_l = l, order = ord, depth = dep;
                    }
                    before += cnt[l];
                }
            } else if (order.empty()) {
                order = ord;
            }
            for (int u : ord) lev[u] = -1;
        }
    }
    if (best_l < 0) {  // (nearly) complete graph: no separator
        T.kind = DTree::TERMINAL;
        T.order = order;
        return;
    }
    {   // the levels of the winning structure again
        std::vector<int> again = bfs(adj, mark, tag, order[0], lev);
        (void)again;
        (void)depth;
    }
    // trim: a node of the separator level without a neighbour beyond it joins the near side; one without a
    // neighbour before it joins the far side unless a level-mate that just left for the near side touches it
    std::vector<int> left, right, sep;
    std::vector<int> moved_left;
    for (int u : order) {
        if (lev[u] != best_l) continue;
        bool ar = false;
        for (int v : adj[u])
            if (mark[v] == tag && lev[v] > best_l) ar = true;
        if (!ar) moved_left.push_back(u);
    }
    for (int u : moved_left) lev[u] = best_l - 1;  // (only the comparison with best_l matters from here on)
    for (int u : order) {
        if (lev[u] < best_l) left.push_back(u);
        else if (lev[u] > best_l) right.push_back(u);
        else {
            bool al = false;
            for (int v : adj[u])
                if (mark[v] == tag && lev[v] < best_l) al = true;
            if (al) sep.push_back(u);
            else right.push_back(u);
        }
    }
    for (int u : order) lev[u] = -1;
    if (left.empty() || right.empty()) {
        T.kind = DTree::TERMINAL;
        T.order = order;
        return;
    }
    T.kind = DTree::SPLIT;
    T.sep = sep;
    T.kids.resize(2);
    if (par > 0 && (int)left.size() > 4 * min_leaf && (int)right.size() > 4 * min_leaf) {
        int tag_r = next_tag + (1 << (18 + par));
        // The second side on a thread of its own when one can be had (the box limits threads: std::system_error -> both sides
        // here, one after the other). Nothing may leave a joinable thread behind or throw on it: an exception on either side is
        // carried to this thread and rethrown after the join.
        std::exception_ptr err_other, err_here;
        std::thread other;
        bool spawned = false;
        try {
            other = std::thread([&] {
                try {
                    dissect_tree(adj, mark, tag_r, right, min_leaf, T.kids[1], lev, variants, par - 1);
                } catch (...) {
                    err_other = std::current_exception();
                }
            });
            spawned = true;
        } catch (const std::system_error&) {
        }
        try {
            dissect_tree(adj, mark, next_tag, left, min_leaf, T.kids[0], lev, variants, spawned ? par - 1 : 0);
        } catch (...) {
            err_here = std::current_exception();
        }
        if (spawned) other.join();
        if (err_here) std::rethrow_exception(err_here);
        if (err_other) std::rethrow_exception(err_other);
        if (!spawned) dissect_tree(adj, mark, next_tag, right, min_leaf, T.kids[1], lev, variants, 0);
    } else {
        dissect_tree(adj, mark, next_tag, left, min_leaf, T.kids[0], lev, variants, 0);
        dissect_tree(adj, mark, next_tag, right, min_leaf, T.kids[1], lev, variants, 0);
    }
}

// the node list of the dissection stopped at `leaf` (>= the min_leaf the tree was built with): children first, separator last
inline void emit_nodes(const DTree& T, int leaf, std::vector<std::vector<int>>& out) {
    if ((int)T.nodes.size() <= leaf) {
        out.push_back(T.nodes);
        return;
    }
    switch (T.kind) {
        case DTree::LEAF: out.push_back(T.nodes); break;
        case DTree::TERMINAL: out.push_back(T.order); break;
        case DTree::COMPONENTS:
            for (const DTree& k : T.kids) emit_nodes(k, leaf, out);
            break;
        case DTree::SPLIT:
            emit_nodes(T.kids[0], leaf, out);
            emit_nodes(T.kids[1], leaf, out);
            if (!T.sep.empty()) out.push_back(T.sep);
            break;
    }
}

inline void dissect(const Adj& adj, std::vector<int>& mark, int& next_tag, const std::vector<int>& nodes, int leaf,
                    std::vector<std::vector<int>>& out, std::vector<int>& lev, int variants = 0) {
    DTree T;
    dissect_tree(adj, mark, next_tag, nodes, leaf, T, lev, variants);
    emit_nodes(T, leaf, out);
}


// reverse Cuthill-McKee over all components
inline std::vector<int> rcm(const Adj& adj) {
    const int n = (int)adj.size();
    std::vector<int> mark(n, 0), lev(n, -1), perm;
    std::vector<char> done(n, 0);
    for (int s = 0; s < n; ++s) {
        if (done[s]) continue;
        std::vector<int> comp = bfs(adj, mark, 0, s, lev);
        for (int u : comp) lev[u] = -1;
        std::vector<int> cm(n, 1);  // restrict to this component
        for (int u : comp) cm[u] = 2;
        const int root = pseudo_peripheral(adj, cm, 2, comp[0], lev);
        // Cuthill-McKee: neighbours in ascending degree
        std::vector<int> order;
        order.push_back(root);
        lev[root] = 0;
        for (size_t h = 0; h < order.size(); ++h) {
            const int u = order[h];
            std::vector<int> nb;
            for (int v : adj[u])
                if (cm[v] == 2 && lev[v] < 0) lev[v] = 1, nb.push_back(v);
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b; });
            order.insert(order.end(), nb.begin(), nb.end());
        }
        for (int u : order) lev[u] = -1, done[u] = 1;
        perm.insert(perm.end(), order.rbegin(), order.rend());
    }
    return perm;
}

}  // namespace plan_detail

// Cost model of one candidate (microseconds on MI355X, calibrated on the S200 / config-4 scenes): a level
// launch is bound either by the chain of a factorising workgroup (panel product + two one-wave 32x32 factors, plus a panel
// product per extra source of that target) or by its tile grid (two workgroups per CU).
struct BaPlanCost {
    double diag_us = 13.0, chain_us = 14.0, extra_src_us = 5.0, launch_us = 3.0, src_us = 5.0, backsolve_level_us = 3.0, backsolve_base_us = 6.0;
    int slots = 512;      // resident workgroups of a level launch (two per CU)
    int src_cap = 2;      // sources an accumulator takes per launch ahead of its target's deadline (two fit under the chain)
    int max_shadows = 7;  // shadow accumulators per target
};

// Lays out the nodes, derives the symbolic factor and the schedule. `nodes` = the ordering (cameras of each node in
// order); every camera exactly once.
inline void plan_from_nodes(int nc, const plan_detail::Adj& adj, const std::vector<std::vector<int>>& nodes, BaPlan& P,
                            const BaPlanCost& cm = BaPlanCost()) {
    P.nc = nc;
    P.pos.assign(nc, 0);
    int col = 0, end = 0;
    for (const auto& nd : nodes) {
        if (nd.empty()) continue;
        for (int c : nd) P.pos[c] = col, col += PLAN_CAM;
        end = col;
        col = (col + PLAN_PANEL - 1) / PLAN_PANEL * PLAN_PANEL;
    }
    // K behind the last camera if it fits the panel together with the rhs row (row 63), else on a fresh panel
    P.posK = end;
    if (end % PLAN_PANEL == 0 || end % PLAN_PANEL + PLAN_K > PLAN_PANEL - 1) P.posK = (end + PLAN_PANEL - 1) / PLAN_PANEL * PLAN_PANEL;
    P.npan = P.posK / PLAN_PANEL + 1;
    P.rhs_row = P.npan * PLAN_PANEL - 1;
    const int np = P.npan, root = np - 1;
    // panel adjacency
    std::vector<std::vector<char>> PA(np, std::vector<char>(np, 0));
    auto touch = [&](int c, int out[2]) { out[0] = P.pos[c] / PLAN_PANEL, out[1] = (P.pos[c] + PLAN_CAM - 1) / PLAN_PANEL; };
    for (int c = 0; c < nc; ++c) {
        int a[2], b[2];
        touch(c, a);
        PA[a[1]][a[0]] = 1;  // a straddling camera couples its two panels
        for (int c2 : adj[c]) {
            touch(c2, b);
            for (int x : a)
                for (int y : b) PA[std::max(x, y)][std::min(x, y)] = 1;
        }
    }
    for (int J = 0; J < root; ++J) PA[root][J] = 1;  // K and the right-hand side are dense
    P.strct.assign(np, {});
    P.parent.assign(np, -1);
    P.level.assign(np, 0);
    {
        std::vector<std::vector<char>> L = PA;  // symbolic factor, column by column
        for (int J = 0; J < np; ++J) {
            for (int I = J + 1; I < np; ++I)
                if (L[I][J]) P.strct[J].push_back(I);
            if (P.strct[J].empty()) continue;
            const int par = P.strct[J][0];
            P.parent[J] = par;
            for (int I : P.strct[J])
                if (I > par) L[I][par] = 1;
        }
    }
    for (int J = 0; J < np; ++J)
        if (P.parent[J] >= 0) P.level[P.parent[J]] = std::max(P.level[P.parent[J]], P.level[J] + 1);
    P.n_levels = 1 + *std::max_element(P.level.begin(), P.level.end());
    // tiles
    P.tile_map.assign((size_t)np * (np + 1) / 2, -1);
    P.diag_tile.assign(np, -1);
    P.ntiles = 0;
    for (int J = 0; J < np; ++J) {
        P.diag_tile[J] = P.tile_map[(size_t)J * (J + 1) / 2 + J] = P.ntiles++;
        for (int I : P.strct[J]) P.tile_map[(size_t)I * (I + 1) / 2 + J] = P.ntiles++;
    }
    // level launches
    P.leaves.clear();
    P.items.clear();
    P.srcs.clear();
    P.launches.clear();
    P.tile_updates = 0;
    for (int J = 0; J < np; ++J)
        if (P.level[J] == 0) P.leaves.push_back(J);
    double est = cm.diag_us;
    // Every update (J; I1 >= I2) of the structure may run in any launch from level(J) (its source exists) to
    // level(I2) - 1 (the launch before its target is read as a strip, or the launch that factorises it): a target whose
    // sources all sit on one level — (root, root) receives one from every leaf — would otherwise make ONE workgroup walk
    // a dozen sources while the launch waits. So (1) the sources of a target are dealt over its window, `src_cap` per
    // launch, and (2) a target that receives more than its window can take at that rate — the tiles of the root's row
    // receive a rank-64 update from EVERY panel, because K is dense — gets SHADOW tiles: zero-initialised accumulators
    // of their own workgroups that take the overflow in the launches before the deadline and are added to the target by
    // its workgroup in the deadline launch. Which source goes to which accumulator in which launch is fixed by the plan,
    // every accumulator sums in a fixed order: the result is deterministic (and no atomics).
    struct Tgt {
        int I1, I2;
        std::vector<int> src;  // source panels J, sorted by (level, J)
        size_t next = 0;
        int shadow0 = -1, nshadow = 0;
    };
    std::vector<Tgt> tgts;
    {
        std::vector<int> tgt_of((size_t)np * (np + 1) / 2, -1);
        std::vector<int> by_level(np);
        std::iota(by_level.begin(), by_level.end(), 0);
        std::stable_sort(by_level.begin(), by_level.end(), [&](int a, int b) { return P.level[a] < P.level[b]; });
        for (int J : by_level) {
            const auto& st = P.strct[J];
            for (size_t a = 0; a < st.size(); ++a)
                for (size_t b = a; b < st.size(); ++b) {
                    const int I2 = st[a], I1 = st[b];
                    const size_t key = (size_t)I1 * (I1 + 1) / 2 + I2;
                    if (tgt_of[key] < 0) {
                        tgt_of[key] = (int)tgts.size();
                        Tgt t;
                        t.I1 = I1, t.I2 = I2;
                        tgts.push_back(std::move(t));
                    }
                    tgts[tgt_of[key]].src.push_back(J);
                    ++P.tile_updates;
                }
        }
    }
    P.n_shadow = 0;
    for (Tgt& g : tgts) {
        const int deadline = P.level[g.I2] - 1, early_launches = deadline - P.level[g.src[0]];
        int early = 0;  // sources that exist before the deadline launch
        for (int J : g.src) early += P.level[J] < deadline ? 1 : 0;
        if (early_launches > 0) {
            const int rate = (early + early_launches - 1) / early_launches;  // sources per launch the target must absorb
            g.nshadow = std::min(cm.max_shadows, std::max(0, (rate + cm.src_cap - 1) / cm.src_cap - 1));
        }
        if (g.nshadow) {
            g.shadow0 = P.ntiles;
            P.ntiles += g.nshadow;
            P.n_shadow += g.nshadow;
        }
    }
    for (int l = 0; l + 1 < P.n_levels; ++l) {
        struct Tmp { int t, acc, first, count; bool fin; };  // acc: 0 = the target itself, k > 0 = its shadow k - 1
        std::vector<Tmp> tmp;
        for (int t = 0; t < (int)tgts.size(); ++t) {
            Tgt& g = tgts[t];
            const int deadline = P.level[g.I2] - 1;
            if (l > deadline) continue;
            size_t avail = g.next;
            while (avail < g.src.size() && P.level[g.src[avail]] <= l) ++avail;
            int have = (int)(avail - g.next);
            if (l == deadline) {  // everything that is left, and the shadows
                if (have || g.nshadow) tmp.push_back(Tmp{t, 0, (int)g.next, have, g.I1 == g.I2});
                g.next += have;
                continue;
            }
            if (!have) continue;
            // the launches before the deadline: src_cap per accumulator, more only if the rest of the window could not take it
            const int nacc = 1 + g.nshadow, left = deadline - l;  // launches of the window after this one (the deadline's among them)
            const int per = std::max(cm.src_cap, (have + nacc * (left + 1) - 1) / (nacc * (left + 1)));
            for (int k = 0; k < nacc && have > 0; ++k) {
                const int take = std::min(have, per);
                tmp.push_back(Tmp{t, k, (int)g.next, take, false});
                g.next += take;
                have -= take;
            }
        }
        // factorising items first (they are the launch's chain), then the longest lists
        std::stable_sort(tmp.begin(), tmp.end(), [&](const Tmp& x, const Tmp& y) {
            if (x.fin != y.fin) return x.fin;
            return x.count > y.count;
        });
        const int first = (int)P.items.size();
        int max_final_src = 0, max_src = 0, n_first = 0;
        long long work = 0;
        for (const Tmp& m : tmp) {
            const Tgt& g = tgts[m.t];
            BaPlanItem it{};
            it.tgt = m.acc == 0 ? P.tile(g.I1, g.I2) : g.shadow0 + m.acc - 1;
            it.nsrc = m.count;
            it.src0 = (int)P.srcs.size();
            it.flags = (g.I1 == g.I2 ? 1 : 0) | (m.fin ? 2 : 0);
            it.panel = g.I1;
            it.col = g.I2;
            const bool fold = m.acc == 0 && l == P.level[g.I2] - 1 && g.nshadow > 0;
            it.shadow0 = fold ? g.shadow0 : 0;
            it.nshadow = fold ? g.nshadow : 0;
            P.items.push_back(it);
            for (int k = m.first; k < m.first + m.count; ++k) {
                const int J = g.src[k];
                P.srcs.push_back(BaPlanSrc{P.tile(g.I1, J), P.tile(g.I2, J), J, 0});
            }
            if (m.fin) max_final_src = std::max(max_final_src, m.count), ++n_first;
            max_src = std::max(max_src, m.count);
            work += std::max(1, m.count);
        }
        P.launches.push_back(BaPlan::Launch{first, (int)P.items.size() - first, n_first});
        const double chain = cm.chain_us + cm.extra_src_us * std::max(0, max_final_src - 1);
        const double longest = cm.launch_us + cm.src_us * max_src;
        const double grid = cm.launch_us + cm.src_us * (double)((work + cm.slots - 1) / cm.slots);
        est += std::max(chain, std::max(longest, grid));
    }
    est += cm.backsolve_base_us + cm.backsolve_level_us * P.n_levels;
    P.est_us = est;
    // back-substitution: root first, a panel after its parent
    P.bs_order.resize(np);
    std::iota(P.bs_order.begin(), P.bs_order.end(), 0);
    std::stable_sort(P.bs_order.begin(), P.bs_order.end(), [&](int a, int b) { return P.level[a] != P.level[b] ? P.level[a] > P.level[b] : a > b; });
    P.bs_ptr.assign(np + 1, 0);
    P.bs_ent.clear();
    for (int b = 0; b < np; ++b) {
        const int J = P.bs_order[b];
        for (int I : P.strct[J]) P.bs_ent.emplace_back(P.tile(I, J), I);
        P.bs_ptr[b + 1] = (int)P.bs_ent.size();
    }
    // columns
    P.col_dest.assign((size_t)np * PLAN_PANEL, -1);
    for (int c = 0; c < nc; ++c)
        for (int a = 0; a < PLAN_CAM; ++a) P.col_dest[P.pos[c] + a] = PLAN_CAM * c + a;
    for (int k = 0; k < PLAN_K; ++k) P.col_dest[P.posK + k] = PLAN_CAM * nc + k;
    P.pad_cols.clear();
    for (int q = 0; q < np * PLAN_PANEL; ++q)
        if (P.col_dest[q] < 0 && q != P.rhs_row) P.pad_cols.push_back(q);
}

// edges: camera pairs (c < c') that share a landmark. hint: BaOrdering.
inline void build_ba_plan(int nc, const std::vector<std::pair<int, int>>& edges, int hint, BaPlan& P,
                          const BaPlanCost& cm = BaPlanCost()) {
    plan_detail::Adj adj(nc);
    for (const auto& e : edges)
        if (e.first != e.second) {
            adj[e.first].push_back(e.second);
            adj[e.second].push_back(e.first);
        }
    for (auto& a : adj) {
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    }
    std::vector<int> all(nc);
    std::iota(all.begin(), all.end(), 0);
    const int cams_per_panel = PLAN_PANEL / PLAN_CAM;  // 10
    // The candidates — the natural order (what a window of a sequence already is), reverse Cuthill-McKee, nested dissection
    // with four leaf sizes and one / several roots per bisection — are independent of each other except that the four leaf
    // sizes of a dissection share its recursion tree (plan_detail::DTree). On a scene of some size (more than four panels)
    // the work is a small task list run by a few host threads: the two dissections and the two band orderings first, the
    // four candidates of a dissection as soon as its tree exists (S200: 1.7 ms in sequence in round 3). The choice is
    // then made in the fixed candidate order — first strictly cheaper — exactly as a sequential loop would make it.
    struct Cand {
        int ordering, leaf, variants;
        bool force;
        BaPlan plan;
    };
    std::vector<Cand> cands;
    cands.push_back({BA_ORDER_NATURAL, 0, 0, false, BaPlan()});
    const bool small = nc <= 2 * cams_per_panel;  // two panels: nothing to reorder
    if (hint == BA_ORDER_RCM || (hint == BA_ORDER_AUTO && !small)) cands.push_back({BA_ORDER_RCM, 0, 0, hint == BA_ORDER_RCM, BaPlan()});
    const int leaves[4] = {cams_per_panel, 2 * cams_per_panel + 1, 3 * cams_per_panel + 2, 4 * cams_per_panel + 2};
    const int variant_list[2] = {0, 3};
    const bool with_nd = hint == BA_ORDER_ND || (hint == BA_ORDER_AUTO && !small);
    size_t nd_first = cands.size();
    if (with_nd) {
        bool first = true;
        for (int variants : variant_list)
            for (int leaf : leaves) {
                cands.push_back({BA_ORDER_ND, leaf, variants, hint == BA_ORDER_ND && first, BaPlan()});
                first = false;
            }
    }
    plan_detail::DTree trees[2];
    auto build_tree = [&](int k) {
        std::vector<int> mark(nc, 0), lev(nc, -1);
        int tag = 1;
        // (the dissection with several roots per bisection is the critical path of the whole plan: its top two splits fan out)
        plan_detail::dissect_tree(adj, mark, tag, all, leaves[0], trees[k], lev, variant_list[k], nc > 8 * cams_per_panel ? (k == 1 ? 2 : 1) : 0);
    };
    auto evaluate = [&](Cand& c) {
        std::vector<std::vector<int>> nodes;
        if (c.ordering == BA_ORDER_NATURAL) nodes = {all};
        else if (c.ordering == BA_ORDER_RCM) nodes = {plan_detail::rcm(adj)};
        else plan_detail::emit_nodes(trees[c.variants == variant_list[0] ? 0 : 1], c.leaf, nodes);
        plan_from_nodes(nc, adj, nodes, c.plan, cm);
        c.plan.ordering = c.ordering;
        c.plan.nd_leaf = c.leaf;
    };
    // tasks: -1 / -2 = build tree 0 / 1 (each releases its four candidates), k >= 0 = evaluate candidate k
    std::vector<int> queue;
    if (with_nd) queue = {-2, -1};  // (the dissection with several roots per bisection is the longest task: first)
    for (size_t k = 0; k < nd_first; ++k) queue.push_back((int)k);
    size_t pending = queue.size() + (with_nd ? 8 : 0);
    std::mutex mu;
    std::condition_variable cv;
    std::exception_ptr failure;  // the first exception of a task: the queue is drained, every worker leaves, it is rethrown after the joins
    auto worker = [&]() {
        for (;;) {
            int task;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return !queue.empty() || pending == 0; });
                if (queue.empty()) return;
                task = queue.front();
                queue.erase(queue.begin());
            }
            try {
                if (task < 0) build_tree(-task - 1);
                else evaluate(cands[task]);
            } catch (...) {
                {
                    std::lock_guard<std::mutex> lock(mu);
                    if (!failure) failure = std::current_exception();
                    queue.clear();
                    pending = 0;
                }
                cv.notify_all();
                return;
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                if (pending == 0) return;  // (another task failed meanwhile)
                --pending;
                if (task < 0)
                    for (int i = 0; i < 4; ++i) queue.push_back((int)nd_first + 4 * (-task - 1) + i);
            }
            cv.notify_all();
        }
    };
    if (with_nd && nc > 4 * cams_per_panel) {
        std::vector<std::thread> th;
        for (int k = 0; k < 3; ++k) {
            try {
                th.emplace_back(worker);
            } catch (const std::system_error&) {  // no more threads to be had: the ones there are (at least this one) do all of it
                break;
            }
        }
        worker();
        for (auto& t : th) t.join();
    } else {
        worker();
    }
    if (failure) std::rethrow_exception(failure);
    size_t best = 0;
    for (size_t k = 1; k < cands.size(); ++k)
        if (cands[k].force || cands[k].plan.est_us < cands[best].plan.est_us - 1e-9) best = k;
    BaPlan& bestp = cands[best].plan;
    P = std::move(bestp);
}

}  // namespace eacham
