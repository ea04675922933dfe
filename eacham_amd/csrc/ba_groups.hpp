// ba_groups.hpp — the landmark-major structure of the Schur stage (round 5), pure C++ (host form + the definition the device form
// of eacham_ba_prepare reproduces bit for bit; tests/cpp/groups_driver.cpp executes it in plain doubles on the CPU).
//
// What it replaces: RefineBA's reduced camera system (the landmark elimination GTSAM's multifrontal solver performs for
// modules/sfm/reconstruction/BundleAdjuster.cpp:182-216) needs S_ab -= sum_j Et_a(j) Et_b(j)^T over the landmarks j seen by both
// cameras a and b. Rounds 1-4 wrote Et (144 B per observation) to HBM and gathered two rows per (landmark, a, b) entry from
// block-ordered pair lists: 353 of the 520 MB an LM iteration moved. Here a WORKGROUP owns a group of landmarks — consecutive in
// an order that keeps their camera sets close — recomputes their Jacobians, keeps Et of the group's <= `rows` observations in
// LDS and multiplies the group's entries out of LDS: a LANE owns a slice (<= GRP_SLICE consecutive entries of ONE camera block) and
// sums its products in entry order, the <= GRP_SEG adjacent lanes of a segment (slices of one block) are folded by three
// shift-and-add steps, the segment's first lane writes one 6x6 partial; ba_assemble_groups adds a block's partials (contiguous
// in memory) in a fixed order. No atomics, every sum has one order.
//
// The calibration border and the right-hand side are the same sum with one more "camera": every landmark gets a row of its own,
// Y_j = [EKt_j (5x3); gt_j (1x3)] (what Et would be for a pseudo-camera with the 5 calibration columns + the right-hand side),
// so block (c, K) = sum Et_c Y^T = [border | rhs share] and block (K, K) = sum Y Y^T = [K corner | rhs_K; . ] come out of the
// same entries.
//
// Definition (both forms of eacham_ba_prepare produce exactly this):
//  1. used landmarks (>= 1 observation) in stable order of key = morton(min camera, max camera) of their observations;
//  2. landmark k (sorted rank) owns rows_k = m_k + 1 consecutive ROWS (its observations in landmark order, then its own row)
//     and e_k entries (every row pair a <= b; two for a != b in the same camera); cost_k = max(rows_k, 4, ceil(e_k rows / ent_max));
//     group(k) = floor(prefix cost(k) / R0), R0 = rows - max cost + 1: a group holds <= rows rows, <= rows / 4 landmarks and
//     <= ent_max entries; if max cost > rows / 2 the structure is not built (the pair-list path of rounds 1-4 serves).
//     Per-group arrays are PADDED to these bounds (row r of group g at g rows + r, landmark t at g rows / 4 + t): a thread's
//     loads do not wait for the group record;
//  3. a group's entries (block key = c1 (nc + 1) + c2 with c1 <= c2, pseudo-camera = nc; local rows r1, r2 with camera(r1) = c1)
//     sorted by (block key, emission index); a run of L entries of one block is cut into n = ceil(L / GRP_SLICE) slices of
//     balanced length (the first L mod n one longer) that take n consecutive LANES; the group's lanes — the runs of more than four
//     entries in block order, then the runs of at most four in block order — fill chunks of 64; a SEGMENT = consecutive lanes of one block, at most GRP_SEG of them, inside one row of GRP_ROW lanes of a chunk; chunk n4 =
//     ceil(longest slice / 4) steps; entries are stored [chunk][step][lane][4] (one 16-byte load per lane and step), shorter
//     slices padded with null entries (row `rows` = zeros);
//  4. laneinfo[64 chunk + lane] = (lanes after this one in its segment) << 28 | (slot + 1 for a segment's first lane, else 0):
//     slot = where the segment's partial is stored: all segments in (group, lane) order stably sorted by block key, so that a
//     block's partials are CONTIGUOUS (blk = {c1, c2, first slot, count}); the mandatory blocks ((c, c), (c, K), (K, K)) are in the
//     table even without entries (count 0); longblk = the blocks with more than GRP_LONG partials.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

namespace eacham {

constexpr int GRP_SLICE = 8;              // entries of one block a lane multiplies at most
constexpr int GRP_SEG = 8;                // lanes of one block folded into one partial at most (three shift-and-add steps)
constexpr int GRP_ROW = 16;               // a segment stays inside a row of 16 lanes (the fold shifts with DPP row operations)
constexpr int GRP_ENT_PER_ROW = 16;       // ent_max = GRP_ENT_PER_ROW * rows
constexpr int GRP_LONG = 48;              // a block with more partials than this is added by a workgroup, not a wave

struct BaGroup { int lm0, nlm, row0, nrows, chunk0, nchunks, n_entries, n_segments; };
struct BaChunk { int ent0, n4; };         // first uint4-row (of 64 lanes) of the chunk's entries; steps of 4 entries
struct GrpI2 { int x, y; };
struct GrpI4 { int x, y, z, w; };

struct BaGroups {
    int rows = 0;        // rows per group (0: not built, the pair lists serve)
    int n_used = 0, n_rows = 0, n_chunks = 0, n_parts = 0, n_blk = 0;
    long long n_entries = 0, n_ent4 = 0;   // real entries; uint4-rows (64 lanes x 4 entries each) of the padded entry array
    std::vector<BaGroup> groups;
    std::vector<int> lm;                   // [n_used] landmark of sorted rank k
    std::vector<int> lrow;                 // [n_used + 1] first row of rank k
    std::vector<int> lmid;                 // [groups x rows / 4] landmark t of group g (-1 beyond nlm)
    std::vector<int> lmrow;                // [groups x rows / 4] local row of that landmark's own row
    std::vector<GrpI2> rowinfo;            // [groups x rows] {camera (nc: a landmark's own row, -1: no row), landmark index inside the group}
    std::vector<double> uv;                // [2 x groups x rows]
    std::vector<BaChunk> chunks;           // [n_chunks]
    std::vector<uint32_t> ent;             // [n_ent4 x 64 x 4] r1 | r2 << 16 (local rows; null row = rows)
    std::vector<uint32_t> laneinfo;        // [64 n_chunks] lanes after this one in its segment << 28 | (slot + 1 at a segment's first lane)
    std::vector<GrpI4> blk;                // {c1, c2, first slot, count}, ascending block key
    std::vector<int> longblk;              // indices of the blocks with more than GRP_LONG partials (a whole workgroup adds those)
};

#ifdef __HIPCC__
#define GRP_HD __host__ __device__
#else
#define GRP_HD
#endif
GRP_HD inline uint32_t grp_spread16(uint32_t x) {
    x &= 0xffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
GRP_HD inline uint32_t grp_morton(uint32_t a, uint32_t b) { return (grp_spread16(a) << 1) | grp_spread16(b); }  // a in the odd bits
// rows per group: small problems (a local window) spread over more, smaller groups; large ones take the largest group of which two
// fit a CU's LDS (ba_schur_groups: 144 B per row) — fewer, longer runs per block: 107 k partials instead of 191 k at 256 rows on the
// 200-camera / 50 k-landmark scene, the assembly 14 instead of 22 us
inline int grp_rows_for(long long total_rows) { return total_rows <= 32768 ? 128 : 480; }

// lm_ptr / obs_cam / obs_uv: the landmark-ordered observation arrays of the problem. Returns false (out.rows = 0) when a
// landmark is too heavy for a group.
inline bool build_groups(int nc, int nl, const int* lm_ptr, const unsigned* obs_cam, const double* obs_uv, BaGroups& out, int rows_override = 0) {
    out = BaGroups();
    if (nc >= 65535) return false;
    // 1. order (a plain sort of key << 32 | landmark: stable by construction)
    std::vector<unsigned long long> order;
    order.reserve(nl);
    long long total_rows = 0;
    for (int j = 0; j < nl; ++j) {
        const int a0 = lm_ptr[j], a1 = lm_ptr[j + 1];
        if (a1 == a0) continue;
        unsigned mn = obs_cam[a0], mx = obs_cam[a0];
        for (int a = a0 + 1; a < a1; ++a) mn = std::min(mn, obs_cam[a]), mx = std::max(mx, obs_cam[a]);
        order.push_back((unsigned long long)grp_morton(mn, mx) << 32 | (unsigned)j);
        total_rows += a1 - a0 + 1;
    }
    const int nu = (int)order.size();
    if (total_rows > 0x7fffffffLL) return false;
    std::sort(order.begin(), order.end());
    const int rows = rows_override > 0 ? rows_override : grp_rows_for(total_rows), ent_max = GRP_ENT_PER_ROW * rows;
    if (rows % 4 != 0 || rows > 32767) return false;
    // 2. costs, groups
    std::vector<int> cost(nu), ecount(nu);
    out.lm.resize(nu);
    out.lrow.assign(nu + 1, 0);
    int cmax = 4;
    bool any_dup = false;  // some landmark sees a camera twice: the entries then go through the general (sorting) form below
    for (int k = 0; k < nu; ++k) {
        const int j = (int)(order[k] & 0xffffffffu);
        out.lm[k] = j;
        const int a0 = lm_ptr[j], a1 = lm_ptr[j + 1], m = a1 - a0;
        long long e = (long long)(m + 1) * (m + 2) / 2;
        for (int a = a0; a < a1; ++a)
            for (int b = a + 1; b < a1; ++b) e += obs_cam[a] == obs_cam[b] ? 1 : 0;
        any_dup = any_dup || e != (long long)(m + 1) * (m + 2) / 2;
        if (e > ent_max) return false;
        ecount[k] = (int)e;
        cost[k] = std::max(std::max(m + 1, 4), (int)((e * rows + ent_max - 1) / ent_max));
        cmax = std::max(cmax, cost[k]);
        out.lrow[k + 1] = out.lrow[k] + m + 1;
    }
    if (cmax > rows / 2) return false;
    const int R0 = rows - cmax + 1;
    std::vector<int> grp(nu);
    {
        long long c = 0;
        for (int k = 0; k < nu; ++k) grp[k] = (int)(c / R0), c += cost[k];
    }
    const int ng = nu ? grp[nu - 1] + 1 : 0;
    out.rows = rows;
    out.n_used = nu;
    out.n_rows = out.lrow[nu];
    out.groups.assign(ng, BaGroup{0, 0, 0, 0, 0, 0, 0, 0});
    for (int g = 0, k = 0; g < ng; ++g) {  // empty groups are legal (a landmark's cost may span a whole window)
        while (k < nu && grp[k] < g) ++k;
        int k1 = k;
        while (k1 < nu && grp[k1] == g) ++k1;
        BaGroup& G = out.groups[g];
        G.lm0 = k; G.nlm = k1 - k; G.row0 = out.lrow[k]; G.nrows = out.lrow[k1] - out.lrow[k];
    }
    // rows and landmarks, padded per group
    const int lmax = rows / 4;
    out.rowinfo.assign((size_t)ng * rows, GrpI2{-1, -1});
    out.uv.assign(2 * (size_t)ng * rows, 0.0);
    out.lmid.assign((size_t)ng * lmax, -1);
    out.lmrow.assign((size_t)ng * lmax, 0);
    for (int g = 0; g < ng; ++g) {
        const BaGroup& G = out.groups[g];
        for (int t = 0; t < G.nlm; ++t) {
            const int k = G.lm0 + t, j = out.lm[k], a0 = lm_ptr[j], m = lm_ptr[j + 1] - a0;
            const size_t base = (size_t)g * rows + (out.lrow[k] - G.row0);
            for (int i = 0; i < m; ++i) {
                out.rowinfo[base + i] = GrpI2{(int)obs_cam[a0 + i], t};
                out.uv[2 * (base + i)] = obs_uv[2 * (size_t)(a0 + i)];
                out.uv[2 * (base + i) + 1] = obs_uv[2 * (size_t)(a0 + i) + 1];
            }
            out.rowinfo[base + m] = GrpI2{nc, t};
            out.lmid[(size_t)g * lmax + t] = j;
            out.lmrow[(size_t)g * lmax + t] = out.lrow[k] - G.row0 + m;
        }
    }
    // 3. entries. Two forms of the same definition: with a landmark that sees a camera twice somewhere, the general one — a
    //    stable counting sort of every entry over the group's own camera pairs (local camera ids in ascending camera order, so
    //    the local pair order is the order of the block keys); otherwise the run of block (la, lb) is the landmarks that see
    //    both cameras — the AND of two per-camera masks over the group's landmarks — in ascending order, and the entries are
    //    read off the mask (no entry list, no sort: 1.1 -> 0.2 ms for a 19-camera window on the host).
    struct Seg { uint32_t key; int where; };  // where = 64 chunk + lane
    struct Slice { uint32_t key; int first, len; int la, lb; };  // general: first = position in `sorted`; masks: position inside the run
    std::vector<Seg> segs;
    std::vector<int> stamp(nc + 1, -1), local(nc + 1, 0), cams;
    std::vector<uint32_t> e_key, e_val, sorted;
    std::vector<int> cnt;
    std::vector<Slice> slices, short_runs;
    const int mw = (lmax + 63) / 64;                        // mask words per camera
    std::vector<unsigned long long> mask;
    std::vector<unsigned short> rowtab;                     // [la][t]: the row of landmark t in local camera la (valid where the mask bit is set)
    const uint32_t null_ent = (uint32_t)rows | ((uint32_t)rows << 16);
    const uint32_t W = (uint32_t)nc + 1;
    segs.reserve((size_t)out.n_rows);
    out.ent.reserve(2 * (size_t)out.n_rows * 8);
    out.laneinfo.reserve((size_t)out.n_rows * 2);
    for (int g = 0; g < ng; ++g) {
        BaGroup& G = out.groups[g];
        // the group's cameras in ascending order -> local ids
        cams.clear();
        for (int r = 0; r < G.nrows; ++r) {
            const int c = out.rowinfo[(size_t)g * rows + r].x;
            if (stamp[c] != g) stamp[c] = g, cams.push_back(c);
        }
        std::sort(cams.begin(), cams.end());
        const int U = (int)cams.size();
        for (int i = 0; i < U; ++i) local[cams[i]] = i;
        slices.clear();
        short_runs.clear();
        int ne = 0;
        auto add_run = [&](int la, int lb, int pos, int L) {  // a run of L > 0 entries of local pair (la, lb)
            const uint32_t key2 = (uint32_t)cams[la] * W + (uint32_t)cams[lb];
            if (L > 4) {
                const int n = (L + GRP_SLICE - 1) / GRP_SLICE;
                for (int i = 0, at = pos; i < n; ++i) {
                    const int len = L / n + (i < L % n ? 1 : 0);
                    slices.push_back(Slice{key2, at, len, la, lb});
                    at += len;
                }
            } else {  // runs of at most four entries (one step) take the group's LAST lanes: the chunks that hold only such
                      // lanes run one step of four entries instead of two
                short_runs.push_back(Slice{key2, pos, L, la, lb});
            }
        };
        if (any_dup) {
            e_key.clear();
            e_val.clear();
            cnt.assign((size_t)U * U + 1, 0);
            for (int t = 0; t < G.nlm; ++t) {
                const int k = G.lm0 + t, r0 = out.lrow[k] - G.row0, m = out.lrow[k + 1] - out.lrow[k] - 1;
                const GrpI2* ri = &out.rowinfo[(size_t)g * rows + r0];
                for (int a = 0; a <= m; ++a) {
                    const int la = local[ri[a].x];
                    const uint32_t ra = (uint32_t)(r0 + a);
                    e_key.push_back((uint32_t)(la * U + la));
                    e_val.push_back(ra | (ra << 16));
                    for (int b = a + 1; b <= m; ++b) {
                        const int lb = local[ri[b].x];
                        const uint32_t rb = (uint32_t)(r0 + b);
                        if (la < lb) e_key.push_back((uint32_t)(la * U + lb)), e_val.push_back(ra | (rb << 16));
                        else if (la > lb) e_key.push_back((uint32_t)(lb * U + la)), e_val.push_back(rb | (ra << 16));
                        else {
                            e_key.push_back((uint32_t)(la * U + la)), e_val.push_back(ra | (rb << 16));
                            e_key.push_back((uint32_t)(la * U + la)), e_val.push_back(rb | (ra << 16));
                        }
                    }
                }
            }
            ne = (int)e_key.size();
            for (int i = 0; i < ne; ++i) cnt[e_key[i] + 1]++;
            int pos = 0;
            for (int q = 0; q < U * U; ++q) {   // cnt[q + 1]: count -> running cursor of pair q; the run's slices
                const int L = cnt[q + 1];
                cnt[q + 1] = pos;
                if (L > 0) add_run(q / U, q % U, pos, L);
                pos += L;
            }
            sorted.assign((size_t)ne, 0);
            for (int i = 0; i < ne; ++i) sorted[cnt[e_key[i] + 1]++] = e_val[i];
        } else {
            mask.assign((size_t)U * mw, 0ull);
            if (rowtab.size() < (size_t)U * lmax) rowtab.resize((size_t)U * lmax);
            for (int r = 0; r < G.nrows; ++r) {
                const GrpI2 ri = out.rowinfo[(size_t)g * rows + r];
                const int la = local[ri.x];
                mask[(size_t)la * mw + (ri.y >> 6)] |= 1ull << (ri.y & 63);
                rowtab[(size_t)la * lmax + ri.y] = (unsigned short)r;
            }
            for (int la = 0; la < U; ++la)
                for (int lb = la; lb < U; ++lb) {
                    int L = 0;
                    for (int w = 0; w < mw; ++w) L += __builtin_popcountll(mask[(size_t)la * mw + w] & mask[(size_t)lb * mw + w]);
                    if (L > 0) add_run(la, lb, 0, L);
                    ne += L;
                }
        }
        slices.insert(slices.end(), short_runs.begin(), short_runs.end());
        G.n_entries = ne;
        out.n_entries += ne;
        const int ns = (int)slices.size();   // = lanes, in block order
        G.chunk0 = out.n_chunks;
        G.nchunks = (ns + 63) / 64;
        G.n_segments = 0;
        for (int c = 0; c < G.nchunks; ++c) {
            const int l_end = std::min(ns, 64 * c + 64);
            int longest = 0;
            for (int l = 64 * c; l < l_end; ++l) longest = std::max(longest, slices[l].len);
            const int n4 = (longest + 3) / 4;
            out.chunks.push_back(BaChunk{(int)out.n_ent4, n4});
            out.ent.resize(out.ent.size() + (size_t)n4 * 256, null_ent);
            uint32_t* dst = out.ent.data() + (size_t)out.n_ent4 * 256;
            const size_t info0 = out.laneinfo.size();
            out.laneinfo.resize(info0 + 64, 0);
            // segments: consecutive lanes of one block, at most GRP_SEG, inside one row of GRP_ROW lanes; one forward pass
            for (int h = 64 * c; h < l_end;) {
                int e = h + 1;
                while (e < l_end && e < h + GRP_SEG && e / GRP_ROW == h / GRP_ROW && slices[e].key == slices[h].key) ++e;
                for (int si = h; si < e; ++si) out.laneinfo[info0 + (si - 64 * c)] = (uint32_t)(e - 1 - si) << 28;
                out.laneinfo[info0 + (h - 64 * c)] |= 1u;  // (the slot is filled in below)
                segs.push_back(Seg{slices[h].key, 64 * (G.chunk0 + c) + (h - 64 * c)});
                ++G.n_segments;
                h = e;
            }
            for (int si = 64 * c; si < l_end; ++si) {
                const Slice& S = slices[si];
                const int l = si - 64 * c;
                if (any_dup) {
                    for (int i = 0; i < S.len; ++i) dst[(size_t)(i / 4) * 256 + 4 * l + (i % 4)] = sorted[S.first + i];
                } else {  // the landmarks of the run from position S.first on
                    int skip = S.first, i = 0;
                    for (int w = 0; w < mw && i < S.len; ++w) {
                        unsigned long long m = mask[(size_t)S.la * mw + w] & mask[(size_t)S.lb * mw + w];
                        for (; m && i < S.len; m &= m - 1) {
                            if (skip > 0) {
                                --skip;
                                continue;
                            }
                            const int t = 64 * w + __builtin_ctzll(m);
                            dst[(size_t)(i / 4) * 256 + 4 * l + (i % 4)] =
                                (uint32_t)rowtab[(size_t)S.la * lmax + t] | ((uint32_t)rowtab[(size_t)S.lb * lmax + t] << 16);
                            ++i;
                        }
                    }
                }
            }
            out.n_ent4 += n4;
        }
        out.n_chunks += G.nchunks;
    }
    out.n_parts = (int)segs.size();
    // slots: the segments in stable order of their block key (a plain sort of key << 32 | sequence number); the block table = that
    // order's runs merged with the mandatory blocks
    std::vector<unsigned long long> sord(segs.size());
    for (size_t i = 0; i < sord.size(); ++i) sord[i] = (unsigned long long)segs[i].key << 32 | (unsigned long long)i;
    std::sort(sord.begin(), sord.end());
    std::vector<int> ord(segs.size());
    for (size_t i = 0; i < ord.size(); ++i) {
        ord[i] = (int)(sord[i] & 0xffffffffu);
        const int where = segs[ord[i]].where;
        out.laneinfo[where] = (out.laneinfo[where] & 0xf0000000u) | (uint32_t)(i + 1);
    }
    {
        size_t i = 0;
        auto runs_below = [&](uint32_t limit, bool inclusive) {  // emit the runs with key < limit (<= limit)
            while (i < ord.size() && (segs[ord[i]].key < limit || (inclusive && segs[ord[i]].key == limit))) {
                const uint32_t kk = segs[ord[i]].key;
                size_t e = i;
                while (e < ord.size() && segs[ord[e]].key == kk) ++e;
                out.blk.push_back(GrpI4{(int)(kk / W), (int)(kk % W), (int)i, (int)(e - i)});
                i = e;
            }
        };
        auto mandatory = [&](uint32_t kk) {
            runs_below(kk, false);
            if (i < ord.size() && segs[ord[i]].key == kk) runs_below(kk, true);
            else out.blk.push_back(GrpI4{(int)(kk / W), (int)(kk % W), (int)i, 0});
        };
        for (int c = 0; c < nc; ++c) mandatory((uint32_t)c * W + c), mandatory((uint32_t)c * W + nc);
        mandatory((uint32_t)nc * W + nc);
        runs_below(0xffffffffu, true);
    }
    out.n_blk = (int)out.blk.size();
    for (int b = 0; b < out.n_blk; ++b)
        if (out.blk[b].w > GRP_LONG) out.longblk.push_back(b);
    return true;
}

}  // namespace eacham
