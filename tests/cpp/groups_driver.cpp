// groups_driver.cpp — CPU check of eacham_amd/csrc/ba_groups.hpp (the landmark-major structure of the Schur stage).
// Reads   nc nl seed max_obs dup   from stdin, draws a random problem (landmarks with 0..max_obs observations; with dup = 1 some
// landmarks see one camera twice), builds the structure, checks its invariants and EXECUTES it in plain doubles the way
// ba_schur_groups / ba_assemble_groups do — random Et rows per observation, a random Y row per landmark, every lane the sum of
// its slice in entry order, a block's partials in slot order — against the direct sum over every row pair of every landmark. Prints one JSON line. Test infrastructure (tests/test_ba_groups.py); nothing here runs on the product path.
#include <cmath>
#include <cstdio>
#include <map>
#include <random>
#include <vector>

#include "../../eacham_amd/csrc/ba_groups.hpp"

using namespace eacham;

int main() {
    int nc, nl, seed, max_obs, dup;
    if (scanf("%d %d %d %d %d", &nc, &nl, &seed, &max_obs, &dup) != 5) return 2;
    std::mt19937_64 rng(seed);
    std::vector<int> lm_ptr(nl + 1, 0);
    std::vector<unsigned> obs_cam;
    std::vector<double> obs_uv;
    for (int j = 0; j < nl; ++j) {
        const int m = (int)(rng() % (max_obs + 1));
        const int base = (int)(rng() % nc);
        for (int i = 0; i < m; ++i) {
            unsigned c = (unsigned)((base + (rng() % std::min(nc, 3 * max_obs + 1))) % nc);
            if (!dup)  // distinct cameras per landmark
                for (bool again = true; again;) {
                    again = false;
                    for (int a = lm_ptr[j]; a < (int)obs_cam.size(); ++a)
                        if (obs_cam[a] == c) c = (c + 1) % nc, again = true;
                }
            obs_cam.push_back(c);
            obs_uv.push_back((double)(rng() % 1000));
            obs_uv.push_back((double)(rng() % 1000));
        }
        lm_ptr[j + 1] = (int)obs_cam.size();
    }
    BaGroups G;
    const bool ok = build_groups(nc, nl, lm_ptr.data(), obs_cam.data(), obs_uv.data(), G);
    if (!ok) {
        printf("{\"built\": false}\n");
        return 0;
    }
    int bad = 0;
    // invariants
    std::vector<char> seen(nl, 0);
    for (int k = 0; k < G.n_used; ++k) seen[G.lm[k]]++;
    for (int j = 0; j < nl; ++j) bad += seen[j] != (lm_ptr[j + 1] > lm_ptr[j] ? 1 : 0);
    int max_rows = 0, max_lm = 0;
    for (const BaGroup& g : G.groups) {
        max_rows = std::max(max_rows, g.nrows), max_lm = std::max(max_lm, g.nlm);
        bad += g.nrows > G.rows || g.nlm > G.rows / 4 || g.n_entries > GRP_ENT_PER_ROW * G.rows;
    }
    // rows of random values (observation rows X, 6 x 3; landmark rows Y), indexed by the padded (group, row)
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    const size_t R = (size_t)G.rows;
    std::vector<double> rowv(18 * G.groups.size() * R);
    for (double& v : rowv) v = U(rng);
    // direct sums, keyed by block
    const long long W = nc + 1;
    std::map<long long, std::vector<double>> direct;
    auto add = [&](std::vector<double>& S, const double* x, const double* y) {
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) S[6 * a + b] += x[3 * a] * y[3 * b] + x[3 * a + 1] * y[3 * b + 1] + x[3 * a + 2] * y[3 * b + 2];
    };
    for (size_t g = 0; g < G.groups.size(); ++g) {
        const BaGroup& gr = G.groups[g];
        for (int t = 0; t < gr.nlm; ++t) {
            const int k = gr.lm0 + t, r0 = G.lrow[k] - gr.row0, m1 = G.lrow[k + 1] - G.lrow[k];
            bad += G.lmid[g * (R / 4) + t] != G.lm[k] || G.lmrow[g * (R / 4) + t] != r0 + m1 - 1;
            for (int a = 0; a < m1; ++a)
                for (int b = a; b < m1; ++b) {
                    const GrpI2 ia = G.rowinfo[g * R + r0 + a], ib = G.rowinfo[g * R + r0 + b];
                    bad += ia.y != t || ib.y != t || (a == m1 - 1) != (ia.x == nc);
                    const int ca = ia.x, cb = ib.x;
                    const double *xa = &rowv[18 * (g * R + r0 + a)], *xb = &rowv[18 * (g * R + r0 + b)];
                    auto& S = direct[(long long)std::min(ca, cb) * W + std::max(ca, cb)];
                    if (S.empty()) S.assign(36, 0.0);
                    if (a == b) add(S, xa, xa);
                    else if (ca < cb) add(S, xa, xb);
                    else if (ca > cb) add(S, xb, xa);
                    else add(S, xa, xb), add(S, xb, xa);
                }
        }
        for (int r = gr.nrows; r < G.rows; ++r) bad += G.rowinfo[g * R + r].x != -1;
    }
    // the structure, executed: a lane sums its slice in entry order, the lanes of a segment are folded by shift-and-add steps
    std::vector<double> partial(36 * (size_t)std::max(G.n_parts, 1), 0.0);
    std::vector<int> written(std::max(G.n_parts, 1), 0);
    long long real_entries = 0, padded = 0;
    for (size_t g = 0; g < G.groups.size(); ++g) {
        const BaGroup& gr = G.groups[g];
        int segments = 0;
        for (int c = 0; c < gr.nchunks; ++c) {
            const BaChunk ch = G.chunks[gr.chunk0 + c];
            padded += 256LL * ch.n4;
            std::vector<double> acc(64 * 36, 0.0);
            int after[64];
            for (int l = 0; l < 64; ++l) {
                int len = 0;
                for (int i = 0; i < 4 * ch.n4; ++i) {
                    const uint32_t e = G.ent[((size_t)ch.ent0 + i / 4) * 256 + 4 * l + i % 4];
                    const int r1 = (int)(e & 0xffffu), r2 = (int)(e >> 16);
                    bad += (r1 == G.rows) != (r2 == G.rows) || (r1 != G.rows && (r1 >= gr.nrows || r2 >= gr.nrows));
                    if (r1 == G.rows) continue;
                    bad += len != i;  // a slice's entries are its first steps
                    ++len, ++real_entries;
                    std::vector<double> one(36, 0.0);
                    add(one, &rowv[18 * (g * R + r1)], &rowv[18 * (g * R + r2)]);
                    for (int v = 0; v < 36; ++v) acc[36 * l + v] += one[v];
                }
                const uint32_t info = G.laneinfo[64 * (size_t)(gr.chunk0 + c) + l];
                after[l] = (int)(info >> 28);
                bad += len > GRP_SLICE || after[l] >= GRP_SEG || (l + after[l]) / GRP_ROW != l / GRP_ROW || (len == 0 && info != 0);
            }
            for (int d = 1; d < GRP_SEG; d <<= 1) {   // what the kernel does with shuffles
                std::vector<double> nx(acc);
                for (int l = 0; l < 64; ++l)
                    if (after[l] >= d)
                        for (int v = 0; v < 36; ++v) nx[36 * l + v] = acc[36 * l + v] + acc[36 * (l + d) + v];
                acc.swap(nx);
            }
            for (int l = 0; l < 64; ++l) {
                const int slot = (int)(G.laneinfo[64 * (size_t)(gr.chunk0 + c) + l] & 0x0fffffffu) - 1;
                bad += slot >= G.n_parts;
                if (slot < 0) continue;
                ++segments, ++written[slot];
                for (int v = 0; v < 36; ++v) partial[36 * (size_t)slot + v] = acc[36 * l + v];
            }
        }
        bad += segments != gr.n_segments;
    }
    for (int q = 0; q < G.n_parts; ++q) bad += written[q] != 1;
    bad += real_entries != G.n_entries;
    double worst = 0.0;
    int n_cam_blocks = 0;
    std::map<long long, int> blocks_seen;
    for (const GrpI4& b : G.blk) {
        const long long key = (long long)b.x * W + b.y;
        bad += blocks_seen[key]++ != 0 || b.x > b.y;
        n_cam_blocks += b.y < nc && b.x != b.y;
        std::vector<double> s(36, 0.0);
        for (int k = 0; k < b.w; ++k)
            for (int v = 0; v < 36; ++v) s[v] += partial[36 * (size_t)(b.z + k) + v];
        auto it = direct.find(key);
        for (int v = 0; v < 36; ++v) {
            const double want = it == direct.end() ? 0.0 : it->second[v];
            worst = std::max(worst, std::fabs(s[v] - want) / (1.0 + std::fabs(want)));
        }
    }
    for (const auto& kv : direct) bad += blocks_seen.find(kv.first) == blocks_seen.end();  // every block with entries is in the table
    for (int c = 0; c < nc; ++c) bad += !blocks_seen.count((long long)c * W + c) || !blocks_seen.count((long long)c * W + nc);
    bad += !blocks_seen.count((long long)nc * W + nc);
    printf("{\"built\": true, \"rows\": %d, \"groups\": %zu, \"max_rows\": %d, \"max_lm\": %d, \"chunks\": %d, \"parts\": %d, \"blocks\": %d, "
           "\"offdiag_blocks\": %d, \"entries\": %lld, \"padded\": %lld, \"bad\": %d, \"worst\": %.3e}\n",
           G.rows, G.groups.size(), max_rows, max_lm, G.n_chunks, G.n_parts, G.n_blk, n_cam_blocks, G.n_entries,
           padded, bad, worst);
    return 0;
}
