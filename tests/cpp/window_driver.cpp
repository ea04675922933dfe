// window_driver.cpp — CPU check of eacham_amd/csrc/ba_window.hpp (the structure of the dense form of the Schur stage for a local
// window). Reads   nc nl seed max_obs dup rows_override   from stdin, draws a random window (landmarks with 0..max_obs observations
// by distinct cameras; with dup = 1 one landmark sees a camera twice), builds the structure and checks the definition of the
// header line by line: every landmark with observations exactly once and in index order, its rows = its observations in ascending
// camera order followed by its own row, groups within their bounds, padding beyond. Prints one JSON line. Test infrastructure
// (tests/test_ba_window.py); nothing here runs on the product path.
#include <cstdio>
#include <random>
#include <vector>

#include "../../eacham_amd/csrc/ba_window.hpp"

using namespace eacham;

int main() {
    int nc, nl, seed, max_obs, dup, rows_override;
    if (scanf("%d %d %d %d %d %d", &nc, &nl, &seed, &max_obs, &dup, &rows_override) != 6) return 2;
    std::mt19937_64 rng(seed);
    std::vector<int> lm_ptr(nl + 1, 0);
    std::vector<unsigned> obs_cam;
    std::vector<double> obs_uv;
    for (int j = 0; j < nl; ++j) {
        const int m = (int)(rng() % (std::min(max_obs, nc) + 1));
        std::vector<unsigned> cams;
        while ((int)cams.size() < m) {
            const unsigned c = (unsigned)(rng() % nc);
            bool seen = false;
            for (unsigned x : cams) seen |= x == c;
            if (!seen) cams.push_back(c);
        }
        if (dup && j == nl / 2 && m >= 1) cams.push_back(cams[0]);
        for (unsigned c : cams) {   // (unsorted on purpose: the structure sorts a landmark's rows by camera)
            obs_cam.push_back(c);
            obs_uv.push_back((double)(rng() % 1000) + 0.25);
            obs_uv.push_back((double)(rng() % 1000) + 0.75);
        }
        lm_ptr[j + 1] = (int)obs_cam.size();
    }
    long long total_rows = 0;
    for (int j = 0; j < nl; ++j)
        if (lm_ptr[j + 1] > lm_ptr[j]) total_rows += lm_ptr[j + 1] - lm_ptr[j] + 1;
    BaWin W;
    const bool built = build_window(nc, nl, lm_ptr.data(), obs_cam.data(), obs_uv.data(), W, rows_override);
    int bad = 0;
    long long rows_used = 0, lm_seen = 0;
    if (built) {
        const int rows = W.rows, lmax = rows / 4;
        int next_lm = 0;   // landmarks are met in index order
        for (size_t g = 0; g < W.groups.size(); ++g) {
            const BaWinGroup G = W.groups[g];
            if (G.nlm < 1 || G.nlm > lmax || G.nrows > rows) ++bad;
            int r = 0;
            for (int t = 0; t < lmax; ++t) {
                const int j = W.lmid[g * lmax + t];
                if (t >= G.nlm) { bad += j != -1; continue; }
                while (next_lm < nl && lm_ptr[next_lm + 1] == lm_ptr[next_lm]) ++next_lm;
                if (j != next_lm) { ++bad; continue; }
                ++next_lm, ++lm_seen;
                const int m = lm_ptr[j + 1] - lm_ptr[j];
                // the landmark's observations, ascending by camera, with their measurements
                std::vector<int> idx(m);
                for (int i = 0; i < m; ++i) idx[i] = lm_ptr[j] + i;
                std::sort(idx.begin(), idx.end(), [&](int a, int b) { return obs_cam[a] < obs_cam[b]; });
                for (int i = 0; i < m; ++i) {
                    const GrpI2 ri = W.rowinfo[g * rows + r + i];
                    bad += ri.x != (int)obs_cam[idx[i]] || ri.y != t;
                    bad += W.uv[2 * (g * rows + r + i)] != obs_uv[2 * (size_t)idx[i]] || W.uv[2 * (g * rows + r + i) + 1] != obs_uv[2 * (size_t)idx[i] + 1];
                }
                const GrpI2 own = W.rowinfo[g * rows + r + m];
                bad += own.x != nc || own.y != t || W.lmrow[g * lmax + t] != r + m;
                r += m + 1;
            }
            bad += r != G.nrows;
            for (int q = r; q < rows; ++q) bad += W.rowinfo[g * rows + q].x != -1;
            rows_used += r;
        }
        while (next_lm < nl && lm_ptr[next_lm + 1] == lm_ptr[next_lm]) ++next_lm;
        bad += next_lm != nl || rows_used != total_rows || W.n_rows != total_rows;
    }
    printf("{\"built\": %s, \"rows\": %d, \"groups\": %zu, \"total_rows\": %lld, \"landmarks\": %lld, \"bad\": %d, \"default_rows\": %d, "
           "\"lds_bytes\": %zu, \"partial_bytes\": %zu}\n",
           built ? "true" : "false", W.rows, W.groups.size(), total_rows, lm_seen, bad, win_rows_for(total_rows, nc),
           built ? win_lds_bytes(nc, W.rows) : (size_t)0, built ? W.groups.size() * win_stride(nc) * sizeof(double) : (size_t)0);
    return 0;
}
