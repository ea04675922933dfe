// The reference's own call pattern on the drop-in matcher (apps/sfm/main.cpp:84-109): every ORDERED frame pair is one
// std::async(&Match, &matcher, d1, d2).get(), issued from a pool of worker threads (the reference uses
// std::for_each(std::execution::par_unseq, ...), i.e. one TBB worker per core) on ONE shared matcher instance.
//   match_async_driver <in.bin> <out.bin> <threads> [repeat]
// in : int32 F, int32 dim, then per frame int32 n + n*dim floats.  out: per ordered pair (i != j, i-major) the flat
// {q, t} list sorted by q, then doubles {seconds of the LAST repeat, calls, batches, uploads, cache_hits}.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <future>
#include <thread>

#include "eacham/FeatureMatcherHip.hpp"

using namespace eacham::hip;

template <class T> static T rd1(std::ifstream& f) { T v; f.read((char*)&v, sizeof(T)); return v; }

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::ofstream out(argv[2], std::ios::binary);
    const int threads = std::atoi(argv[3]), repeat = argc > 4 ? std::atoi(argv[4]) : 1;
    const int F = rd1<int32_t>(in), dim = rd1<int32_t>(in);
    std::vector<std::vector<float>> store(F);
    std::vector<DescriptorView> frames(F);
    for (int f = 0; f < F; ++f) {
        const int n = rd1<int32_t>(in);
        store[f].resize((size_t)n * dim);
        in.read((char*)store[f].data(), sizeof(float) * store[f].size());
        frames[f] = DescriptorView{store[f].data(), n, dim};
    }
    std::vector<std::pair<int, int>> pairs;  // main.cpp:84-92: (i, j) and (j, i) for every i < j
    for (int i = 0; i < F; ++i)
        for (int j = 0; j < F; ++j)
            if (i != j) pairs.push_back({i, j});
    FeatureMatcherHip matcher(0.8f);
    std::vector<FeatureMatcherHip::MatchType> res(pairs.size());
    double seconds = 0.0;
    for (int rep = 0; rep < repeat; ++rep) {
        std::atomic<size_t> next{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        for (int w = 0; w < threads; ++w)
            pool.emplace_back([&] {
                for (size_t p = next.fetch_add(1); p < pairs.size(); p = next.fetch_add(1)) {
                    auto fut = std::async(std::launch::async, [&, p] {       // main.cpp:107-109
                        return matcher.Match(frames[pairs[p].first], frames[pairs[p].second]);
                    });
                    res[p] = fut.get();
                }
            });
        for (auto& t : pool) t.join();
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    for (size_t p = 0; p < pairs.size(); ++p) {
        std::vector<uint32_t> flat;
        for (unsigned q = 0; q < (unsigned)frames[pairs[p].first].rows; ++q) {
            auto it = res[p].find(q);
            if (it != res[p].end()) { flat.push_back(q); flat.push_back(it->second); }
        }
        const int64_t n = (int64_t)flat.size();
        out.write((const char*)&n, sizeof(n));
        out.write((const char*)flat.data(), sizeof(uint32_t) * flat.size());
    }
    const auto st = matcher.stats();
    const double tail[5] = {seconds, (double)st.calls, (double)st.batches, (double)st.uploads, (double)st.cache_hits};
    out.write((const char*)tail, sizeof(tail));
    {   // A descriptor buffer rewritten IN PLACE, only at words the sampled fingerprint does not look at (2048 words: the
        // samples are the even ones + the last): the default cache cannot see it; SetFullContentCheck(true) and
        // Invalidate(data) must both make the next Match() use the new values.
        const int rows = 64;
        std::vector<float> X((size_t)rows * dim);
        for (size_t i = 0; i < X.size(); ++i) X[i] = (float)((i * 37 + 11) % 251);
        const DescriptorView xv{X.data(), rows, dim}, other = frames[0];
        auto rewrite = [&](int salt) {
            for (size_t i = 1; i + 1 < X.size(); i += 2) X[i] = (float)((i * 53 + 7 * salt) % 241);  // odd words, not the last
        };
        auto fresh = [&] { FeatureMatcherHip m(0.8f); return m.Match(xv, other); };
        int bad = 0;
        if ((size_t)rows * dim == 2048 && other.rows >= 2) {
            FeatureMatcherHip sampled(0.8f), full(0.8f), told(0.8f);
            full.SetFullContentCheck(true);
            const auto before = sampled.Match(xv, other);
            (void)full.Match(xv, other);
            (void)told.Match(xv, other);
            rewrite(1);
            const auto want = fresh();
            const bool changed = want != before;                 // the rewrite does change the answer
            const bool stale = sampled.Match(xv, other) == before && changed;
            told.Invalidate(X.data());
            if (full.Match(xv, other) != want) ++bad;
            if (told.Match(xv, other) != want) ++bad;
            if (full.stats().uploads < 3) ++bad;                 // X, other, X again
            std::printf("stale-buffer check: answer changed %d, sampled fingerprint served the old frame %d, full check / Invalidate wrong %d\n",
                        (int)changed, (int)stale, bad);
            if (!changed) bad += 100;
        }
        if (bad) return 3;
    }
    std::printf("match_async_driver: %zu ordered pairs, %d threads, %.4f s (last of %d) = %.0f Match()/s; calls %llu batches %llu uploads %llu hits %llu\n",
                pairs.size(), threads, seconds, repeat, pairs.size() / seconds, (unsigned long long)st.calls,
                (unsigned long long)st.batches, (unsigned long long)st.uploads, (unsigned long long)st.cache_hits);
    return 0;
}
