// idtable_driver.cpp — CPU check of include/eacham/FlatMap.hpp: IdTable (the generation-stamped landmark table of the glue's graph
// walks) and FlatMap against std::unordered_map / std::map on random key streams, over many "calls" on ONE table (growth, reuse without
// clearing, the stamp's wrap-around). Prints "ok" or the first difference. Test infrastructure (tests/test_flatmap.py).
#include <cstdio>
#include <map>
#include <random>
#include <unordered_map>

#include "../../include/eacham/FlatMap.hpp"

using namespace eacham::hip;

int main() {
    std::mt19937_64 rng(12345);
    IdTable T;
    for (int call = 0; call < 400; ++call) {
        const size_t expect = call % 7 == 0 ? 20000 : 1 + rng() % 3000;   // sizes that make the table grow and then be reused far below its size
        if (call == 200) T.cur = 0xfffffffeu;                              // two calls from the stamp's wrap-around
        T.reset(expect);
        std::unordered_map<unsigned, uint32_t> ref;
        const unsigned range = call % 3 == 0 ? 64 : 1u << 30;              // dense ids (many repeats) and sparse ones
        for (size_t k = 0; k < expect; ++k) {
            const unsigned key = (unsigned)(rng() % range);
            bool fresh = false;
            uint32_t& v = T.slot(key, fresh);
            const auto it = ref.find(key);
            if (fresh != (it == ref.end())) return std::printf("call %d: key %u fresh %d, reference says %d\n", call, key, (int)fresh, (int)(it == ref.end())), 1;
            if (fresh) {
                v = (uint32_t)ref.size();
                ref[key] = v;
            } else if (v != it->second) {
                return std::printf("call %d: key %u holds %u, reference %u\n", call, key, v, it->second), 1;
            }
        }
    }
    FlatMap F;
    std::map<unsigned, unsigned> M;
    for (int k = 0; k < 5000; ++k) {
        const unsigned key = (unsigned)(rng() % 300), val = (unsigned)rng();
        if (k % 5 == 4) {
            if (F.erase(key) != M.erase(key)) return std::printf("FlatMap erase %u\n", key), 1;
        } else {
            F[key] = val, M[key] = val;
        }
        if (F.size() != M.size() || F.count(key) != M.count(key)) return std::printf("FlatMap size / count at %d\n", k), 1;
    }
    auto it = M.begin();
    for (const auto& kv : F) {
        if (kv.first != it->first || kv.second != it->second) return std::printf("FlatMap order\n"), 1;
        ++it;
    }
    std::printf("ok\n");
    return 0;
}
