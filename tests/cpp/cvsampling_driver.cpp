// cvsampling_driver.cpp — prints what include/eacham/CvSampling.hpp computes, for tests/test_cv_sampling.py (CPU only).
//   draws            the first 20 values of CvRNG((uint64)-1).next()
//   uniform n k      k values of uniform(0, n) from a fresh generator
//   subsets n m k    k subsets of m out of n (getSubset without a checkSubset), one per line, one stream
//   check            reads "x0 y0 .. (8 src) (8 dst)" lines, prints checkSubset of the homography callback per line
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>

#include "eacham/CvSampling.hpp"

using namespace eacham::hip;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    if (!std::strcmp(argv[1], "draws")) {
        CvRNG rng(0xffffffffffffffffull);
        for (int i = 0; i < 20; ++i) std::printf("%u\n", rng.next());
        return 0;
    }
    if (!std::strcmp(argv[1], "uniform") && argc >= 4) {
        CvRNG rng(0xffffffffffffffffull);
        const int n = std::atoi(argv[2]), k = std::atoi(argv[3]);
        for (int i = 0; i < k; ++i) std::printf("%d\n", rng.uniform(0, n));
        return 0;
    }
    if (!std::strcmp(argv[1], "subsets") && argc >= 5) {
        CvRNG rng(0xffffffffffffffffull);
        const int n = std::atoi(argv[2]), m = std::atoi(argv[3]), k = std::atoi(argv[4]);
        std::vector<int32_t> idx(m);
        for (int s = 0; s < k; ++s) {
            if (!cv_get_subset(rng, n, m, idx.data(), 1000, [](const int32_t*) { return true; })) return 3;
            for (int i = 0; i < m; ++i) std::printf("%d%c", idx[i], i + 1 < m ? ' ' : '\n');
        }
        return 0;
    }
    if (!std::strcmp(argv[1], "check")) {
        std::string line;
        while (std::getline(std::cin, line)) {
            std::istringstream is(line);
            float src[8], dst[8];
            for (float& v : src) is >> v;
            for (float& v : dst) is >> v;
            if (!is) break;
            std::printf("%d\n", cv_check_subset_homography(src, dst, 4) ? 1 : 0);
        }
        return 0;
    }
    return 2;
}
