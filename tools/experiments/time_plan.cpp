#include <chrono>
#include <cstdio>
#include "../../eacham_amd/csrc/ba_plan.hpp"
using namespace eacham;
int main() {
    int nc, ne;
    if (scanf("%d %d", &nc, &ne) != 2) return 2;
    std::vector<std::pair<int,int>> edges(ne);
    for (auto& e : edges) if (scanf("%d %d", &e.first, &e.second) != 2) return 2;
    for (int rep = 0; rep < 5; ++rep) {
        BaPlan P;
        auto t0 = std::chrono::steady_clock::now();
        build_ba_plan(nc, edges, 0, P);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("auto: %.0f us, ordering %d leaf %d levels %d\n", us, P.ordering, P.nd_leaf, P.n_levels);
    }
    for (int hint : {1, 2, 3}) {
        BaPlan P;
        auto t0 = std::chrono::steady_clock::now();
        build_ba_plan(nc, edges, hint, P);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("hint %d: %.0f us\n", hint, us);
    }
}
