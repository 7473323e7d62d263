// Host Horner (32 windows, c = 8) with the portable 8 x 32-bit host code against host_ec64.hpp.
// build: clang++ -O3 -std=c++17 -I uzkge_amd/csrc tools/microbench/host_horner.cpp
#include <chrono>
#include <cstdio>
#include <vector>
#include "host_ec64.hpp"
using namespace uzk;
int main() {
    Affine G; G.x = Fq::one(); G.y = Fq::dbl(Fq::one());
    XYZZ g1 = xyzz_from_affine(G);
    std::vector<XYZZ> s(32);
    XYZZ acc = g1;
    for (int w = 0; w < 32; ++w) { acc = xyzz_dbl(acc); xyzz_add(acc, g1); s[w] = acc; }
    auto t0 = std::chrono::steady_clock::now();
    Jac r{};
    for (int rep = 0; rep < 200; ++rep) {
        XYZZ total = xyzz_inf();
        for (int w = 31; w >= 0; --w) { if (w != 31) for (int d = 0; d < 8; ++d) total = xyzz_dbl(total); xyzz_add(total, s[w]); }
        r = xyzz_to_jac(total);
    }
    auto t1 = std::chrono::steady_clock::now();
    Jac r2{};
    for (int rep = 0; rep < 200; ++rep) r2 = h64::horner(32, 8, [&](uint32_t w) -> const XYZZ& { return s[w]; });
    auto t2 = std::chrono::steady_clock::now();
    std::printf("old %.1f us  new %.1f us (%u %u)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 200,
                std::chrono::duration<double, std::micro>(t2 - t1).count() / 200, r.x.v[0], r2.x.v[0]);
}
