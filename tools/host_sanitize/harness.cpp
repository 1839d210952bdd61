#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../../include/maxsim.h"
int main() {
    const int n_q = 6, Lq = 20, n_d = 200, dim = 128, rows_per = 37;
    std::vector<uint16_t> Q((size_t)n_q * Lq * dim), D((size_t)n_d * rows_per * dim);
    for (auto &v : Q) v = (uint16_t)(0x3c00 + rand() % 512);
    for (auto &v : D) v = (uint16_t)(0x3c00 + rand() % 512);
    std::vector<int32_t> off(n_d + 1);
    for (int i = 0; i <= n_d; ++i) off[i] = i * rows_per;
    std::vector<float> ref((size_t)n_q * n_d);
    msim_fwd_host(0, Q.data(), n_q, Lq, D.data(), off.data(), nullptr, n_d, dim, ref.data(), n_d, 0, 1);
    std::vector<std::thread> ts;
    int bad = 0;
    for (int t = 0; t < 4; ++t)
        ts.emplace_back([&, t] {
            std::vector<float> out((size_t)n_q * n_d);
            for (int it = 0; it < 20; ++it) {
                msim_fwd_host(0, Q.data(), n_q, Lq, D.data(), off.data(), nullptr, n_d, dim, out.data(), n_d, 0, 2 + t);
                for (size_t i = 0; i < out.size(); ++i) if (out[i] != ref[i]) { __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED); break; }
            }
        });
    for (auto &t : ts) t.join();
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
