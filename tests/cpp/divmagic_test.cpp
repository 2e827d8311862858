// CPU check of flock_amd/csrc/divmagic.hpp (the multiply-high remainder the generic predicate kernel uses for `col % m`):
// against the hardware `%` for divisors of every shape and dividends at every edge.  Built and run by tests/test_divmagic.py.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "divmagic.hpp"

using namespace flockgpu;

int main() {
    std::vector<uint32_t> ds = {1, 2, 3, 5, 6, 7, 10, 11, 12, 13, 25, 100, 123, 125, 127, 128, 129, 255, 256, 257, 641, 1000, 1009, 4095, 4096, 4097,
                                65535, 65536, 65537, 1000003, 16777215, 16777216, 16777217, 0x7ffffffeu, 0x7fffffffu, 0x80000000u, 0x80000001u,
                                0xfffffffeu, 0xffffffffu};
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < 4000; ++i) ds.push_back(rnd() >> (rnd() & 31));
    long checked = 0;
    for (uint32_t d : ds) {
        if (!d) continue;
        const UMod32 m = umod32_make(d);
        std::vector<uint32_t> ns = {0, 1, 2, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu};
        for (uint64_t q = 1; q * d <= 0xffffffffull; q = q * 3 + 1) { ns.push_back((uint32_t)(q * d)); ns.push_back((uint32_t)(q * d - 1)); ns.push_back((uint32_t)(q * d + 1)); }
        for (int i = 0; i < 2000; ++i) ns.push_back(rnd());
        for (uint32_t n : ns) {
            if (umod32_apply(n, m) != n % d) { printf("FAIL u %u %% %u: %u\n", n, d, umod32_apply(n, m)); return 1; }
            ++checked;
        }
        if (d <= 0x7fffffffu)
            for (uint32_t n : ns) {
                const int32_t x = (int32_t)n;
                const int64_t want = (int64_t)x % (int64_t)d;   // truncated: the sign of the dividend
                if ((int64_t)smod32_apply(x, m) != want) { printf("FAIL s %d %% %u: %d\n", x, d, smod32_apply(x, m)); return 1; }
                ++checked;
            }
    }
    // small divisors exhaustively over a dense range of dividends
    for (uint32_t d = 1; d <= 300; ++d) {
        const UMod32 m = umod32_make(d);
        for (uint32_t n = 0; n < 200000; ++n)
            if (umod32_apply(n, m) != n % d) { printf("FAIL dense %u %% %u\n", n, d); return 1; }
        for (uint32_t n = 0xffffffffu; n > 0xffffffffu - 50000; --n)
            if (umod32_apply(n, m) != n % d) { printf("FAIL top %u %% %u\n", n, d); return 1; }
    }
    printf("ok %ld\n", checked);
    return 0;
}
