// thread_scaling.cpp -- the single-buffer C ABI (include/vbz.h) called from N host threads at once.
//   g++ -O2 -std=c++17 -pthread tools/thread_scaling.cpp -Iinclude -Lvbz_compression_amd/lib -lvbz_hip -Wl,-rpath,$PWD/vbz_compression_amd/lib -o /tmp/thread_scaling
// Prints, per thread count, the time of one compress + decompress of a 100 000-sample read and the aggregate rate.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "vbz.h"

int main(int argc, char** argv)
{
    const int n = 100000, reads = 64, rounds = argc > 1 ? atoi(argv[1]) : 16;
    std::vector<std::vector<int16_t>> sig(reads, std::vector<int16_t>(n));
    uint64_t x = 88172645463325252ull;
    for (auto& s : sig) {
        int level = 400;
        for (int i = 0; i < n; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            if (i % 32 == 0) level = 200 + (int)(x >> 40) % 320;
            s[i] = (int16_t)(level + (int)((x >> 20) & 63) - 32);
        }
    }
    CompressionOptions o{ true, 2, 1, 1 };
    const vbz_size_t cap = vbz_max_compressed_size(2 * n, &o);
    std::vector<uint8_t> warm(cap);
    if (vbz_is_error(vbz_compress_sized(sig[0].data(), 2 * n, warm.data(), cap, &o))) return 1;
    for (int nt : { 64, 1, 2, 4, 8, 16, 32, 64 }) {  // the first line warms the library up (contexts, pinned arenas)
        std::vector<std::thread> ts;
        int bad = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < nt; ++t)
            ts.emplace_back([&, t] {
                std::vector<uint8_t> c(cap);
                std::vector<int16_t> back(n);
                for (int r = 0; r < rounds; ++r)
                    for (int i = t; i < reads; i += nt) {
                        const vbz_size_t cs = vbz_compress_sized(sig[i].data(), 2 * n, c.data(), cap, &o);
                        const vbz_size_t ds = vbz_is_error(cs) ? cs : vbz_decompress_sized(c.data(), cs, back.data(), 2 * n, &o);
                        if (vbz_is_error(ds) || memcmp(back.data(), sig[i].data(), 2 * n) != 0) ++bad;
                    }
            });
        for (auto& t : ts) t.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %2d: %.3f ms per round trip, %.0f MB/s of raw signal each way, bad %d\n", nt, dt * 1e3 / (reads * rounds),
               reads * rounds * 2.0 * n / dt / 1e6, bad);
    }
    return 0;
}
