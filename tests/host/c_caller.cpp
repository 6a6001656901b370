// A C++ caller written against the reference's public interface only (include/vbz.h mirrors vbz/vbz.h:11-141),
// linked with -lvbz_hip instead of -lvbz: what a maintainer's existing code does after re-linking.  Mirrors the
// idiom of the reference's own tests (vbz/test/vbz_test.cpp: perform_compression_test): bound, compress, size,
// decompress, compare; sized and unsized; int16 zig-zag and uint32 plain.  Prints "ok" and exits 0 on success.
#include <vbz.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

template <typename T>
static bool round_trip(const std::vector<T>& in, bool zigzag, unsigned level, unsigned version, bool sized)
{
    CompressionOptions opt{ zigzag, (unsigned)sizeof(T), level, version };
    const vbz_size_t bytes = (vbz_size_t)(in.size() * sizeof(T));
    const vbz_size_t bound = vbz_max_compressed_size(bytes, &opt);
    if (vbz_is_error(bound)) return false;
    std::vector<char> comp(bound);
    const vbz_size_t used = sized ? vbz_compress_sized(in.data(), bytes, comp.data(), (vbz_size_t)comp.size(), &opt)
                                  : vbz_compress(in.data(), bytes, comp.data(), (vbz_size_t)comp.size(), &opt);
    if (vbz_is_error(used)) {
        std::fprintf(stderr, "compress: %s\n", vbz_error_string(used));
        return false;
    }
    std::vector<T> out(in.size());
    vbz_size_t got;
    if (sized) {
        if (vbz_decompressed_size(comp.data(), used, &opt) != bytes) return false;
        got = vbz_decompress_sized(comp.data(), used, out.data(), bytes, &opt);
    } else {
        got = vbz_decompress(comp.data(), used, out.data(), bytes, &opt);
    }
    if (vbz_is_error(got)) {
        std::fprintf(stderr, "decompress: %s\n", vbz_error_string(got));
        return false;
    }
    return got == bytes && std::memcmp(out.data(), in.data(), bytes) == 0;
}

int main()
{
    std::vector<int16_t> sig(100000);
    uint32_t x = 12345;
    int32_t level = 400;
    for (auto& s : sig) {  // a slowly moving level with noise: the shape of nanopore signal
        x = x * 1664525u + 1013904223u;
        if ((x >> 20) % 97 == 0) level = 300 + (int32_t)((x >> 8) % 400);
        s = (int16_t)(level + (int32_t)((x >> 12) % 31) - 15);
    }
    std::vector<uint32_t> wide(50000);
    for (size_t i = 0; i < wide.size(); ++i) {
        x = x * 1664525u + 1013904223u;
        wide[i] = x >> ((x >> 28) * 2);
    }
    bool ok = true;
    for (unsigned version = 0; version <= 1; ++version)
        for (unsigned lvl = 0; lvl <= 1; ++lvl)
            for (int sized = 0; sized <= 1; ++sized) {
                ok = ok && round_trip(sig, true, lvl, version, sized != 0);
                ok = ok && round_trip(wide, false, lvl, version, sized != 0);
            }
    CompressionOptions bad{ true, 3, 1, 0 };
    ok = ok && vbz_max_compressed_size(30, &bad) == VBZ_INTEGER_SIZE_ERROR;
    std::puts(ok ? "ok" : "FAILED");
    return ok ? 0 : 1;
}
