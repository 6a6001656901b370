// The reference's HDF5 integration test (vbz_plugin/test/vbz_hdf_plugin_test.cpp:15-136) against this library's
// plugin and a real libhdf5: register the filter statically (vbz_plugin_info, as vbz_register() does), then for
// {int,uint}{8,16,32}: create a chunked dataset (chunk = count / 8) with filter 32020, H5Dwrite, close, reopen,
// H5Dread, compare -- iota(100) at zstd level 5 and random values at level 1, zig-zag on for every type as there,
// FILTER_VBZ_VERSION 1.  libhdf5 is loaded at run time (argv[1] or the usual names).  Prints "ok", exits 0.
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>
#include <vector>

#include <vbz_hdf_plugin.h>

typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;

static void* h5;
template <typename F>
static F sym(const char* name)
{
    void* p = dlsym(h5, name);
    if (!p) {
        std::fprintf(stderr, "libhdf5 lacks %s\n", name);
        std::exit(3);
    }
    return reinterpret_cast<F>(p);
}
#define H5(ret, name, ...) static auto name = sym<ret (*)(__VA_ARGS__)>(#name)

template <typename T>
static const char* type_global()
{
    if (std::is_same<T, int8_t>::value) return "H5T_NATIVE_INT8_g";
    if (std::is_same<T, uint8_t>::value) return "H5T_NATIVE_UINT8_g";
    if (std::is_same<T, int16_t>::value) return "H5T_NATIVE_INT16_g";
    if (std::is_same<T, uint16_t>::value) return "H5T_NATIVE_UINT16_g";
    if (std::is_same<T, int32_t>::value) return "H5T_NATIVE_INT32_g";
    return "H5T_NATIVE_UINT32_g";
}

template <typename T>
static bool run(const char* path, const std::vector<T>& data, unsigned level)
{
    H5(hid_t, H5Fcreate, const char*, unsigned, hid_t, hid_t);
    H5(hid_t, H5Fopen, const char*, unsigned, hid_t);
    H5(herr_t, H5Fclose, hid_t);
    H5(hid_t, H5Screate_simple, int, const hsize_t*, const hsize_t*);
    H5(herr_t, H5Sclose, hid_t);
    H5(hid_t, H5Pcreate, hid_t);
    H5(herr_t, H5Pclose, hid_t);
    H5(herr_t, H5Pset_chunk, hid_t, int, const hsize_t*);
    H5(herr_t, H5Pset_filter, hid_t, int, unsigned, size_t, const unsigned*);
    H5(hid_t, H5Dcreate2, hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t);
    H5(hid_t, H5Dopen2, hid_t, const char*, hid_t);
    H5(herr_t, H5Dclose, hid_t);
    H5(herr_t, H5Dwrite, hid_t, hid_t, hid_t, hid_t, hid_t, const void*);
    H5(herr_t, H5Dread, hid_t, hid_t, hid_t, hid_t, hid_t, void*);
    H5(hsize_t, H5Dget_storage_size, hid_t);
    const hid_t type = *sym<hid_t*>(type_global<T>());
    const hid_t dcpl_class = *sym<hid_t*>("H5P_CLS_DATASET_CREATE_ID_g");

    const hsize_t count = data.size(), chunk = count / 8;  // vbz_hdf_plugin_test.cpp:31,85
    const unsigned cd[4] = { 1u, (unsigned)sizeof(T), 1u, level };  // vbz_filter_enable(plist, sizeof(T), true, level), user_utils.h:17-52
    const hid_t file = H5Fcreate(path, 2 /* H5F_ACC_TRUNC */, 0, 0);
    const hid_t space = H5Screate_simple(1, &count, nullptr), dcpl = H5Pcreate(dcpl_class);
    bool ok = file >= 0 && space >= 0 && dcpl >= 0 && H5Pset_chunk(dcpl, 1, &chunk) >= 0 &&
              H5Pset_filter(dcpl, FILTER_VBZ_ID, 0 /* mandatory, as vbz_filter() sets it */, 4, cd) >= 0;
    hid_t ds = ok ? H5Dcreate2(file, "foo", type, space, 0, dcpl, 0) : -1;
    ok = ok && ds >= 0 && H5Dwrite(ds, type, 0, 0, 0, data.data()) >= 0;
    if (ds >= 0) ok = H5Dclose(ds) >= 0 && ok;  // the chunk cache is flushed (and the filter called) here at the latest
    if (dcpl >= 0) H5Pclose(dcpl);
    if (space >= 0) H5Sclose(space);
    if (file >= 0) ok = H5Fclose(file) >= 0 && ok;
    if (!ok) {
        std::fprintf(stderr, "write failed (%zu-byte type, %llu values)\n", sizeof(T), count);
        return false;
    }
    const hid_t again = H5Fopen(path, 0 /* H5F_ACC_RDONLY */, 0);
    ds = again >= 0 ? H5Dopen2(again, "foo", 0) : -1;
    std::vector<T> back(data.size());
    ok = ds >= 0 && H5Dread(ds, type, 0, 0, 0, back.data()) >= 0 && std::memcmp(back.data(), data.data(), data.size() * sizeof(T)) == 0;
    const hsize_t stored = ds >= 0 ? H5Dget_storage_size(ds) : 0;
    if (ds >= 0) H5Dclose(ds);
    if (again >= 0) H5Fclose(again);
    if (!ok) std::fprintf(stderr, "read back differs (%zu-byte type, %llu values)\n", sizeof(T), count);
    if (ok && stored == 0) ok = false;
    std::printf("%s%d_t x %llu, level %u: %llu -> %llu bytes\n", std::is_signed<T>::value ? "int" : "uint", (int)(8 * sizeof(T)), count, level,
                count * (hsize_t)sizeof(T), stored);
    return ok;
}

template <typename T>
static bool both(const char* path, size_t random_count)
{
    std::vector<T> iota(100);  // vbz_hdf_plugin_test.cpp:102-117
    for (size_t i = 0; i < iota.size(); ++i) iota[i] = (T)i;
    std::vector<T> noise(random_count);  // :119-136: values in [min / 2, max / 2]
    uint64_t x = 0x9E3779B97F4A7C15ull ^ sizeof(T) ^ (std::is_signed<T>::value ? 77 : 0);
    const int64_t lo = std::numeric_limits<T>::min() / 2, hi = std::numeric_limits<T>::max() / 2;
    for (T& v : noise) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        v = (T)(lo + (int64_t)(x % (uint64_t)(hi - lo + 1)));
    }
    return run<T>(path, iota, 5) && run<T>(path, noise, 1);
}

int main(int argc, char** argv)
{
    const char* names[] = { argc > 2 ? argv[2] : "libhdf5.so", "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310",
                            "/opt/conda/lib/libhdf5.so" };
    for (const char* n : names)
        if ((h5 = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h5) {
        std::fprintf(stderr, "no libhdf5\n");
        return 3;
    }
    const std::string path = argc > 1 ? argv[1] : "./test_file.h5";
    sym<herr_t (*)()>("H5open")();
    if (sym<herr_t (*)(const void*)>("H5Zregister")(vbz_plugin_info()) < 0) return 1;  // vbz_plugin_user_utils.h:54-62
    const size_t n = 10000000;  // vbz_hdf_plugin_test.cpp:119-136: 10 M random values, 8 chunks of 1.25 M values each
    const bool ok = both<int8_t>(path.c_str(), n) && both<uint8_t>(path.c_str(), n) && both<int16_t>(path.c_str(), n) &&
                    both<uint16_t>(path.c_str(), n) && both<int32_t>(path.c_str(), n) && both<uint32_t>(path.c_str(), n);
    std::printf(ok ? "ok\n" : "FAILED\n");
    return ok ? 0 : 1;
}
