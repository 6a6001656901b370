// fast5_repack.cpp -- bulk (de)compression of the raw signal of fast5 files on the MI355X.
//
// The reference's tool for this job is python/fast5compress/fast5vbz.py:17-55: copy the file, then for every
// "read_*" group read Raw/Signal, delete it and create it again with filter 32020 and one chunk per read -- one
// filter call per read, serially, inside libhdf5.  This tool keeps the command line and the result (same
// dataset type, shape, chunking and filter parameters; chunks in the sized VBZ format, vbz.cpp:302-330), but moves
// the codec out of the per-chunk callback: all signals of a file are read first, coded in ONE batched call on the
// GPU (include/vbz_gpu.h), and the finished chunks are handed to libhdf5 with H5Dwrite_chunk, which stores them
// as they are.  The other direction (-d: back to gzip, as the reference's script does) reads the stored chunks
// with H5Dread_chunk and decodes them in one batch.  gzip, the other side of both conversions, is host work: chunks
// stored with deflate alone are read and written raw as well and (de)flated by a pool of threads, not one by one
// inside libhdf5.
//
//   vbz_fast5_repack [-d] [-s SUFFIX] [--vbz-version N] [--device D] FILE...      (fast5vbz.py:58-75)
//   vbz_fast5_repack --list FILE [--export-signal OUT] [--export-chunks OUT]
//   vbz_fast5_repack --samples FILE...          (one line per file: name, reads, samples -- what a work queue deals files by)
//
// Several files are a PIPELINE (the reference's users run many files side by side: README.md:36-40 `xargs -P 10`): a loader thread, the
// GPU and a writer thread work on files k + 1, k and k - 1 -- the codec is a few milliseconds of a file's 100+, the rest is libhdf5 and
// zlib on the host.  libhdf5 is not thread-safe in its usual builds: every call into it is made under one mutex, and the loader holds
// it for a file's whole load, the writer for its whole store -- so the two libhdf5 stages take turns; what overlaps them is the GPU's
// work and the inflate / deflate pools (which run outside the mutex).  Several GPUs: one process per device over a share of the file
// list (python -m vbz_compression_amd.fast5 --gpus N deals the files by their sample counts).
//
// libhdf5 (>= 1.10.3, for the direct chunk calls) is loaded at run time: --hdf5-lib PATH, $VBZ_HDF5_LIB, or the
// usual names.  Its few entry points used here are declared below with their public 1.10 signatures.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vbz.h"
#include "../../include/vbz_gpu.h"
#include "../../include/vbz_hdf_plugin.h"

namespace {

// ---- libhdf5, loaded at run time ------------------------------------------------------------------------
typedef int64_t hid_t;  // 1.10 and later
typedef int herr_t;
typedef int htri_t;
typedef unsigned long long hsize_t;
typedef long long hssize_t;

struct Hdf5
{
    void* lib = nullptr;
    herr_t (*H5open)();
    herr_t (*H5get_libversion)(unsigned*, unsigned*, unsigned*);
    herr_t (*H5Eset_auto2)(hid_t, void*, void*);
    hid_t (*H5Fopen)(const char*, unsigned, hid_t);
    herr_t (*H5Fclose)(hid_t);
    herr_t (*H5Gget_info)(hid_t, void*);
    ssize_t (*H5Lget_name_by_idx)(hid_t, const char*, int, int, hsize_t, char*, size_t, hid_t);
    htri_t (*H5Lexists)(hid_t, const char*, hid_t);
    herr_t (*H5Ldelete)(hid_t, const char*, hid_t);
    hid_t (*H5Dopen2)(hid_t, const char*, hid_t);
    hid_t (*H5Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t);
    herr_t (*H5Dclose)(hid_t);
    hid_t (*H5Dget_space)(hid_t);
    hid_t (*H5Dget_type)(hid_t);
    hid_t (*H5Dget_create_plist)(hid_t);
    hsize_t (*H5Dget_storage_size)(hid_t);
    herr_t (*H5Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*);
    herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*);
    herr_t (*H5Dread_chunk)(hid_t, hid_t, const hsize_t*, uint32_t*, void*);
    herr_t (*H5Dwrite_chunk)(hid_t, hid_t, uint32_t, const hsize_t*, size_t, const void*);
    herr_t (*H5Dget_chunk_storage_size)(hid_t, const hsize_t*, hsize_t*);
    hid_t (*H5Screate_simple)(int, const hsize_t*, const hsize_t*);
    herr_t (*H5Sclose)(hid_t);
    hssize_t (*H5Sget_simple_extent_npoints)(hid_t);
    int (*H5Sget_simple_extent_ndims)(hid_t);
    size_t (*H5Tget_size)(hid_t);
    int (*H5Tget_class)(hid_t);
    herr_t (*H5Tclose)(hid_t);
    hid_t (*H5Pcreate)(hid_t);
    herr_t (*H5Pclose)(hid_t);
    herr_t (*H5Pset_chunk)(hid_t, int, const hsize_t*);
    int (*H5Pget_chunk)(hid_t, int, hsize_t*);
    int (*H5Pget_layout)(hid_t);
    herr_t (*H5Pset_deflate)(hid_t, unsigned);
    herr_t (*H5Pset_filter)(hid_t, int, unsigned, size_t, const unsigned*);
    int (*H5Pget_nfilters)(hid_t);
    int (*H5Pget_filter2)(hid_t, unsigned, unsigned*, size_t*, unsigned*, size_t, char*, unsigned*);
    herr_t (*H5Zregister)(const void*);
    hid_t dataset_create_class = 0;  // H5P_DATASET_CREATE
};

// libhdf5's usual builds are not thread-safe: whichever thread calls into it holds this mutex
std::mutex& g_h5()
{
    static std::mutex m;
    return m;
}

const int H5D_CHUNKED = 2, H5T_INTEGER = 0, H5Z_FILTER_DEFLATE = 1;
const unsigned H5F_ACC_RDONLY = 0, H5F_ACC_RDWR = 1, H5Z_FLAG_OPTIONAL = 1;

bool load_hdf5(Hdf5& h, const char* wanted)
{
    std::vector<std::string> names;
    if (wanted) names.push_back(wanted);
    if (const char* e = getenv("VBZ_HDF5_LIB")) names.push_back(e);
    for (const char* n : { "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.310", "libhdf5.so.200", "libhdf5_serial.so.103", "libhdf5.so.103",
                           "/opt/conda/lib/libhdf5.so" })
        names.push_back(n);
    for (const std::string& n : names) {
        h.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h.lib) break;
    }
    if (!h.lib) {
        fprintf(stderr, "vbz_fast5_repack: no libhdf5 found (use --hdf5-lib PATH or VBZ_HDF5_LIB)\n");
        return false;
    }
    bool ok = true;
#define SYM(name)                                                      \
    do {                                                               \
        *(void**)(&h.name) = dlsym(h.lib, #name);                      \
        if (!h.name) {                                                 \
            fprintf(stderr, "vbz_fast5_repack: libhdf5 lacks %s\n", #name); \
            ok = false;                                                \
        }                                                              \
    } while (0)
    SYM(H5open); SYM(H5get_libversion); SYM(H5Eset_auto2); SYM(H5Fopen); SYM(H5Fclose); SYM(H5Gget_info); SYM(H5Lget_name_by_idx);
    SYM(H5Lexists); SYM(H5Ldelete); SYM(H5Dopen2); SYM(H5Dcreate2); SYM(H5Dclose); SYM(H5Dget_space); SYM(H5Dget_type);
    SYM(H5Dget_create_plist); SYM(H5Dget_storage_size); SYM(H5Dread); SYM(H5Dwrite); SYM(H5Dread_chunk); SYM(H5Dwrite_chunk);
    SYM(H5Dget_chunk_storage_size); SYM(H5Screate_simple); SYM(H5Sclose); SYM(H5Sget_simple_extent_npoints);
    SYM(H5Sget_simple_extent_ndims); SYM(H5Tget_size); SYM(H5Tget_class); SYM(H5Tclose); SYM(H5Pcreate); SYM(H5Pclose);
    SYM(H5Pset_chunk); SYM(H5Pget_chunk); SYM(H5Pget_layout); SYM(H5Pset_deflate); SYM(H5Pset_filter); SYM(H5Pget_nfilters);
    SYM(H5Pget_filter2); SYM(H5Zregister);
#undef SYM
    if (!ok) return false;
    unsigned maj = 0, min = 0, rel = 0;
    h.H5open();
    h.H5get_libversion(&maj, &min, &rel);
    if (maj == 1 && (min < 10 || (min == 10 && rel < 3))) {
        fprintf(stderr, "vbz_fast5_repack: libhdf5 %u.%u.%u is too old (direct chunk I/O needs 1.10.3)\n", maj, min, rel);
        return false;
    }
    hid_t* cls = (hid_t*)dlsym(h.lib, "H5P_CLS_DATASET_CREATE_ID_g");
    if (!cls) return false;
    h.dataset_create_class = *cls;
    h.H5Eset_auto2(0, nullptr, nullptr);  // failures are reported by this tool
    // reads of vbz datasets that are not one chunk go through the filter pipeline: register the filter (vbz_plugin.cpp:242-245)
    h.H5Zregister(vbz_plugin_info());
    return true;
}

// ---- one Raw/Signal dataset -----------------------------------------------------------------------------
struct Read
{
    std::string name;      // "read_..."
    uint64_t samples = 0;
    uint32_t elem = 0;     // bytes per sample
    bool vbz = false;      // stored with filter 32020 ...
    unsigned cd[4] = { 0, 0, 0, 1 };
    bool one_chunk = false;  // ... as a single chunk (read directly, decoded in the batch)
    uint64_t stored = 0;
    std::string filters;
    std::vector<uint8_t> signal;  // raw samples, file byte order
    std::vector<uint8_t> chunk;   // stored chunk (vbz, one chunk)
    bool gz = false;              // stored with deflate alone, as a single chunk: read raw, inflated by the pool
    uint64_t chunk_elems = 0;     // chunk dimension (the chunk holds this many samples, the dataset may be shorter)
};

uint64_t fnv1a64(const uint8_t* p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Timers
{
    double h5_read = 0, h5_write = 0, gpu = 0, copies = 0, zlib = 0;
};

// the read groups of a multi-read file (fast5vbz.py:38-41)
std::vector<std::string> read_groups(const Hdf5& h, hid_t file)
{
    std::vector<std::string> out;
    alignas(8) unsigned char info[64] = {};
    if (h.H5Gget_info(file, info) < 0) return out;
    hsize_t nlinks;
    memcpy(&nlinks, info + 8, sizeof nlinks);  // H5G_info_t { storage_type; hsize_t nlinks; ... }
    for (hsize_t i = 0; i < nlinks; ++i) {
        char name[512];
        const ssize_t len = h.H5Lget_name_by_idx(file, ".", 0 /* H5_INDEX_NAME */, 0 /* H5_ITER_INC */, i, name, sizeof name, 0);
        if (len > 0 && strncmp(name, "read_", 5) == 0) out.push_back(name);
    }
    return out;
}

bool load_read(const Hdf5& h, hid_t file, Read& r, Timers& t)
{
    const std::string raw = r.name + "/Raw", path = raw + "/Signal";
    if (h.H5Lexists(file, raw.c_str(), 0) <= 0 || h.H5Lexists(file, path.c_str(), 0) <= 0) return false;
    const double t0 = now_ms();
    const hid_t d = h.H5Dopen2(file, path.c_str(), 0);
    if (d < 0) return false;
    const hid_t sp = h.H5Dget_space(d), ty = h.H5Dget_type(d), pl = h.H5Dget_create_plist(d);
    bool ok = sp >= 0 && ty >= 0 && pl >= 0 && h.H5Tget_class(ty) == H5T_INTEGER;
    if (ok) {
        r.samples = (uint64_t)h.H5Sget_simple_extent_npoints(sp);
        r.elem = (uint32_t)h.H5Tget_size(ty);
        r.stored = h.H5Dget_storage_size(d);
        const int nf = h.H5Pget_nfilters(pl);
        for (int i = 0; i < nf; ++i) {
            unsigned flags = 0, cd[8] = {}, cfg = 0;
            size_t ncd = 8;
            char nm[64] = {};
            const int id = h.H5Pget_filter2(pl, (unsigned)i, &flags, &ncd, cd, sizeof nm, nm, &cfg);
            r.filters += (r.filters.empty() ? "" : ",") + std::to_string(id);
            if (id == FILTER_VBZ_ID && nf == 1 && ncd >= 3) {
                r.vbz = true;
                for (size_t k = 0; k < 4 && k < ncd; ++k) r.cd[k] = cd[k];
            }
            if (id == H5Z_FILTER_DEFLATE && nf == 1) r.gz = true;
        }
        if (r.filters.empty()) r.filters = "-";
        hsize_t cdim[8] = {};
        const bool single = h.H5Pget_layout(pl) == H5D_CHUNKED && h.H5Sget_simple_extent_ndims(sp) == 1 && h.H5Pget_chunk(pl, 8, cdim) == 1 &&
                            cdim[0] >= r.samples && r.samples > 0;
        r.one_chunk = r.vbz && single;
        r.gz = r.gz && single;
        r.chunk_elems = cdim[0];
        if ((uint64_t)r.samples * r.elem >= 0xFFFFFFF0ull) ok = false;  // vbz sizes are 32 bits (vbz.h:11)
    }
    if (ok) {
        if (r.one_chunk || r.gz) {
            const hsize_t off[1] = { 0 };
            hsize_t bytes = 0;
            uint32_t mask = 0;
            ok = h.H5Dget_chunk_storage_size(d, off, &bytes) >= 0 && bytes > 0;
            if (ok) {
                r.chunk.resize(bytes);
                ok = h.H5Dread_chunk(d, 0, off, &mask, r.chunk.data()) >= 0 && mask == 0;
            }
            if (!ok) {  // not there, or stored unfiltered: let libhdf5 read it
                r.one_chunk = false;
                r.gz = false;
                r.chunk.clear();
                ok = true;
            }
        }
        if (!r.one_chunk && !r.gz) {
            r.signal.resize(r.samples * r.elem);
            // memory type = file type: no conversion, the bytes as stored (little endian in every fast5)
            if (r.samples) ok = h.H5Dread(d, ty, 0, 0, 0, r.signal.data()) >= 0;
        }
    }
    if (pl >= 0) h.H5Pclose(pl);
    if (ty >= 0) h.H5Tclose(ty);
    if (sp >= 0) h.H5Sclose(sp);
    h.H5Dclose(d);
    t.h5_read += now_ms() - t0;
    if (!ok) fprintf(stderr, "vbz_fast5_repack: cannot read %s\n", path.c_str());
    return ok;
}

// ---- gzip on the host's cores -----------------------------------------------------------------------------
template <typename F>
void parallel_for(size_t n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 64 ? 64 : nt);
    if (nt > n) nt = (unsigned)n;
    std::atomic<size_t> next{ 0 };
    std::vector<std::thread> pool;
    auto work = [&] {
        for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i);
    };
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

// chunks that were read raw because deflate is their only filter: inflate them all at once.  A chunk that does not
// inflate to its size is read again through libhdf5 (which also is what reports a damaged file).
bool inflate_chunks(const Hdf5& h, hid_t file, std::vector<Read>& reads, Timers& t)
{
    const double t0 = now_ms();
    std::vector<size_t> todo;
    for (size_t i = 0; i < reads.size(); ++i)
        if (reads[i].gz) todo.push_back(i);
    std::vector<char> good(todo.size(), 0);
    parallel_for(todo.size(), [&](size_t k) {
        Read& r = reads[todo[k]];
        std::vector<uint8_t> full(r.chunk_elems * r.elem);
        uLongf len = (uLongf)full.size();
        if (uncompress(full.data(), &len, r.chunk.data(), (uLong)r.chunk.size()) == Z_OK && len == full.size()) {
            full.resize(r.samples * r.elem);
            r.signal.swap(full);
            good[k] = 1;
        }
    });
    t.zlib += now_ms() - t0;
    bool ok = true;
    for (size_t k = 0; k < todo.size(); ++k) {
        Read& r = reads[todo[k]];
        r.chunk.clear();
        r.gz = false;
        if (good[k]) continue;
        const std::string path = r.name + "/Raw/Signal";
        std::lock_guard<std::mutex> lock(g_h5());
        const hid_t d = h.H5Dopen2(file, path.c_str(), 0);
        const hid_t ty = d >= 0 ? h.H5Dget_type(d) : -1;
        r.signal.resize(r.samples * r.elem);
        if (d < 0 || ty < 0 || h.H5Dread(d, ty, 0, 0, 0, r.signal.data()) < 0) {
            fprintf(stderr, "vbz_fast5_repack: cannot read %s\n", path.c_str());
            ok = false;
        }
        if (ty >= 0) h.H5Tclose(ty);
        if (d >= 0) h.H5Dclose(d);
    }
    return ok;
}

// ---- the batched codec calls ----------------------------------------------------------------------------
struct Gpu
{
    vbz_gpu_ctx* ctx = nullptr;
    int device = 0;
    bool init()
    {
        if (!ctx) {
            (void)hipSetDevice(device);   // (the tool's own hipMalloc / hipMemcpy calls go to the context's device)
            ctx = vbz_gpu_create(device, nullptr);
        }
        if (!ctx) fprintf(stderr, "vbz_fast5_repack: no usable MI355X (gfx950) device\n");
        return ctx != nullptr;
    }
    ~Gpu()
    {
        if (ctx) vbz_gpu_destroy(ctx);
    }
};

struct DevBuf
{
    void* p = nullptr;
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 16) == hipSuccess; }
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
};

// Runs one batch: in[i] -> out[i].  compress: out = sized chunks; else: out = the samples of sized chunks.
// Reads with different filter parameters go in different calls (the options are per call).
bool run_batch(Gpu& g, const std::vector<const std::vector<uint8_t>*>& in, const std::vector<std::vector<uint8_t>*>& out,
               const std::vector<uint32_t>& out_cap, const CompressionOptions& opt, bool compress, Timers& t)
{
    const uint32_t n = (uint32_t)in.size();
    if (n == 0) return true;
    std::vector<uint64_t> soff(n), doff(n);
    std::vector<uint32_t> ssize(n), res(n);
    uint64_t sbytes = 0, dbytes = 0;
    for (uint32_t i = 0; i < n; ++i) {
        soff[i] = sbytes;
        ssize[i] = (uint32_t)in[i]->size();
        sbytes += ((uint64_t)ssize[i] + 15) & ~15ull;
        doff[i] = dbytes;
        dbytes += ((uint64_t)out_cap[i] + 15) & ~15ull;
    }
    DevBuf dsrc, ddst, dmeta;
    const size_t meta = (size_t)n * (8 + 8 + 4 + 4 + 4);
    if (!dsrc.alloc(sbytes + 64) || !ddst.alloc(dbytes + 64) || !dmeta.alloc(meta)) {
        fprintf(stderr, "vbz_fast5_repack: out of device memory (%llu + %llu bytes)\n", (unsigned long long)sbytes, (unsigned long long)dbytes);
        return false;
    }
    const double t0 = now_ms();
    uint8_t* m = (uint8_t*)dmeta.p;
    uint64_t* d_soff = (uint64_t*)m;
    uint64_t* d_doff = (uint64_t*)(m + 8ull * n);
    uint32_t* d_ssize = (uint32_t*)(m + 16ull * n);
    uint32_t* d_cap = (uint32_t*)(m + 20ull * n);
    uint32_t* d_res = (uint32_t*)(m + 24ull * n);
    bool ok = hipMemcpy(d_soff, soff.data(), 8ull * n, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_doff, doff.data(), 8ull * n, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_ssize, ssize.data(), 4ull * n, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_cap, out_cap.data(), 4ull * n, hipMemcpyHostToDevice) == hipSuccess;
    {  // one arena, one copy
        std::vector<uint8_t> arena(sbytes);
        for (uint32_t i = 0; i < n; ++i)
            if (ssize[i]) memcpy(arena.data() + soff[i], in[i]->data(), ssize[i]);
        if (ok && sbytes) ok = hipMemcpy(dsrc.p, arena.data(), sbytes, hipMemcpyHostToDevice) == hipSuccess;
    }
    t.copies += now_ms() - t0;
    if (!ok) return false;
    vbz_gpu_batch b = {};
    b.n_reads = n;
    b.src = dsrc.p;
    b.src_off = d_soff;
    b.src_size = d_ssize;
    b.src_bytes = sbytes;
    b.dst = ddst.p;
    b.dst_off = d_doff;
    b.dst_cap = d_cap;
    b.dst_bytes = dbytes;
    b.result = d_res;
    const double t1 = now_ms();
    const int rc = compress ? vbz_gpu_compress_batch(g.ctx, &b, &opt, 1) : vbz_gpu_decompress_batch(g.ctx, &b, &opt, 1);
    if (rc != 0 || vbz_gpu_synchronize(g.ctx) != 0) {
        fprintf(stderr, "vbz_fast5_repack: %s\n", vbz_gpu_last_error(g.ctx));
        return false;
    }
    t.gpu += now_ms() - t1;
    const double t2 = now_ms();
    ok = hipMemcpy(res.data(), d_res, 4ull * n, hipMemcpyDeviceToHost) == hipSuccess;
    for (uint32_t i = 0; ok && i < n; ++i) {
        if (vbz_is_error(res[i])) {
            fprintf(stderr, "vbz_fast5_repack: read %u: %s\n", i, vbz_error_string(res[i]));
            return false;
        }
        out[i]->resize(res[i]);
        if (res[i]) ok = hipMemcpy(out[i]->data(), (uint8_t*)ddst.p + doff[i], res[i], hipMemcpyDeviceToHost) == hipSuccess;
    }
    t.copies += now_ms() - t2;
    return ok;
}

CompressionOptions options_of(const unsigned cd[4])
{
    CompressionOptions o;
    o.vbz_version = cd[FILTER_VBZ_VERSION_OPTION];
    o.integer_size = cd[FILTER_VBZ_INTEGER_SIZE_OPTION];
    o.perform_delta_zig_zag = cd[FILTER_VBZ_USE_DELTA_ZIG_ZAG_COMPRESSION] != 0;
    o.zstd_compression_level = cd[FILTER_VBZ_ZSTD_COMPRESSION_LEVEL_OPTION];
    return o;
}

// every read that was loaded as a stored vbz chunk gets its samples
bool decode_chunks(Gpu& g, std::vector<Read>& reads, Timers& t)
{
    std::vector<bool> done(reads.size(), false);
    for (size_t a = 0; a < reads.size(); ++a) {
        if (done[a] || !reads[a].one_chunk) continue;
        std::vector<const std::vector<uint8_t>*> in;
        std::vector<std::vector<uint8_t>*> out;
        std::vector<uint32_t> cap;
        for (size_t k = a; k < reads.size(); ++k) {
            if (done[k] || !reads[k].one_chunk || memcmp(reads[k].cd, reads[a].cd, sizeof reads[a].cd) != 0) continue;
            done[k] = true;
            in.push_back(&reads[k].chunk);
            out.push_back(&reads[k].signal);
            cap.push_back((uint32_t)(reads[k].samples * reads[k].elem));
        }
        if (!g.init() || !run_batch(g, in, out, cap, options_of(reads[a].cd), false, t)) return false;
    }
    for (const Read& r : reads)
        if (r.signal.size() != r.samples * r.elem) {
            fprintf(stderr, "vbz_fast5_repack: %s: chunk does not hold %llu samples\n", r.name.c_str(), (unsigned long long)r.samples);
            return false;
        }
    return true;
}

// ---- the commands ----------------------------------------------------------------------------------------
// (every call into libhdf5 is made under g_h5(): see its definition)

// One file on its way through the three stages of compress_fast5(filename, output_suffix, vbz_version, decompress)
// (fast5vbz.py:17-55): load (copy the file, read every read's stored chunk or samples, inflate) -> code (the GPU: decode stored vbz
// chunks, code the new ones; or gzip on the host's cores) -> store (the datasets deleted and created again, the chunks written as
// they are).
struct Job
{
    std::string in_name, out_name;
    hid_t file = -1;
    std::vector<Read> reads;
    std::vector<std::vector<uint8_t>> packed;
    Timers t;
    double t_start = 0;
    bool ok = false;
};

bool job_load(const Hdf5& h, Job& j, const std::string& filename, const std::string& suffix)
{
    namespace fs = std::filesystem;
    j.t_start = now_ms();
    std::error_code ec;
    j.in_name = fs::absolute(filename, ec).string();
    j.out_name = j.in_name + suffix;
    fs::copy_file(j.in_name, j.out_name, fs::copy_options::overwrite_existing, ec);
    if (ec) {
        fprintf(stderr, "vbz_fast5_repack: cannot copy %s: %s\n", j.in_name.c_str(), ec.message().c_str());
        return false;
    }
    fs::permissions(j.out_name, fs::perms::owner_write, fs::perm_options::add, ec);
    {
        std::lock_guard<std::mutex> lock(g_h5());
        j.file = h.H5Fopen(j.out_name.c_str(), H5F_ACC_RDWR, 0);
        if (j.file < 0) {
            fprintf(stderr, "vbz_fast5_repack: cannot open %s\n", j.out_name.c_str());
            return false;
        }
        for (const std::string& name : read_groups(h, j.file)) {
            Read r;
            r.name = name;
            if (load_read(h, j.file, r, j.t)) j.reads.push_back(std::move(r));
        }
    }
    return inflate_chunks(h, j.file, j.reads, j.t);   // (zlib on a pool of threads; its fall-back into libhdf5 takes the mutex itself)
}

bool job_code(Gpu& g, Job& j, unsigned vbz_version, bool decompress)
{
    std::vector<Read>& reads = j.reads;
    Timers& t = j.t;
    bool ok = decode_chunks(g, reads, t);
    // the new chunks: 2-byte (here: elem-byte) integers with zig-zag, level 1 zstd (fast5vbz.py:33-36)
    j.packed.assign(reads.size(), std::vector<uint8_t>());
    std::vector<std::vector<uint8_t>>& packed = j.packed;
    if (ok && !decompress) {
        std::vector<bool> done(reads.size(), false);
        for (size_t a = 0; ok && a < reads.size(); ++a) {
            if (done[a]) continue;
            const unsigned cd[4] = { vbz_version, reads[a].elem, 1, 1 };
            const CompressionOptions opt = options_of(cd);
            std::vector<const std::vector<uint8_t>*> in;
            std::vector<std::vector<uint8_t>*> out;
            std::vector<uint32_t> cap;
            for (size_t k = a; k < reads.size(); ++k) {
                if (done[k] || reads[k].elem != reads[a].elem) continue;
                done[k] = true;
                if (reads[k].samples == 0) continue;
                in.push_back(&reads[k].signal);
                out.push_back(&packed[k]);
                cap.push_back(vbz_max_compressed_size((vbz_size_t)reads[k].signal.size(), &opt));
            }
            ok = g.init() && run_batch(g, in, out, cap, opt, true, t);
        }
    }
    if (ok && decompress) {  // gzip level 1 (fast5vbz.py:30), every read on its own core
        const double t0 = now_ms();
        std::vector<char> good(reads.size(), 1);
        parallel_for(reads.size(), [&](size_t k) {
            const Read& r = reads[k];
            if (r.samples == 0) return;
            uLongf len = compressBound((uLong)r.signal.size());
            packed[k].resize(len);
            if (compress2(packed[k].data(), &len, r.signal.data(), (uLong)r.signal.size(), 1) != Z_OK) good[k] = 0;
            packed[k].resize(len);
        });
        for (char gk : good) ok = ok && gk;
        t.zlib += now_ms() - t0;
    }
    return ok;
}

bool job_store(const Hdf5& h, Job& j, unsigned vbz_version, bool decompress, bool coded)
{
    std::lock_guard<std::mutex> lock(g_h5());
    const hid_t file = j.file;
    std::vector<Read>& reads = j.reads;
    std::vector<std::vector<uint8_t>>& packed = j.packed;
    Timers& t = j.t;
    bool ok = coded;
    uint64_t raw_bytes = 0, new_bytes = 0;
    for (size_t k = 0; ok && k < reads.size(); ++k) {
        const Read& r = reads[k];
        const double t0 = now_ms();
        const std::string path = r.name + "/Raw/Signal";
        hid_t d = h.H5Dopen2(file, path.c_str(), 0);
        const hid_t ty = d >= 0 ? h.H5Dget_type(d) : -1;
        if (d >= 0) h.H5Dclose(d);
        ok = ty >= 0 && h.H5Ldelete(file, path.c_str(), 0) >= 0;  // fast5vbz.py:47-48
        if (ok) {
            const hsize_t dims[1] = { r.samples };
            const hid_t sp = h.H5Screate_simple(1, dims, nullptr), pl = h.H5Pcreate(h.dataset_create_class);
            if (r.samples) {  // chunks=(len(raw),)  (fast5vbz.py:51-53); an empty signal cannot be chunked
                h.H5Pset_chunk(pl, 1, dims);
                if (decompress) {
                    h.H5Pset_deflate(pl, 1);  // fast5vbz.py:30
                } else {
                    const unsigned cd[4] = { vbz_version, r.elem, 1, 1 };
                    h.H5Pset_filter(pl, FILTER_VBZ_ID, H5Z_FLAG_OPTIONAL, 4, cd);  // as h5py sets an integer filter id
                }
            }
            d = h.H5Dcreate2(file, path.c_str(), ty, sp, 0, pl, 0);
            ok = d >= 0;
            if (ok && r.samples) {
                const hsize_t off[1] = { 0 };
                // the chunk is stored as it is: filter mask 0 = every filter of the pipeline has been applied
                ok = h.H5Dwrite_chunk(d, 0, 0, off, packed[k].size(), packed[k].data()) >= 0;
            }
            if (d >= 0) {
                new_bytes += h.H5Dget_storage_size(d);
                h.H5Dclose(d);
            }
            h.H5Pclose(pl);
            h.H5Sclose(sp);
        }
        if (ty >= 0) h.H5Tclose(ty);
        raw_bytes += r.signal.size();
        t.h5_write += now_ms() - t0;
        if (!ok) fprintf(stderr, "vbz_fast5_repack: cannot rewrite %s\n", path.c_str());
    }
    if (file >= 0) ok = h.H5Fclose(file) >= 0 && ok;
    j.file = -1;
    if (ok) {
        printf("%s\n", j.out_name.c_str());  // fast5vbz.py:60-64
        fflush(stdout);
        fprintf(stderr,
                "vbz_fast5_repack: %zu reads, %llu raw bytes -> %llu stored; hdf5 read %.1f ms, gzip on the host %.1f ms, "
                "host<->device %.1f ms, codec %.1f ms, hdf5 write %.1f ms, in the pipeline %.1f ms\n",
                reads.size(), (unsigned long long)raw_bytes, (unsigned long long)new_bytes, t.h5_read, t.zlib, t.copies, t.gpu, t.h5_write,
                now_ms() - j.t_start);
    }
    return ok;
}

// a bounded hand-over between two stages
template <typename T>
struct Channel
{
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    size_t cap;
    bool closed = false;
    explicit Channel(size_t c) : cap(c) {}
    void put(T v)
    {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return q.size() < cap; });
        q.push_back(std::move(v));
        cv.notify_all();
    }
    bool get(T& v)
    {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return true;
    }
    void close()
    {
        std::lock_guard<std::mutex> lock(m);
        closed = true;
        cv.notify_all();
    }
};

// every file through load -> code -> store, the three stages on threads of their own (one file in each at any time, one more
// waiting between them); results in the order of the command line, like the reference's loop (fast5vbz.py:72-75)
int repack_files(const Hdf5& h, Gpu& g, const std::vector<std::string>& files, const std::string& suffix, unsigned vbz_version, bool decompress)
{
    Channel<Job*> loaded(1), coded(1);
    std::atomic<int> failed{ 0 };
    const double t0 = now_ms();
    std::thread loader([&] {
        for (const std::string& f : files) {
            Job* j = new Job();
            j->ok = job_load(h, *j, f, suffix);
            loaded.put(j);
        }
        loaded.close();
    });
    std::thread writer([&] {
        Job* j;
        while (coded.get(j)) {
            if (!job_store(h, *j, vbz_version, decompress, j->ok)) failed++;
            if (j->file >= 0) {   // (a job that failed before the store stage still holds its file)
                std::lock_guard<std::mutex> lock(g_h5());
                h.H5Fclose(j->file);
            }
            delete j;
        }
    });
    Job* j;
    while (loaded.get(j)) {
        if (j->ok) j->ok = job_code(g, *j, vbz_version, decompress);
        coded.put(j);
    }
    coded.close();
    loader.join();
    writer.join();
    if (files.size() > 1)
        fprintf(stderr, "vbz_fast5_repack: %zu files in %.1f ms (%d failed)\n", files.size(), now_ms() - t0, failed.load());
    return failed.load() ? 1 : 0;
}

// one line per file: name, read_* groups with a signal, samples in all (dataset extents only: nothing is read or decoded)
bool count_samples(const Hdf5& h, const std::string& filename)
{
    const hid_t file = h.H5Fopen(filename.c_str(), H5F_ACC_RDONLY, 0);
    if (file < 0) {
        fprintf(stderr, "vbz_fast5_repack: cannot open %s\n", filename.c_str());
        return false;
    }
    uint64_t reads = 0, samples = 0;
    for (const std::string& name : read_groups(h, file)) {
        const std::string raw = name + "/Raw", path = raw + "/Signal";
        if (h.H5Lexists(file, raw.c_str(), 0) <= 0 || h.H5Lexists(file, path.c_str(), 0) <= 0) continue;
        const hid_t d = h.H5Dopen2(file, path.c_str(), 0);
        if (d < 0) continue;
        const hid_t sp = h.H5Dget_space(d);
        if (sp >= 0) {
            ++reads;
            samples += (uint64_t)h.H5Sget_simple_extent_npoints(sp);
            h.H5Sclose(sp);
        }
        h.H5Dclose(d);
    }
    h.H5Fclose(file);
    printf("%s\t%llu\t%llu\n", filename.c_str(), (unsigned long long)reads, (unsigned long long)samples);
    return true;
}

// one line per read: name, samples, bytes per sample, filter ids, stored bytes, FNV-1a-64 of the samples
bool list_fast5(const Hdf5& h, Gpu& g, const std::string& filename, const char* export_signal, const char* export_chunks)
{
    Timers t;
    const hid_t file = h.H5Fopen(filename.c_str(), H5F_ACC_RDONLY, 0);
    if (file < 0) {
        fprintf(stderr, "vbz_fast5_repack: cannot open %s\n", filename.c_str());
        return false;
    }
    std::vector<Read> reads;
    for (const std::string& name : read_groups(h, file)) {
        Read r;
        r.name = name;
        if (load_read(h, file, r, t)) reads.push_back(std::move(r));
    }
    const bool inflated = inflate_chunks(h, file, reads, t);
    h.H5Fclose(file);
    if (!inflated || !decode_chunks(g, reads, t)) return false;
    FILE* fs = export_signal ? fopen(export_signal, "wb") : nullptr;
    FILE* fc = export_chunks ? fopen(export_chunks, "wb") : nullptr;
    if ((export_signal && !fs) || (export_chunks && !fc)) return false;
    for (const Read& r : reads) {
        printf("%s\t%llu\t%u\t%s\t%llu\t%016llx\t%zu\n", r.name.c_str(), (unsigned long long)r.samples, r.elem, r.filters.c_str(),
               (unsigned long long)r.stored, (unsigned long long)fnv1a64(r.signal.data(), r.signal.size()), r.chunk.size());
        if (fs) fwrite(r.signal.data(), 1, r.signal.size(), fs);
        if (fc) fwrite(r.chunk.data(), 1, r.chunk.size(), fc);
    }
    if (fs) fclose(fs);
    if (fc) fclose(fc);
    return true;
}

void usage()
{
    fprintf(stderr,
            "usage: vbz_fast5_repack [-d] [-s SUFFIX] [--vbz-version N] [--device D] [--hdf5-lib PATH] FILE...\n"
            "       vbz_fast5_repack --list FILE [--export-signal OUT] [--export-chunks OUT] [--hdf5-lib PATH]\n"
            "       vbz_fast5_repack --samples FILE...      (name, reads, samples of every file; no GPU needed)\n"
            "  --device D            the GPU to use (default 0, or $VBZ_HIP_DEVICE)\n"
            "  -d, --decompress      store the signal with gzip level 1 instead of vbz\n"
            "  -s, --output-suffix   appended to the name of the copy that is rewritten (default .tmp)\n"
            "  --vbz-version N       0 or 1 (default 1)\n");
}

}  // namespace

int main(int argc, char** argv)
{
    bool decompress = false, list = false, samples = false;
    int device = getenv("VBZ_HIP_DEVICE") ? atoi(getenv("VBZ_HIP_DEVICE")) : 0;
    std::string suffix = ".tmp";
    unsigned vbz_version = 1;
    const char *hdf5_lib = nullptr, *export_signal = nullptr, *export_chunks = nullptr;
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        if (a == "-d" || a == "--decompress") decompress = true;
        else if (a == "-s" || a == "--output-suffix") { const char* v = value(); if (!v) { usage(); return 2; } suffix = v; }
        else if (a == "--vbz-version") { const char* v = value(); if (!v) { usage(); return 2; } vbz_version = (unsigned)atoi(v); }
        else if (a == "--hdf5-lib") hdf5_lib = value();
        else if (a == "--list") list = true;
        else if (a == "--samples") samples = true;
        else if (a == "--device") { const char* v = value(); if (!v) { usage(); return 2; } device = atoi(v); }
        else if (a == "--export-signal") export_signal = value();
        else if (a == "--export-chunks") export_chunks = value();
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a == "-v" || a == "--version") { printf("%s\n", vbz_gpu_version()); return 0; }
        else if (!a.empty() && a[0] == '-') { usage(); return 2; }
        else files.push_back(a);
    }
    if (files.empty() || (list && files.size() != 1) || vbz_version > 1) {
        usage();
        return 2;
    }
    Hdf5 h;
    if (!load_hdf5(h, hdf5_lib)) return 3;
    if (samples) {
        int rc = 0;
        for (const std::string& f : files)
            if (!count_samples(h, f)) rc = 1;
        return rc;
    }
    Gpu g;
    g.device = device;
    if (list) return list_fast5(h, g, files[0], export_signal, export_chunks) ? 0 : 1;
    return repack_files(h, g, files, suffix, vbz_version, decompress);
}
