// vbz_plugin.cpp -- HDF5 filter 32020 on top of the MI355X codec (include/vbz_hdf_plugin.h).
//
// Mirrors the reference's vbz_plugin/vbz_plugin.cpp:97-261 (POSIX branch: malloc/free buffers) and
// calls the same sized C API the reference plugin calls -- here served by libvbz_hip.so.
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "../../include/vbz.h"
#include "../../include/vbz_hdf_plugin.h"

extern "C" {

size_t vbz_filter(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t /*nbytes*/, size_t* buf_size,
                  void** buf)
{
    if (cd_nelmts < 3) return 0;  // reference :109-112
    CompressionOptions options;
    options.vbz_version = cd_values[FILTER_VBZ_VERSION_OPTION];
    options.integer_size = cd_values[FILTER_VBZ_INTEGER_SIZE_OPTION];
    options.perform_delta_zig_zag = cd_values[FILTER_VBZ_USE_DELTA_ZIG_ZAG_COMPRESSION] != 0;
    options.zstd_compression_level = 1;  // reference :118-122
    if (cd_nelmts > FILTER_VBZ_ZSTD_COMPRESSION_LEVEL_OPTION)
        options.zstd_compression_level = cd_values[FILTER_VBZ_ZSTD_COMPRESSION_LEVEL_OPTION];

    if (*buf_size > std::numeric_limits<vbz_size_t>::max()) {
        fprintf(stderr, "vbz_filter: Chunk size too large.\n");
        return 0;
    }
    void* out = nullptr;
    vbz_size_t out_alloc = 0, used = 0;
    if (flags & H5Z_FLAG_REVERSE) {  // reference :136-182
        const vbz_size_t expected = vbz_decompressed_size(*buf, (vbz_size_t)*buf_size, &options);
        if (vbz_is_error(expected)) {
            fprintf(stderr, "vbz_filter: size error\n");
            return 0;
        }
        out_alloc = expected;
        out = malloc(expected ? expected : 1);
        if (!out) return 0;
        used = vbz_decompress_sized(*buf, (vbz_size_t)*buf_size, out, expected, &options);
        if (vbz_is_error(used)) {
            fprintf(stderr, "vbz_filter: compression error (%s)\n", vbz_error_string(used));
            free(out);
            return 0;
        }
        if (used != expected) {
            fprintf(stderr, "vbz_filter: decompressed size error\n");
            free(out);
            return 0;
        }
    } else {  // reference :183-222
        if (options.integer_size == 0 || *buf_size % options.integer_size != 0) {
            fprintf(stderr, "vbz_filter: Invalid integer_size specified\n");
            return 0;
        }
        out_alloc = vbz_max_compressed_size((vbz_size_t)*buf_size, &options);
        if (vbz_is_error(out_alloc)) return 0;
        out = malloc(out_alloc);
        if (!out) return 0;
        used = vbz_compress_sized(*buf, (vbz_size_t)*buf_size, out, out_alloc, &options);
        if (vbz_is_error(used)) {
            fprintf(stderr, "vbz_filter: compression error (%s)\n", vbz_error_string(used));
            free(out);
            return 0;
        }
    }
    free(*buf);  // reference :225-228
    *buf = out;
    *buf_size = out_alloc;  // the reference leaves 0 here on the decode branch; HDF5 accepts either
    return used;
}

static const vbz_H5Z_class2_t vbz_filter_struct = {
    1,              // H5Z_CLASS_T_VERS
    FILTER_VBZ_ID,  // id
    1,              // encoder_present
    1,              // decoder_present
    "vbz",          // name
    nullptr,        // can_apply
    nullptr,        // set_local
    vbz_filter      // filter
};

const void* vbz_plugin_info(void) { return &vbz_filter_struct; }
int H5PLget_plugin_type(void) { return VBZ_H5PL_TYPE_FILTER; }
const void* H5PLget_plugin_info(void) { return vbz_plugin_info(); }

}  // extern "C"
