/*
 * vbz_plugin_user_utils.h -- helpers for an application that links libvbz_hdf_plugin.so instead of letting libhdf5 discover it.
 *
 * The counterpart of the reference's vbz_plugin/vbz_plugin_user_utils.h:16-62 (same three names, same arguments, same
 * cd_values[] layout), so that a caller which today includes the reference header keeps compiling when it switches its
 * include path and -lvbz_hdf_plugin to this library:
 *
 *   vbz_register()                        H5Zregister(vbz_plugin_info()): 1 on success, 0 on failure
 *   vbz_filter_enable(dcpl, integer_size, use_zig_zag, zstd_compression_level)
 *   vbz_filter_enable_versioned(dcpl, integer_size, use_zig_zag, zstd_compression_level, vbz_version)
 *                                         H5Pset_filter(dcpl, 32020, 0, 4, {version, integer_size, zig_zag, level})
 *
 * integer_size 0 asks the filter to leave the element size to the caller's type (the reference's meaning); the filter itself
 * (include/vbz_hdf_plugin.h) reads cd_values exactly as the reference's vbz_filter does.  This header needs <hdf5.h>, the
 * plugin itself does not.  Plain C or C++.
 */
#ifndef VBZ_PLUGIN_USER_UTILS_H_MI355X
#define VBZ_PLUGIN_USER_UTILS_H_MI355X

#include <hdf5.h>

#include "vbz_hdf_plugin.h"

#define FILTER_VBZ_VERSION 1

static inline int vbz_filter_enable_versioned(hid_t creation_properties, unsigned int integer_size, int use_zig_zag,
                                              unsigned int zstd_compression_level, int vbz_version)
{
    unsigned int cd[4];
    cd[FILTER_VBZ_VERSION_OPTION] = (unsigned int)vbz_version;
    cd[FILTER_VBZ_INTEGER_SIZE_OPTION] = integer_size;
    cd[FILTER_VBZ_USE_DELTA_ZIG_ZAG_COMPRESSION] = use_zig_zag ? 1u : 0u;
    cd[FILTER_VBZ_ZSTD_COMPRESSION_LEVEL_OPTION] = zstd_compression_level;
    return (int)H5Pset_filter(creation_properties, FILTER_VBZ_ID, 0, 4, cd);
}

static inline int vbz_filter_enable(hid_t creation_properties, unsigned int integer_size, int use_zig_zag, unsigned int zstd_compression_level)
{
    return vbz_filter_enable_versioned(creation_properties, integer_size, use_zig_zag, zstd_compression_level, FILTER_VBZ_VERSION);
}

static inline int vbz_register(void)
{
    return H5Zregister(vbz_plugin_info()) < 0 ? 0 : 1;
}

#endif
