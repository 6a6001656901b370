/*
 * vbz_hdf_plugin.h -- HDF5 filter 32020 ("vbz") entry points of libvbz_hdf_plugin.so.
 *
 * Drop-in for the reference's HDF5 plugin: libhdf5 dlopens the shared object found on
 * HDF5_PLUGIN_PATH and calls H5PLget_plugin_type / H5PLget_plugin_info; an application that links it
 * can register the filter itself with H5Zregister(vbz_plugin_info()).
 *
 *   reference vbz_plugin/vbz_plugin.cpp:97-229   vbz_filter (the H5Z_func_t)
 *   reference vbz_plugin/vbz_plugin.cpp:231-261  vbz_filter_struct, vbz_plugin_info, H5PLget_plugin_*
 *   reference vbz_plugin/vbz_plugin.h:5-10       FILTER_VBZ_ID and the cd_values[] layout
 *   reference third_party/hdf5/hdf5_plugin_types.h:58-75  H5Z_class2_t (restated below so that the
 *                                                plugin builds without HDF5 headers, as the reference does)
 *
 * cd_values = [vbz_version, integer_size, use_zig_zag, zstd_level]; cd_nelmts >= 3, the level
 * defaults to 1.  One call handles one chunk: flags & H5Z_FLAG_REVERSE decodes.  The chunk format is the
 * "sized" format of vbz.h ([u32 LE original size][payload]).  On success the old *buf is freed with
 * free(), replaced by a malloc()ed buffer, and the number of valid bytes is returned; on failure 0 is
 * returned and the arguments are left untouched.  The codec work runs on the MI355X through vbz.h.
 */
#ifndef VBZ_HDF_PLUGIN_H_MI355X
#define VBZ_HDF_PLUGIN_H_MI355X

#include <stddef.h>

#if defined(__cplusplus)
extern "C" {
#endif

#ifndef VBZ_HDF_PLUGIN_EXPORT
#define VBZ_HDF_PLUGIN_EXPORT __attribute__((visibility("default")))
#endif

#define FILTER_VBZ_ID 32020
#define FILTER_VBZ_VERSION_OPTION 0
#define FILTER_VBZ_INTEGER_SIZE_OPTION 1
#define FILTER_VBZ_USE_DELTA_ZIG_ZAG_COMPRESSION 2
#define FILTER_VBZ_ZSTD_COMPRESSION_LEVEL_OPTION 3

#ifndef H5Z_FLAG_REVERSE
#define H5Z_FLAG_REVERSE 0x0100
#endif

/* H5PL_type_t / H5Z_class2_t as HDF5 1.8+ defines them (public, stable ABI) */
typedef enum vbz_H5PL_type_t { VBZ_H5PL_TYPE_ERROR = -1, VBZ_H5PL_TYPE_FILTER = 0, VBZ_H5PL_TYPE_NONE = 1 } vbz_H5PL_type_t;
typedef size_t (*vbz_H5Z_func_t)(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes,
                                 size_t* buf_size, void** buf);
typedef struct vbz_H5Z_class2_t
{
    int version;              /* H5Z_CLASS_T_VERS == 1 */
    int id;                   /* 32020 */
    unsigned encoder_present; /* 1 */
    unsigned decoder_present; /* 1 */
    const char* name;         /* "vbz" */
    void* can_apply;          /* NULL */
    void* set_local;          /* NULL */
    vbz_H5Z_func_t filter;    /* vbz_filter */
} vbz_H5Z_class2_t;

/* replaces reference vbz_plugin.cpp:97-229 */
VBZ_HDF_PLUGIN_EXPORT size_t vbz_filter(unsigned int flags, size_t cd_nelmts, const unsigned int cd_values[], size_t nbytes,
                                        size_t* buf_size, void** buf);
/* replaces reference vbz_plugin.cpp:242-245 (static registration with H5Zregister) */
VBZ_HDF_PLUGIN_EXPORT const void* vbz_plugin_info(void);
/* replaces reference vbz_plugin.cpp:248-261 (dynamic discovery by libhdf5) */
VBZ_HDF_PLUGIN_EXPORT int H5PLget_plugin_type(void);
VBZ_HDF_PLUGIN_EXPORT const void* H5PLget_plugin_info(void);

#if defined(__cplusplus)
}
#endif
#endif
