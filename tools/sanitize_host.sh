#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST-only code of this repository, in the CPU container (the
# reference does the same through cmake/FindSanitizers.cmake:56-96; GPU-side sanitizers are not available on the pool):
#   * oracle/*.c                       -- the CPU suites that drive it (tests/test_oracle.py, test_oracle_zstd.py)
#   * tests/host/entropy_harness.cpp   -- zstd_entropy.h compiled for the host (tests/test_entropy_host.py)
#   * csrc/vbz_plugin.cpp, csrc/fast5_repack.cpp -- compiled with the sanitizers against the shipped libvbz_hip.so and run as
#     far as a box without a GPU lets them: the filter's error path (no device -> 0), the re-packer's read side (every
#     Raw/Signal dataset of the golden file read and inflated) up to the device call, which fails loudly.
# Leaves the ordinary builds in place afterwards.   bash tools/sanitize_host.sh   (exit 0 = no finding)
set -eu
cd "$(dirname "$0")/.."
ROOT=$PWD
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g -O1"
ASAN_LIB=$(gcc -print-file-name=libasan.so)
# (the python interpreter itself is not leak-clean; the fuzz replay asks malloc for the sizes hostile frames claim, as the
# reference does -- malloc returns NULL there and the path reports VBZ_OUT_OF_MEMORY_ERROR, so the allocator may too)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
restore() {
    make -C oracle -s clean; make -C oracle -s
    rm -f tests/host/libentropy_harness.so
}
trap restore EXIT
echo "== oracle/ under ASan + UBSan"
make -C oracle -s clean
make -C oracle -s CFLAGS="$SAN -fPIC -Wall -Wextra -std=c11"
echo "== tests/host/entropy_harness.cpp under ASan + UBSan"
g++ $SAN -std=c++17 -shared -fPIC -Ivbz_compression_amd/csrc -o tests/host/libentropy_harness.so tests/host/entropy_harness.cpp
LD_PRELOAD=$ASAN_LIB python -m pytest tests/test_oracle.py tests/test_oracle_zstd.py tests/test_entropy_host.py -x -q -p no:cacheprovider
echo "== csrc/vbz_plugin.cpp, csrc/fast5_repack.cpp under ASan + UBSan (no GPU here: error paths and the read side)"
TMP=$(mktemp -d)
LIBDIR=$ROOT/vbz_compression_amd/lib
ROCM=$(dirname "$(dirname "$(readlink -f /opt/rocm/bin/hipcc)")")
g++ $SAN -std=c++17 -fPIC -shared -o "$TMP/libvbz_hdf_plugin.so" vbz_compression_amd/csrc/vbz_plugin.cpp -L"$LIBDIR" -lvbz_hip -Wl,-rpath,"$LIBDIR"
g++ $SAN -std=c++17 -Wall -D__HIP_PLATFORM_AMD__ -I"$ROCM/include" -o "$TMP/vbz_fast5_repack" vbz_compression_amd/csrc/fast5_repack.cpp \
    -L"$TMP" -lvbz_hdf_plugin -L"$LIBDIR" -lvbz_hip -L"$ROCM/lib" -lamdhip64 -ldl -lz -pthread -Wl,-rpath,"$TMP" -Wl,-rpath,"$LIBDIR" -Wl,-rpath,"$ROCM/lib"
cat > "$TMP/filter_call.c" <<'EOF'
/* the H5Z callback of the plugin, called the way libhdf5 calls it (vbz_plugin/vbz_plugin.cpp:97-261), on a box without a device */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vbz_hdf_plugin.h"
int main(void)
{
    const vbz_H5Z_class2_t* cls = (const vbz_H5Z_class2_t*)H5PLget_plugin_info();
    if (!cls || cls->id != 32020 || H5PLget_plugin_type() != VBZ_H5PL_TYPE_FILTER) return 2;
    unsigned cd[5] = { 0, 2, 1, 1, 1 };
    size_t n = 20000, cap = n;
    void* buf = malloc(n);
    memset(buf, 1, n);
    size_t r = cls->filter(0, 5, cd, n, &cap, &buf);          /* compress: no device -> 0, buffer untouched */
    size_t r2 = cls->filter(0x0100, 5, cd, n, &cap, &buf);     /* decompress of garbage: 0 */
    size_t r3 = cls->filter(0, 1, cd, n, &cap, &buf);          /* too few parameters: 0 */
    free(buf);
    printf("filter results %zu %zu %zu\n", r, r2, r3);
    return (r == 0 && r2 == 0 && r3 == 0) ? 0 : 3;
}
EOF
gcc $SAN -Iinclude -o "$TMP/filter_call" "$TMP/filter_call.c" -L"$TMP" -lvbz_hdf_plugin -Wl,-rpath,"$TMP" -Wl,-rpath,"$LIBDIR"
"$TMP/filter_call" 2> "$TMP/filter.err" || { cat "$TMP/filter.err"; echo "filter_call failed"; exit 1; }
grep -q "ERROR: AddressSanitizer\|runtime error" "$TMP/filter.err" && { cat "$TMP/filter.err"; exit 1; }
set +e
cp tests/golden/multi_fast5_zip.fast5 "$TMP/in.fast5"   # (the tool writes its <name>.tmp beside the input: not into the fixture directory)
"$TMP/vbz_fast5_repack" "$TMP/in.fast5" "$TMP/out.fast5" > "$TMP/repack.out" 2> "$TMP/repack.err"
rc=$?
set -e
if grep -q "ERROR: AddressSanitizer\|runtime error" "$TMP/repack.err"; then cat "$TMP/repack.err"; exit 1; fi
echo "re-packer without a device: exit code $rc (expected non-zero), last lines:"; tail -3 "$TMP/repack.err"
[ $rc -ne 0 ] || { echo "the re-packer claims success without a GPU"; exit 1; }
rm -rf "$TMP"
echo "sanitize_host: no finding"
