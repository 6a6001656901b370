"""Build the HIP shared libraries in-tree (so they travel with gpurun snapshots).

    python -m vbz_compression_amd.build            # libvbz_hip.so + libvbz_hdf_plugin.so + bin/vbz_fast5_repack

hipcc cross-compiles gfx950 code objects without a GPU present.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SOURCES = ["svb_kernels.hip", "zstd_encode.hip", "zstd_decode.hip", "zstd_decode_fast.hip", "zstd_decode_ref.hip", "helpers.hip", "vbz_api.hip"]
HEADERS = ["vbz_kernels.h", "zstd_entropy.h", "svb_wave.h", "zstd_runs.h", "../../include/vbz.h", "../../include/vbz_gpu.h", "../../include/vbz_hdf_plugin.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# The SDWA peephole is off: a byte-1 SDWA shift feeding v_bitop3_b16 produced a wrong block-header byte on hardware
# (see put_le in zstd_encode.hip); the kernels measure the same with and without the peephole.
FLAGS += ["-mllvm", "-amdgpu-sdwa-peephole=false"]
FLAGS += os.environ.get("VBZ_HIPCC_EXTRA", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _build_lib(name, objsub, extra, force, verbose):
    objdir = os.path.join(LIBDIR, objsub)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for name_, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % name_)
    lib = os.path.join(LIBDIR, name)
    if force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


def build_experiments(force=False, verbose=True):
    """lib/libvbz_hip_x.so: the same sources with -DVBZ_EXPERIMENTS -- the timed kernel instantiations (VBZ_HIP_PHASE_TIMING) and the
    measured-slower variants (VBZ_HIP_FUSE_SVB, VBZ_HIP_LONG_REPEATS=2|3, VBZ_HIP_ROUTING=2) for tools/ and for the tests that hold
    them to the product's bytes.  Not linked by the plugin or the re-packer; select it with VBZ_HIP_LIB."""
    os.makedirs(LIBDIR, exist_ok=True)
    return _build_lib("libvbz_hip_x.so", "obj_x", ["-DVBZ_EXPERIMENTS"], force, verbose)


def build(force=False, verbose=True, experiments=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    lib = _build_lib("libvbz_hip.so", "obj", [], force, verbose)
    if experiments:
        build_experiments(force, verbose)
    plugin_src = os.path.join(CSRC, "vbz_plugin.cpp")
    plugin = os.path.join(LIBDIR, "libvbz_hdf_plugin.so")
    if os.path.exists(plugin_src) and (force or _stale(plugin, [plugin_src, lib] + hdrs)):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-o", plugin, plugin_src,
               "-L" + LIBDIR, "-lvbz_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # bulk fast5 re-packer (host program: libhdf5 is loaded at run time, the codec is libvbz_hip.so)
    tool_src = os.path.join(CSRC, "fast5_repack.cpp")
    bindir = os.path.join(HERE, "bin")
    tool = os.path.join(bindir, "vbz_fast5_repack")
    if os.path.exists(tool_src) and (force or _stale(tool, [tool_src, lib, plugin] + hdrs)):
        os.makedirs(bindir, exist_ok=True)
        rocm = os.path.dirname(os.path.dirname(os.path.realpath(HIPCC)))
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-o", tool, tool_src,
               "-L" + LIBDIR, "-lvbz_hdf_plugin", "-lvbz_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-ldl", "-lz", "-pthread",
               "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + os.path.join(rocm, "lib")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, experiments="--no-experiments" not in sys.argv)
    print("ok")
