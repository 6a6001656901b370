"""Bulk (de)compression of the raw signal of fast5 files -- the interface of the reference's
python/fast5compress/fast5vbz.py (compress_fast5 at :17-55, the command line at :58-75).

The reference re-creates every read's Raw/Signal dataset through h5py, so libhdf5 calls the filter once per read.
Here the work is done by the native tool bin/vbz_fast5_repack (csrc/fast5_repack.cpp): all signals of a file are coded
in one batched call on the GPU and the finished chunks are stored with H5Dwrite_chunk.  Same arguments, same result
(a copy of the file named filename + output_suffix whose signal datasets are one vbz chunk each, or gzip level 1
with decompress=True); nothing here falls back to a CPU codec.

    python -m vbz_compression_amd.fast5 [-d] [-s SUFFIX] [--vbz-version N] [--gpus N] FILE...

Several files are a pipeline inside the tool (load | code | store on threads of their own) and, with --gpus N, one such process per
device over a share of the file list dealt by sample counts (compress_many).
"""
import argparse
import os
import subprocess

__version__ = "0.2.0"

HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(HERE, "bin", "vbz_fast5_repack")


class Hdf5NotFound(RuntimeError):
    """No usable libhdf5 (>= 1.10.3) on this machine: the tool's exit code 3."""


def _fail(what, filename, code):
    if code == 3:
        raise Hdf5NotFound("vbz_fast5_repack found no usable libhdf5 (pass hdf5_lib= or set VBZ_HDF5_LIB)")
    raise RuntimeError("%s failed on %s (exit code %d)" % (what, filename, code))


def _tool():
    if not os.path.exists(TOOL):
        raise RuntimeError("%s is missing: run python -m vbz_compression_amd.build" % TOOL)
    return TOOL


def compress_fast5(filename, output_suffix, vbz_version=0, decompress=False, hdf5_lib=None):
    """(De)compress the raw signal in the fast5; returns the name of the rewritten copy (fast5vbz.py:17-55)."""
    cmd = [_tool(), "-s", output_suffix, "--vbz-version", str(int(vbz_version))]
    if decompress:
        cmd.append("-d")
    if hdf5_lib:
        cmd += ["--hdf5-lib", hdf5_lib]
    cmd.append(filename)
    out = subprocess.run(cmd, stdout=subprocess.PIPE, universal_newlines=True)
    if out.returncode != 0:
        _fail("vbz_fast5_repack", filename, out.returncode)
    return out.stdout.strip().splitlines()[-1]


def file_samples(filenames, hdf5_lib=None):
    """[(reads, samples)] of every file: the read_*/Raw/Signal datasets' extents, nothing read or decoded (no GPU needed)."""
    cmd = [_tool(), "--samples"] + (["--hdf5-lib", hdf5_lib] if hdf5_lib else []) + list(filenames)
    out = subprocess.run(cmd, stdout=subprocess.PIPE, universal_newlines=True)
    if out.returncode != 0:
        _fail("vbz_fast5_repack --samples", filenames[0], out.returncode)
    rows = [ln.rsplit("\t", 2) for ln in out.stdout.splitlines()]
    assert [r[0] for r in rows] == list(filenames)
    return [(int(r[1]), int(r[2])) for r in rows]


def deal_files(samples, gpus):
    """Contiguous shares of the file list for `gpus` devices, balanced by cumulative SAMPLES (the work queue of shard.py:
    files hold reads of very different lengths, so neither the file count nor the read count balances).  Returns a list of
    (first, last_exclusive) per device."""
    from . import shard

    return shard.partition_reads(samples, gpus)


def compress_many(filenames, output_suffix, vbz_version=0, decompress=False, gpus=1, hdf5_lib=None):
    """compress_fast5 for a list of files, the way the reference's users run it (README.md:36-40: many files side by side): one
    process of the native tool per GPU, each over its share of the list, each a pipeline -- the next file is read and inflated and
    the previous one stored while the GPU codes the current one.  Returns the names of the rewritten copies, in input order."""
    filenames = list(filenames)
    if not filenames:
        return []
    gpus = max(1, min(int(gpus), len(filenames)))
    if gpus > 1:   # (the devices the kernel driver knows; a parent that never loads the HIP runtime cannot hand children an initialised one)
        import glob

        have = 0
        for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            try:
                with open(path) as f:
                    have += any(ln.split()[0] == "simd_count" and int(ln.split()[1]) > 0 for ln in f if ln.strip())
            except (OSError, ValueError, IndexError):
                pass
        if have and gpus > have:
            raise ValueError("--gpus %d but %d GPU(s) in /sys/class/kfd" % (gpus, have))
    shares = [(0, len(filenames))] if gpus == 1 else deal_files([s for _, s in file_samples(filenames, hdf5_lib)], gpus)
    base = [_tool(), "-s", output_suffix, "--vbz-version", str(int(vbz_version))] + (["-d"] if decompress else []) + (["--hdf5-lib", hdf5_lib] if hdf5_lib else [])
    procs = []
    for device, (a, b) in enumerate(shares):
        if b > a:
            # (one process: the tool's own default -- $VBZ_HIP_DEVICE, else device 0 -- stands)
            dev = ["--device", str(device)] if gpus > 1 else []
            procs.append((a, b, subprocess.Popen(base + dev + filenames[a:b], stdout=subprocess.PIPE, universal_newlines=True)))
    names = [None] * len(filenames)
    failed = None
    for a, b, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = failed or (filenames[a], p.returncode)
            continue
        got = out.strip().splitlines()
        assert len(got) == b - a, (got, filenames[a:b])
        names[a:b] = got
    if failed:
        _fail("vbz_fast5_repack", *failed)
    return names


def list_fast5(filename, export_signal=None, export_chunks=None, hdf5_lib=None):
    """The read_*/Raw/Signal datasets of a file: dicts with name, samples, integer size, filter ids, stored bytes
    and the FNV-1a-64 of the samples (vbz datasets are decoded on the GPU)."""
    cmd = [_tool(), "--list", filename]
    if export_signal:
        cmd += ["--export-signal", export_signal]
    if export_chunks:
        cmd += ["--export-chunks", export_chunks]
    if hdf5_lib:
        cmd += ["--hdf5-lib", hdf5_lib]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, universal_newlines=True)
    if out.returncode != 0:
        _fail("vbz_fast5_repack --list", filename, out.returncode)
    reads = []
    for line in out.stdout.splitlines():
        name, samples, size, filters, stored, fnv, chunk = line.split("\t")
        reads.append({"name": name, "samples": int(samples), "integer_size": int(size),
                      "filters": [] if filters == "-" else [int(x) for x in filters.split(",")],
                      "stored_bytes": int(stored), "fnv1a64": fnv, "chunk_bytes": int(chunk)})
    return reads


def main(args=None):
    parser = argparse.ArgumentParser("fast5compress")
    parser.add_argument("files", nargs="*", help="input files")
    parser.add_argument("-d", "--decompress", action="store_true", default=False)
    parser.add_argument("-s", "--output-suffix", default=".tmp")
    parser.add_argument("-v", "--version", action="version", version=__version__)
    parser.add_argument("--vbz-version", type=int, default=1)
    parser.add_argument("--hdf5-lib", default=None, help="libhdf5 shared object to load (default: search)")
    parser.add_argument("--gpus", type=int, default=1, help="deal the files over this many GPUs (one process each), by their sample counts")
    a = parser.parse_args(args)
    for name in compress_many(a.files, a.output_suffix, a.vbz_version, a.decompress, a.gpus, a.hdf5_lib):
        print(name)


if __name__ == "__main__":
    main()
