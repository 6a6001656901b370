#!/opt/conda/bin/python3.9
"""Harvest golden fixtures (DATA only) from the reference's own test data.

Run in the build container only (needs /root/reference and the conda python that has h5py):
    /opt/conda/bin/python3.9 tools/make_golden.py

Writes under tests/golden/:
  test_data_read.i16      the 15 643-sample real read the reference's unit tests use
                          (values parsed out of vbz/test/test_data.h, stored as int16 LE)
  fast5_chunks.bin/.json  the raw HDF5 chunk payloads (filter 32020 output, sized format) of the
                          10 reads in test_data/multi_fast5_vbz.fast5 (v0) and the sha256 of the
                          v1 file's chunks, plus the sha256 of the int16 samples each must decode
                          to, taken from test_data/multi_fast5_zip.fast5
                          (reference python/test/test_vbz_filter.py:57-73 asserts exactly this).
  multi_fast5_zip.fast5   the reference's own test file (test_data/, gzip-compressed signal of 10 reads), byte for byte:
                          the input of the fast5 re-packer tests (python/test/test_vbz_filter.py:57-73 reads it too)
  fuzz_corpus.bin/.json   the 238 inputs of the reference's fuzz corpus (vbz/fuzzing/fuzz_corpus/, arbitrary bytes <= 2.3 KB
                          each) packed into one blob + index; `--fuzz-only` (any python) writes just these
Nothing here copies reference source text; these are inputs and expected outputs.
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def pack_fuzz_corpus():
    d = os.path.join(REF, "vbz/fuzzing/fuzz_corpus")
    blob = bytearray()
    index = []
    for name in sorted(os.listdir(d)):
        data = open(os.path.join(d, name), "rb").read()
        index.append(dict(name=name, offset=len(blob), size=len(data), sha256=hashlib.sha256(data).hexdigest()))
        blob += data
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, "fuzz_corpus.bin"), "wb").write(bytes(blob))
    json.dump(index, open(os.path.join(OUT, "fuzz_corpus.json"), "w"), indent=0)
    print("fuzz corpus", len(index), "files", len(blob), "bytes")


def main():
    pack_fuzz_corpus()
    if "--fuzz-only" in sys.argv:
        return 0
    import h5py

    os.makedirs(OUT, exist_ok=True)
    # --- real read used by vbz/test/vbz_test.cpp:248-288
    text = open(os.path.join(REF, "vbz/test/test_data.h")).read()
    body = text[text.index("{") + 1 : text.rindex("}")]
    vals = np.array([int(x) for x in re.findall(r"-?\d+", body)], dtype=np.int64)
    assert vals.min() >= -32768 and vals.max() <= 32767
    vals.astype("<i2").tofile(os.path.join(OUT, "test_data_read.i16"))
    print("test_data_read", len(vals))

    # --- shipped fast5 files
    fz = h5py.File(os.path.join(REF, "test_data/multi_fast5_zip.fast5"), "r")
    f0 = h5py.File(os.path.join(REF, "test_data/multi_fast5_vbz.fast5"), "r")
    f1 = h5py.File(os.path.join(REF, "test_data/multi_fast5_vbz_v1.fast5"), "r")
    index = []
    blob = bytearray()
    for key in sorted(fz.keys()):
        raw = fz[key]["Raw/Signal"][:]
        assert raw.dtype == np.int16
        d0 = f0[key]["Raw/Signal"]
        d1 = f1[key]["Raw/Signal"]
        assert d0.id.get_num_chunks() == 1 and d1.id.get_num_chunks() == 1
        _, c0 = d0.id.read_direct_chunk((0,))
        _, c1 = d1.id.read_direct_chunk((0,))
        plist0 = d0.id.get_create_plist()
        plist1 = d1.id.get_create_plist()
        filt0 = [plist0.get_filter(i) for i in range(plist0.get_nfilters())]
        filt1 = [plist1.get_filter(i) for i in range(plist1.get_nfilters())]
        index.append(
            dict(
                read=key,
                samples=int(len(raw)),
                raw_sha256=hashlib.sha256(raw.astype("<i2").tobytes()).hexdigest(),
                chunk_offset=len(blob),
                chunk_size=len(c0),
                chunk_sha256=hashlib.sha256(c0).hexdigest(),
                v1_chunk_sha256=hashlib.sha256(c1).hexdigest(),
                v1_identical=bool(c0 == c1),
                filter_v0=[int(filt0[0][0]), [int(x) for x in filt0[0][2]]],
                filter_v1=[int(filt1[0][0]), [int(x) for x in filt1[0][2]]],
            )
        )
        blob += c0
        print(key, len(raw), len(c0), c0 == c1)
    import shutil

    shutil.copyfile(os.path.join(REF, "test_data/multi_fast5_zip.fast5"), os.path.join(OUT, "multi_fast5_zip.fast5"))
    open(os.path.join(OUT, "fast5_chunks.bin"), "wb").write(bytes(blob))
    json.dump(index, open(os.path.join(OUT, "fast5_chunks.json"), "w"), indent=1)
    print("chunks bytes", len(blob))


if __name__ == "__main__":
    sys.exit(main())
