"""Host side of the bulk fast5 re-packer (csrc/fast5_repack.cpp, vbz_compression_amd/fast5.py) that needs no GPU: the sample counts
a work queue deals files by (reference: the per-file loop of python/fast5compress/fast5vbz.py:58-75, which users parallelise over
files: README.md:36-40), and that nothing pretends to work without a device."""
import os
import shutil

import pytest

from vbz_compression_amd import fast5

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _copies(tmp_path, k):
    out = []
    for i in range(k):
        f = str(tmp_path / ("reads%d.fast5" % i))
        shutil.copy(os.path.join(GOLDEN, "multi_fast5_zip.fast5"), f)
        out.append(f)
    return out


def test_sample_counts_and_the_deal_over_gpus(tmp_path):
    files = _copies(tmp_path, 3)
    try:
        counts = fast5.file_samples(files)
    except fast5.Hdf5NotFound:
        pytest.skip("no libhdf5 >= 1.10.3 on this box")
    assert counts == [(10, 1548931)] * 3          # the ten reads of the reference's test file (tests/golden/fast5_chunks.json)
    # contiguous shares balanced by samples, not by files
    assert fast5.deal_files([s for _, s in counts], 3) == [(0, 1), (1, 2), (2, 3)]
    assert fast5.deal_files([10, 10, 10, 1000, 10], 2) == [(0, 3), (3, 5)]
    shares = fast5.deal_files([5, 1, 1, 1, 5, 1, 1, 1], 4)
    assert shares[0][0] == 0 and shares[-1][1] == 8 and all(a[1] == b[0] for a, b in zip(shares, shares[1:]))


def test_no_device_no_result(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    files = _copies(tmp_path, 2)
    try:
        fast5.file_samples(files[:1])
    except fast5.Hdf5NotFound:
        pytest.skip("no libhdf5 >= 1.10.3 on this box")
    with pytest.raises(RuntimeError):
        fast5.compress_many(files, ".vbz", vbz_version=1)
