#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/compare_libs.py xlibs/r04_base.so vbz_compression_amd/lib/libvbz_hip.so --env VBZ_HIP_SHARED_TABLES=0 2>&1 | tail -30
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
