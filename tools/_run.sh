#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/ab_libs.py xlibs/r05_head.so xlibs/svbfull.so --rounds 10 2>&1 | tail -12
python tools/ab_libs.py xlibs/svbfull.so xlibs/r05_head.so --rounds 10 2>&1 | tail -12
