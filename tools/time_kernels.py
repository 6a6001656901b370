#!/usr/bin/env python3
"""Average duration of each kernel of the path on the bench workload (8192 synthetic reads resident in HBM), from the
library's own HIP events.  For A/B runs of kernel variants inside one gpurun call (box-to-box variance is up to 25 %):

    VBZ_HIPCC_EXTRA=-DSOMETHING python -m vbz_compression_amd.build --force && python tools/time_kernels.py [reads]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbz_compression_amd import batch
codec=batch.GpuCodec(0); torch.cuda.set_stream(codec.stream)
opts=codec.options(True,2,1,1); L=codec.L; n=int(sys.argv[1]) if len(sys.argv)>1 else 8192
lens=codec.synth_lengths(5,0,n); sizes=lens.to(torch.int64)*2
off,total=batch.layout(sizes.cpu(),64); raw=torch.empty(total,dtype=torch.uint8,device="cuda"); off=off.cuda()
codec.synth_signal(5,0,raw,off,lens); s32=sizes.to(torch.int32)
caps=torch.tensor([L.vbz_max_compressed_size(int(s),ctypes.byref(opts)) for s in sizes.cpu().tolist()],dtype=torch.int64)
coff,ctotal=batch.layout(caps,64); comp=torch.empty(ctotal,dtype=torch.uint8,device="cuda"); coff=coff.cuda(); cap32=caps.to(torch.int32).cuda()
cs=torch.zeros(n,dtype=torch.int32,device="cuda"); back=torch.empty_like(raw); res=torch.zeros(n,dtype=torch.int32,device="cuda")
for _ in range(2):
    codec.compress(raw,off,s32,comp,coff,cap32,cs,opts); codec.decompress(comp,coff,cs,back,off,s32,res,opts)
assert torch.equal(raw,back)
codec.profile_reset(); codec.profile(True)
for _ in range(10):
    codec.compress(raw,off,s32,comp,coff,cap32,cs,opts); codec.decompress(comp,coff,cs,back,off,s32,res,opts)
codec.profile(False); print({k: round(v[1]/max(v[0],1),3) for k,v in codec.profile_read().items() if "zstd" in k or "svb" in k})
