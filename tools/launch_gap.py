#!/usr/bin/env python
"""Cost of one dependent launch in a stream: N back-to-back launches of a trivial kernel (ssp_bn_eval_prepare on 64
channels), wall time / N, with the host far ahead of the GPU (N is large) - eager and as a captured graph."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from singleshotpose_amd import _lib
v = torch.ones(8, 64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def chain(n):
    for _ in range(n):
        _lib.call('ssp_bn_eval_prepare', 64, v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), 1e-4,
                  v[4].data_ptr(), v[5].data_ptr(), v[6].data_ptr(), v[7].data_ptr(), st)
chain(10); torch.cuda.synchronize()
for n in (200, 2000):
    t0 = time.perf_counter(); chain(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('eager  n=%4d: host issue %.2f us/launch, total %.2f us/launch' % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st = s.cuda_stream
    chain(3); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        chain(200)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('graph  n= 200: total %.2f us/launch' % ((t2 - t0) / 200 * 1e6))
