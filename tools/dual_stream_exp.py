#!/usr/bin/env python3
"""Experiment: does running two half-batch forwards on two HIP streams (so that one's HBM-bound GroupNorm passes overlap
the other's MFMA-bound convolutions) beat one full-batch forward?   python tools/dual_stream_exp.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nice-diffusion_amd')); sys.path.insert(0, ROOT)
import torch
import bench
from nicediffusion._engine import UNetPlan
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
margs, model, diff = bench.build(dev)
with torch.no_grad():
    full = model._plan(B)
    halves = [UNetPlan(model, B // 2) for _ in range(2)]
for p in [full] + halves:
    p.t_in.fill_(500)
    p.x_in.normal_()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print('one stream, B=%d: %.2f ms' % (B, timeit(full.run)))
def seq():
    halves[0].run(); halves[1].run()
print('one stream, 2 x B=%d back to back: %.2f ms' % (B // 2, timeit(seq)))
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def dual():
    for p, st in zip(halves, s):
        with torch.cuda.stream(st):
            p.run()
print('two streams, B=%d each: %.2f ms' % (B // 2, timeit(dual)))
# graph-captured dual (no host launch skew)
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    dual()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=cap):
    e = torch.cuda.Event(); e.record()
    for p, st in zip(halves, s):
        st.wait_event(e)
        with torch.cuda.stream(st):
            p.run()
        cap.wait_stream(st)
print('two streams inside one hipGraph: %.2f ms' % timeit(g.replay))
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=cap):
    full.run()
print('one stream inside a hipGraph, B=%d: %.2f ms' % (B, timeit(g1.replay)))
