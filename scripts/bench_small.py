"""Microbenchmark of the small full-resolution kernels at 4 x 128^3 (heads, pooling, up-sampling)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import seunet_amd
from seunet_amd import ops as S

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

n, d = 4, 128
which = sys.argv[1:] or ["head"]
if "head" in which:
    for nl in (4, 3):
        maps = [torch.randn(n, d >> l, d >> l, d >> l, device="cuda") for l in range(nl)]
        bias = torch.zeros(1, device="cuda")
        g = torch.randn(n, 1, d, d, d, device="cuda")
        print("head_fwd %d levels: %.1f us" % (nl, timeit(lambda: S.head_fwd(maps, bias))), flush=True)
        print("head_bwd %d levels: %.1f us" % (nl, timeit(lambda: S.head_bwd(g, nl))), flush=True)
