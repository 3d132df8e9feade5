"""Dump conv fwd (+stats) / dgrad results of one library build for a set of shapes; compare two dumps bitwise.
usage: diag_conv_dump.py dump <tag> | diag_conv_dump.py cmp <tagA> <tagB>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
CASES = [([64, 64], 64, 1, 8), ([128], 64, 1, 4), ([64], 128, 2, 8), ([128, 128], 128, 1, 4), ([32, 32], 32, 1, 16), ([64], 32, 1, 16), ([128], 128, 2, 2)]
if sys.argv[1] == "dump":
    import seunet_amd
    from seunet_amd import ops as S
    out = {}
    for ci, (split, cout, dil, size) in enumerate(CASES):
        torch.manual_seed(ci)
        cin = sum(split)
        srcs = [torch.randn((2, size, size, size, c), device="cuda") for c in split]
        w = torch.randn((cout, cin, 3, 3, 3), device="cuda") * 0.05
        b = torch.randn((cout,), device="cuda")
        (raw,), part, slots = S.conv3d(srcs, w, b, dil, 0, want_stats=True)
        mean, rstd = S.stats_finalize(part, slots, size ** 3)
        dy = torch.randn((2, size, size, size, cout), device="cuda")
        dsts = [torch.randn((2, size, size, size, c), device="cuda") for c in split]
        for t in dsts: t.copy_(torch.sin(torch.arange(t.numel(), device="cuda", dtype=torch.float32)).reshape(t.shape))
        S.conv3d([dy], w, None, dil, 0, transpose_flip=True, dsts=dsts, accumulate=[1] * len(split))
        out[ci] = dict(raw=raw.cpu(), mean=mean.cpu(), rstd=rstd.cpu(), part=part.sum(1).cpu(), dg=[t.cpu() for t in dsts])
    torch.save(out, os.path.join(ROOT, "gpurun_out", "dump_%s.pt" % sys.argv[2]))
else:
    a = torch.load(os.path.join(ROOT, "gpurun_out", "dump_%s.pt" % sys.argv[2])); b = torch.load(os.path.join(ROOT, "gpurun_out", "dump_%s.pt" % sys.argv[3]))
    for ci in a:
        A, B = a[ci], b[ci]
        f = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30))
        print(CASES[ci], "raw %.2e mean %.2e rstd %.2e part %.2e dgrad %s" % (f(A["raw"], B["raw"]), f(A["mean"], B["mean"]), f(A["rstd"], B["rstd"]), f(A["part"], B["part"]),
              " ".join("%.2e" % f(x, y) for x, y in zip(A["dg"], B["dg"]))))
