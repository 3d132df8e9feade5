"""Generate tests/golden/adamw_known.npz from the real torch.optim.AdamW + MultiStepLR (CPU, this container's PyTorch).

The reference's optimizer is third-party code (train.py:188-191: torch.optim.AdamW(lr=0.0001) + MultiStepLR); this script
runs that implementation itself on seeded tensors and stores inputs and expected outputs (data only).
Usage (from the repo root):  python oracle/make_golden_adamw.py
"""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "adamw_known.npz")

SHAPES = [(8, 2, 3, 3, 3), (8,), (2, 8, 1, 1, 1), (2,), (1, 8, 1, 1, 1), (1,), (33, 7), (1031,)]
STEPS = 5
MILESTONES = [2, 4]   # scheduler stepped once per optimizer step here, so the lr changes inside the fixture


def main():
    g = torch.Generator().manual_seed(20240501)
    params = [torch.nn.Parameter(torch.randn(s, generator=g) * 0.3) for s in SHAPES]
    init = [p.detach().clone().numpy() for p in params]
    opt = torch.optim.AdamW(params, lr=0.0001)                      # reference defaults (train.py:188)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=MILESTONES, gamma=0.1)
    grads, lrs = [], []
    for _ in range(STEPS):
        step_grads = [torch.randn(s, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g))) for s in SHAPES]
        for p, gr in zip(params, step_grads):
            p.grad = gr.clone()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        grads.append([gr.numpy() for gr in step_grads])
    data = {"steps": STEPS, "n": len(SHAPES), "lrs": np.array(lrs, dtype=np.float64), "torch_version": torch.__version__}
    for i in range(len(SHAPES)):
        data[f"init_{i}"] = init[i]
        data[f"final_{i}"] = params[i].detach().numpy()
        data[f"exp_avg_{i}"] = opt.state[params[i]]["exp_avg"].numpy()
        data[f"exp_avg_sq_{i}"] = opt.state[params[i]]["exp_avg_sq"].numpy()
        for t in range(STEPS):
            data[f"grad_{t}_{i}"] = grads[t][i]
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; lrs", lrs)


if __name__ == "__main__":
    main()
