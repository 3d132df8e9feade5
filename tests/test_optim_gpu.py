"""Fused AdamW (csrc/optim.hip through seunet_adamw_step) against the CPU oracle, the committed torch.optim.AdamW
fixture and PyTorch's own GPU AdamW."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _A():
    import seunet_amd as A
    return A


def test_fixture_parity(golden_dir):
    """Same inputs as the torch.optim.AdamW + MultiStepLR fixture: 5 steps, lr 1e-4 -> 1e-5 -> 1e-6."""
    A = _A()
    d = np.load(os.path.join(golden_dir, "adamw_known.npz"))
    n, steps = int(d["n"]), int(d["steps"])
    params = [torch.nn.Parameter(torch.from_numpy(d[f"init_{i}"]).cuda()) for i in range(n)]
    opt = A.AdamW(params, lr=0.0001)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2, 4], gamma=0.1)
    for t in range(steps):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(d[f"grad_{t}_{i}"]).cuda()
        assert opt.param_groups[0]["lr"] == pytest.approx(float(d["lrs"][t]))
        opt.step()
        sched.step()
    for i, p in enumerate(params):
        # f32 arithmetic with fused multiply-adds: a few ulp of the update, far below one ulp of most parameters
        np.testing.assert_allclose(p.detach().cpu().numpy(), d[f"final_{i}"], rtol=2e-7, atol=1e-9)
        for k in ("exp_avg", "exp_avg_sq"):
            ref = d[f"{k}_{i}"]
            np.testing.assert_allclose(opt.state[p][k].cpu().numpy(), ref, rtol=2e-6, atol=3e-7 * float(np.abs(ref).max()))


def test_against_oracle_on_the_network_registry():
    """All 117 tensors of the base network (ragged sizes 1 ... 110,592), 3 steps, against oracle/adamw_oracle.py."""
    import adamw_oracle as ao
    import seunet_oracle as orc
    A = _A()
    g = torch.Generator().manual_seed(7)
    shapes = [s for _, s in orc.parameter_registry(2, 1, 1)]
    init = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    grads = [[torch.randn(s, generator=g) * 1e-3 for s in shapes] for _ in range(3)]
    params = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    opt = A.AdamW(params, lr=1e-4)
    for t in range(3):
        for p, gr in zip(params, grads[t]):
            p.grad = gr.cuda()
        opt.step()
    want = ao.run([t.numpy() for t in init], [[gr.numpy() for gr in gs] for gs in grads], [1e-4] * 3)
    for i, p in enumerate(params):
        np.testing.assert_allclose(p.detach().cpu().numpy(), want["params"][i], rtol=2e-7, atol=1e-9)
        np.testing.assert_allclose(opt.state[p]["exp_avg_sq"].cpu().numpy(), want["exp_avg_sq"][i], rtol=2e-6,
                                   atol=3e-7 * float(np.abs(want["exp_avg_sq"][i]).max()))
    assert int(opt.state[params[0]]["step"].item()) == 3


def test_matches_torch_gpu_adamw_and_edge_cases():
    """Against torch.optim.AdamW on the same device; None gradients are skipped, weight_decay / maximize honoured,
    state_dict round-trips into a fresh optimizer."""
    A = _A()
    g = torch.Generator().manual_seed(3)
    shapes = [(5,), (1,), (257, 9), (4, 4, 3, 3, 3), (2049,)]
    a = [torch.nn.Parameter((torch.randn(s, generator=g)).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    kw = dict(lr=3e-3, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.1, maximize=True)
    oa, ob = A.AdamW(a, **kw), torch.optim.AdamW(b, foreach=False, **kw)
    for t in range(4):
        for i, (p, q) in enumerate(zip(a, b)):
            if i == 1 and t < 2:
                p.grad = q.grad = None            # this tensor joins at step 3 with its own step counter
                continue
            gr = torch.randn(p.shape, generator=g).cuda()
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=1e-7)
    assert int(oa.state[a[1]]["step"].item()) == 2 and int(oa.state[a[0]]["step"].item()) == 4
    fresh = A.AdamW(a, **kw)
    fresh.load_state_dict(oa.state_dict())
    assert torch.equal(fresh.state[a[2]]["exp_avg"], oa.state[a[2]]["exp_avg"])


def test_bad_arguments_report_errors():
    A = _A()
    lib = A._lib.load()
    import ctypes as C
    arr = (C.c_void_p * 1)()
    cnt = (C.c_longlong * 1)(4)
    assert lib.seunet_adamw_step(arr, arr, arr, arr, cnt, 1, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 0, None) != 0   # null tensors
    assert "null" in A._lib.last_error()
    assert lib.seunet_adamw_step(arr, arr, arr, arr, cnt, 0, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0, 0, None) != 0   # step counts from 1
