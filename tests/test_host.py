"""Host-side logic that needs no GPU: module surface, DropLayer, window tables, failure modes, and the
N>1 data-parallel exchange steps under gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import seunet_oracle as orc


def test_module_surface_matches_reference_contract():
    import seunet_amd as A
    cfg, net = A.get_model()
    assert cfg == {} and net.in_channel == 2 and net.n_classes == 1
    m = A.SE_UNet()                                  # class default in_channel=1 (SE_UNet.py:100)
    assert m.in_channel == 1
    m = A.SE_UNet(in_channel=2, n_classes=1)
    sd = m.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == orc.parameter_registry(2, 1, 1)
    assert len(list(m.buffers())) == 0
    m.load_state_dict(orc.deterministic_state_dict(2, 1, 1, 0), strict=False)     # train.py:196
    torch.optim.AdamW(m.parameters(), lr=1e-4)
    assert isinstance(torch.nn.DataParallel(m).module, A.SE_UNet)                  # train.py:577, 626
    assert m.training and not m.eval().training


def test_default_init_is_pytorch_default():
    import seunet_amd as A
    torch.manual_seed(0)
    m = A.SE_UNet(2, 1)
    w = m.ec1.conv1.weight
    assert float(w.abs().max()) <= 1 / np.sqrt(54) + 1e-6 and float(w.abs().max()) > 0.9 / np.sqrt(54)


def test_droplayer_matches_oracle_and_reference_formula():
    import seunet_amd as A
    d = A.DropLayer(24, 0.3)
    torch.manual_seed(7)
    s = d.scale(2)
    torch.manual_seed(7)
    ref = orc.drop_scale_from_uniform(torch.rand(2, 24, 1, 1, 1), 24)
    assert torch.equal(s, ref)
    keep = (s > 0).sum()
    assert torch.allclose(s[s > 0], torch.tensor(24.0) / (keep + 0.01))
    x = torch.ones(2, 24, 2, 2, 2)
    assert torch.equal(d.eval()(x), x)


def test_window_starts_and_two_channel(golden_dir):
    import seunet_amd as A
    g = np.load(os.path.join(golden_dir, "window_starts.npz"))
    for k in g.files:
        assert A.window_starts(int(k)) == list(g[k])
    with pytest.raises(ValueError):
        A.window_starts(64)
    hu = np.linspace(-1500, 1500, 31)
    for a, b in zip(A.two_channel(hu), orc.two_channel(hu)):
        np.testing.assert_allclose(a, b)


def test_cpu_input_fails_loudly():
    import seunet_amd as A
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.SE_UNet(2, 1)(torch.zeros(1, 2, 8, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.dice_loss(torch.rand(4), torch.rand(4))
    with pytest.raises(RuntimeError):
        A.sliding_window_predict(A.SE_UNet(2, 1), torch.zeros(1, 2, 128, 128, 128))


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from seunet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _lib.load()


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import seunet_amd as A
    from seunet_amd import ddp
    from seunet_amd.losses import _value
    ddp.init_from_env("gloo")
    torch.manual_seed(0)
    m = A.SE_UNet(2, 1)
    ddp.broadcast_parameters(m)
    # gradients laid out by the very helper SE_UNet's backward uses: views of one flat buffer, the dead dc62 block
    # (mid-list in registry order) takes no space, so the live gradients are back to back
    from seunet_amd.SE_UNet import alloc_flat_grads
    plist = list(m.parameters())
    names = [n for n, _ in m.named_parameters()]
    # layout of the backward pass: encoder blocks + heads first, decoder blocks (whose gradients are final first) last
    flat, grads, split = alloc_flat_grads(plist, m._dead, torch.device("cpu"), names)
    dec = sum(p.numel() for n, p in m.named_parameters() if n.startswith("dc") and not n.startswith("dc0_") and not n.startswith("dc62."))
    ok_layout = split == flat.numel() - dec and m.dc1.conv1.weight.numel() > 0
    pattern = (torch.arange(flat.numel()) % 4096).float()       # (small integers: every partial sum of any reduction order is exact)
    flat.copy_(pattern * (rank + 1))
    for p, g in zip(plist, grads):
        p.grad = g
    ok_layout = ok_layout and dict(zip(names, grads))["dc1.conv1.weight"].data_ptr() == flat[split:].data_ptr() \
        and dict(zip(names, grads))["ec1.conv1.weight"].data_ptr() == flat.data_ptr()
    live = [p for n, p in m.named_parameters() if not n.startswith("dc62.")]
    zero_copy = ddp._flat_view([p.grad for p in m.parameters() if p.grad is not None]) is not None
    n = ddp.allreduce_gradients(m.parameters())
    ok_flat = ok_layout and zero_copy and n == 1_520_314 - m.dc62.conv1.weight.numel() and torch.equal(flat, pattern * (world * (world + 1) // 2))
    # fallback path: separately allocated gradients
    for p in live:
        p.grad = torch.full_like(p, float(rank + 1))
    ddp.allreduce_gradients(m.parameters(), average=True)
    ok_fallback = all(torch.allclose(p.grad, torch.full_like(p, (world + 1) / 2)) for p in live) and m.dc62.conv1.weight.grad is None
    # loss semantics: ratio of ALL-REDUCED sums == loss of the concatenated global batch
    g = torch.Generator().manual_seed(3)
    p_all, t_all = torch.rand(world, 1, 8, 8, 8, generator=g), (torch.rand(world, 1, 8, 8, 8, generator=g) > 0.8).float()
    p, t = p_all[rank:rank + 1], t_all[rank:rank + 1]
    sums = torch.zeros(7, dtype=torch.float64)
    sums[0], sums[1], sums[2] = (p * t).sum(), p.sum(), t.sum()
    dist.all_reduce(sums)
    ok_loss = abs(float(_value(sums, 1.0, 0.0, 0.0)) - float(orc.dice_loss(p_all, t_all))) < 1e-6
    q.put((rank, bool(ok_flat), bool(ok_fallback), bool(ok_loss)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_data_parallel_exchange_gloo(world):
    """The flat-bucket exchange, its fallback and the global-batch loss ratio with 2 ranks and with the 8 ranks of BASELINE
    configs[2] / [4] (CPU tensors over gloo; the layout helper is the one SE_UNet's backward uses)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 31 * world) % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True, True, True) for r in range(world)], res


def test_adamw_surface_and_no_cpu_path():
    """optim.AdamW mirrors torch.optim.AdamW's constructor / param_groups and refuses CPU tensors."""
    import seunet_amd as A
    p = torch.nn.Parameter(torch.zeros(4))
    opt = A.AdamW([p], lr=1e-4)
    ref = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(4))], lr=1e-4)
    for k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"):
        assert opt.param_groups[0][k] == ref.param_groups[0][k], k
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)   # train.py:189-191 works on it
    sched.step()
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-5)
    with pytest.raises(ValueError):
        A.AdamW([p], amsgrad=True)
    with pytest.raises(ValueError):
        A.AdamW([p], lr=-1.0)
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):     # no CPU fallback: either the library is missing or the tensor is not on the GPU
        opt.step()


def test_postprocess_surface_and_no_cpu_path():
    import numpy as np
    import seunet_amd as A
    v = np.zeros((20, 20, 4)); v[:] = 1.0
    out = A.zero_borders(v.copy())                      # prediction.py:111-114 literals
    assert out[:3].sum() == 0 and out[17:].sum() == 0 and out[:, :3].sum() == 0 and out[:, 17:].sum() == 0
    assert out[3:17, 3:17].min() == 1.0
    with pytest.raises(RuntimeError):                   # CPU tensor: refused, there is no CPU implementation
        A.double_threshold_iteration(torch.zeros(2, 2, 2), 0.5, 0.4)


def test_reference_checkpoints_load_unchanged(tmp_path):
    """train.py:322-324 saves ``model.module.state_dict()``; train.py:194-196 loads with strict=False.  A file written that
    way from a reference-shaped model (the oracle has the reference's registry) loads into the HIP model, with or without a
    DataParallel ``module.`` prefix, and a state_dict saved from the HIP model loads back into the reference-shaped one."""
    import seunet_amd as A
    o = orc.build_oracle(2, 1, 1, seed=3)
    path = tmp_path / "SE_UNet_9.pth"
    torch.save(o.state_dict(), path)
    m = A.SE_UNet(in_channel=2, n_classes=1)
    rec = A.load_reference_checkpoint(m, str(path))
    assert not rec.missing_keys and not rec.unexpected_keys
    for (k, v), (k2, v2) in zip(m.state_dict().items(), o.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    m2 = A.SE_UNet(in_channel=2, n_classes=1)
    rec = A.load_reference_checkpoint(m2, {"module." + k: v for k, v in o.state_dict().items()})
    assert not rec.missing_keys and not rec.unexpected_keys and torch.equal(m2.dc5.conv1.weight, o.dc5.conv1.weight)
    o2 = orc.build_oracle(2, 1, 1, seed=4)
    o2.load_state_dict(m.state_dict(), strict=True)
    assert torch.equal(o2.ec33.conv1.weight, o.ec33.conv1.weight)
