"""The CPU oracle (oracle/seunet_oracle.py) against the fixtures generated from the
real reference by oracle/make_golden.py.  Runs everywhere (no GPU, no /root/reference)."""
import os

import numpy as np
import pytest
import torch

import seunet_oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_registry_matches_survey_counts():
    reg = orc.parameter_registry(2, 1, 1)
    assert len(reg) == 117
    assert sum(int(np.prod(s)) for _, s in reg) == 1_520_314
    m = orc.OracleSEUNet(2, 1)
    assert [k for k, _ in reg] == list(m.state_dict().keys())
    assert all(tuple(v.shape) == s for (k, s), v in zip(reg, m.state_dict().values()))
    assert len(list(m.buffers())) == 0


@pytest.mark.parametrize("tag,inch", [("fwd32_in2", 2), ("fwd32_in1", 1)])
def test_eval_forward_32(golden_dir, tag, inch):
    g = _load(golden_dir, tag + ".npz")
    m = orc.build_oracle(inch, 1, 1, seed=0)
    x = orc.synthetic_batch(2, (32, 32, 32), inch, seed=1)["image"]
    with torch.no_grad():
        p0, p1 = m(x)
    np.testing.assert_allclose(p0.numpy(), g["pred0"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(p1.numpy(), g["pred1"], atol=1e-6, rtol=0)


def test_eval_forward_64_config1(golden_dir):
    """BASELINE.json configs[0]: fp32 1x64^3 CPU forward."""
    g = _load(golden_dir, "fwd64_in2.npz")
    m = orc.build_oracle(2, 1, 1, seed=0)
    x = orc.synthetic_batch(1, (64, 64, 64), 2, seed=2)["image"]
    with torch.no_grad():
        p0, p1 = m(x)
    np.testing.assert_allclose(p0[0, 0, ::4, ::4, ::4].numpy(), g["pred0_s"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(p1[0, 0, ::4, ::4, ::4].numpy(), g["pred1_s"], atol=1e-6, rtol=0)
    assert abs(float(p0.double().sum()) - float(g["pred0_sum"])) < 1e-2
    assert abs(float(p1.double().abs().sum()) - float(g["pred1_abs"])) < 1e-2


@pytest.mark.parametrize("stage", [1, 3])
def test_backward_32(golden_dir, stage):
    g = _load(golden_dir, f"bwd32_stage{stage}.npz")
    m = orc.build_oracle(2, 1, 1, seed=0)
    b = orc.synthetic_batch(2, (32, 32, 32), 2, seed=3)
    pe, pd = m(b["image"])
    loss = orc.stage_loss(stage, pe, pd, b["label"], b["weight"], b["skel"])
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    for name, p in m.named_parameters():
        if name + "|none" in g.files:
            assert p.grad is None, name                     # dead dc62
            continue
        gn = float(g[name + "|norm"])
        assert abs(float(p.grad.double().norm()) - gn) <= 1e-5 * max(gn, 1e-6) + 1e-9, name
        np.testing.assert_allclose(p.grad.reshape(-1)[:8].numpy(), g[name + "|head"],
                                   rtol=1e-4, atol=1e-8, err_msg=name)


def test_train_mode_droplayer(golden_dir):
    """DropLayer RNG order and batch-coupled scale."""
    g = _load(golden_dir, "fwd32_train.npz")
    m = orc.build_oracle(2, 1, 1, seed=0, train=True)
    x = orc.synthetic_batch(2, (32, 32, 32), 2, seed=4)["image"]
    torch.manual_seed(123)
    with torch.no_grad():
        p0, p1 = m(x)
    np.testing.assert_allclose(p0.numpy(), g["pred0"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(p1.numpy(), g["pred1"], atol=1e-6, rtol=0)


def test_losses_known_answers(golden_dir):
    g = _load(golden_dir, "loss_known.npz")
    # values quoted in SURVEY.md section 8(c)
    assert abs(float(g["dice_loss"]) - 0.81824607) < 1e-6
    assert abs(float(g["general_union_loss_lib"]) - 0.65585136) < 1e-6
    assert abs(float(g["atr_loss"]) - 0.67657852) < 1e-6
    gen = torch.Generator().manual_seed(1234)
    p = torch.rand(2, 1, 16, 16, 16, generator=gen)
    t = (torch.rand(2, 1, 16, 16, 16, generator=gen) > 0.9).float()
    w = 1 + torch.rand(2, 1, 16, 16, 16, generator=gen)
    s = t * (torch.rand(2, 1, 16, 16, 16, generator=gen) > 0.5).float()
    for name, fn, args in (("dice_loss", orc.dice_loss, (t,)),
                           ("general_union_loss_lib", orc.general_union_loss_lib, (t, w)),
                           ("atr_loss", orc.atr_loss, (t, s, w))):
        pp = p.clone().requires_grad_(True)
        l = fn(pp, *args)
        l.backward()
        assert abs(float(l) - float(g[name])) < 1e-6
        np.testing.assert_allclose(pp.grad.numpy(), g[name + "|grad"], rtol=1e-5, atol=1e-10)


def test_window_starts(golden_dir):
    g = _load(golden_dir, "window_starts.npz")
    for k in g.files:
        assert orc.window_starts(int(k)) == list(g[k]), k
    assert len(orc.window_starts(512)) == 7                   # 343 windows for 512^3
    with pytest.raises(ValueError):
        orc.window_starts(100)


def test_sliding_window_equals_bruteforce():
    """Assembled volume == per-voxel mean over the window outputs covering it."""
    class Tiny(torch.nn.Module):
        def forward(self, x):
            y = x[:, :1] * 3.0 - x[:, 1:2] + x[:, :1].mean()
            return y, y
    x = torch.rand(1, 2, 24, 20, 16)
    out = orc.sliding_window_predict(Tiny(), x, cube=16, step=8)
    acc = np.zeros((24, 20, 16)); cnt = np.zeros((24, 20, 16))
    for a in orc.window_starts(24, 16, 8):
        for b in orc.window_starts(20, 16, 8):
            for c in orc.window_starts(16, 16, 8):
                w = x[:, :, a:a + 16, b:b + 16, c:c + 16]
                acc[a:a + 16, b:b + 16, c:c + 16] += torch.sigmoid(Tiny()(w)[1])[0, 0].numpy()
                cnt[a:a + 16, b:b + 16, c:c + 16] += 1
    np.testing.assert_allclose(out, acc / cnt, atol=1e-12)


def test_two_channel():
    hu = np.array([-2000.0, -1024, -1000, 0, 500, 1024, 3000])
    c0, c1 = orc.two_channel(hu)
    np.testing.assert_allclose(c0, [0, 0, 24 / 2048, 0.5, 1524 / 2048, 1, 1])
    np.testing.assert_allclose(c1, [0, 0, 0, 1000 / 1500, 1, 1, 1])


def test_adamw_oracle_matches_torch_adamw_fixture(golden_dir):
    """oracle/adamw_oracle.py against torch.optim.AdamW + MultiStepLR outputs (oracle/make_golden_adamw.py)."""
    import adamw_oracle as ao
    d = _load(golden_dir, "adamw_known.npz")
    n, steps = int(d["n"]), int(d["steps"])
    lrs = list(d["lrs"])
    assert lrs[0] == 1e-4 and lrs[2] == pytest.approx(1e-5) and lrs[4] == pytest.approx(1e-6)   # MultiStepLR inside the fixture
    assert [ao.multistep_lr(1e-4, e, (2, 4)) for e in range(5)] == pytest.approx(lrs)
    res = ao.run([d[f"init_{i}"] for i in range(n)], [[d[f"grad_{t}_{i}"] for i in range(n)] for t in range(steps)], lrs)
    for i in range(n):
        np.testing.assert_array_equal(res["params"][i], d[f"final_{i}"])                      # bit-exact parameters
        # moments: the vectorised torch kernels contract multiply-adds differently -> a rounding of the largest element
        for k in ("exp_avg", "exp_avg_sq"):
            ref = d[f"{k}_{i}"]
            np.testing.assert_allclose(res[k][i], ref, rtol=1e-6, atol=2e-7 * float(np.abs(ref).max()))


def test_dti_oracle_matches_reference_function_fixture(golden_dir):
    """oracle/dti_oracle.c against outputs of the reference's own double_threshold_iteration (oracle/make_golden_dti.py)."""
    import dti_oracle as do
    d = _load(golden_dir, "dti_known.npz")
    assert int(d["n"]) >= 6
    for c in range(int(d["n"])):
        got = do.double_threshold_iteration(d[f"pred_{c}"], float(d[f"h_{c}"]), float(d[f"l_{c}"]))
        assert got.dtype == np.float64 and set(np.unique(got)) <= {0.0, 1.0}
        np.testing.assert_array_equal(got.astype(np.uint8), d[f"out_{c}"])
    # the sweep is order dependent: a weak chain fed from its far end along k stays off except next to the seed
    chain = np.zeros((3, 3, 10)); chain[1, 1, :] = 0.45; chain[1, 1, 9] = 0.9
    out = do.double_threshold_iteration(chain, 0.5, 0.4)
    assert out[1, 1].tolist() == [0, 0, 0, 0, 0, 0, 0, 0, 1, 1]
    chain[1, 1, 9] = 0.45; chain[1, 1, 0] = 0.9          # fed from the near end: the whole chain switches on
    assert do.double_threshold_iteration(chain, 0.5, 0.4)[1, 1].sum() == 10


def test_dti_oracle_float32_copies_match_reference_fixture(golden_dir):
    """The float32 copies of the function (train.py:25-49, test.py:18-42: pred*255 rounded to float32) differ from
    prediction.py's float64 copy on voxels within a float32 ulp of a threshold; fixture from the reference's own copies."""
    import dti_oracle as do
    d = _load(golden_dir, "dti_known_f32.npz")
    differ = 0
    for c in range(int(d["n"])):
        v, h, l = d[f"pred_{c}"], float(d[f"h_{c}"]), float(d[f"l_{c}"])
        got32 = do.double_threshold_iteration(v, h, l, "float32").astype(np.uint8)
        got64 = do.double_threshold_iteration(v, h, l, "float64").astype(np.uint8)
        np.testing.assert_array_equal(got32, d[f"out_{c}"])
        np.testing.assert_array_equal(got64, d[f"out64_{c}"])
        differ += int((got32 != got64).sum())
    assert differ > 100


def test_pipeline_oracle_matches_reference_helpers_fixture(golden_dir):
    """oracle/pipeline_oracle.py against outputs of the reference's own data.py helpers (oracle/make_golden_pipeline.py):
    the flip / rotate index maps (all 24 combinations), a seeded CropSegData batch including the order of the random
    draws, the float16 weight exponentiation, and the float64 normalisation of int16 crops -- all bit-exact."""
    import random
    import pipeline_oracle as po
    g = _load(golden_dir, "pipeline_known.npz")
    n = g["aug_out"].shape[1]
    code = np.arange(n ** 3, dtype=np.int32).reshape(n, n, n)
    assert len(g["aug_params"]) == 24
    for (f0, f1, f2, rot), out in zip(g["aug_params"], g["aug_out"]):
        c = po.aug_code(None if (f0, f1, f2) == (1, 1, 1) else (f0, f1, f2), {0: None, 1: "left", 2: "right"}[int(rot)])
        np.testing.assert_array_equal(po.apply_code(code, c), out)
        x = {0: lambda a: a, 1: po.rotate_left, 2: po.rotate_right}[int(rot)](po.flip(code, (f0, f1, f2)))
        np.testing.assert_array_equal(x, out)
    img, label, w16, cube, b = g["img"], g["label"], g["weight16"], int(g["cube"]), int(g["batch"])
    assert w16.dtype == np.float16 and img.dtype == np.int16
    for seed in (1, 2):
        random.seed(100 + seed)
        np.random.seed(200 + seed)
        plan = po.draw_stage1_plan(img.shape, b, cube)
        out = po.crop_batch(img.astype(np.float32), plan["starts"], plan["codes"], cube, label, w16, None, plan["u"], f64_math=False)
        for k in ("data", "label", "weight"):
            np.testing.assert_array_equal(out[k], g[f"s1_{seed}_{k}"])
    out = po.crop_batch(img, [tuple(s) for s in g["s2_starts"]], [0, 0], cube)
    np.testing.assert_array_equal(out["data"], g["s2_data"])
    # the host-side plan of the product package draws the same numbers in the same order
    import seunet_amd as A
    for seed in (1, 2):
        random.seed(100 + seed); np.random.seed(200 + seed)
        a = po.draw_stage1_plan(img.shape, b, cube)
        random.seed(100 + seed); np.random.seed(200 + seed)
        assert A.draw_stage1_plan(img.shape, b, cube) == a
    for f in ((1, 1, -1), (-1, 1, 1), (-1, -1, -1), None):
        for r in (None, "left", "right"):
            assert A.aug_code(f, r) == po.aug_code(f, r)


def _hm_locs(g):
    """The candidate lists of the hard-mining samplers as AirwayHMData.crop / AirwayHMData3.crop build them (data.py:305-306,
    454-457): small-airway voxels from the label's Euclidean distance transform, skeleton voxels the previous stage missed."""
    from scipy import ndimage
    label, skeleton, pred = g["label"], g["skeleton"], g["pred"]
    dis = ndimage.distance_transform_edt(label)
    return np.where(skeleton * (1 - pred)), np.where((dis * skeleton) < 2), tuple(g["br_skel"])


def test_stage23_sampler_plans_match_reference_samplers_fixture(golden_dir):
    """The stage-2 / stage-3 draw order (AirwayHMData / AirwayHMData3 ``__getitem__``: data.py:359-408, 546-584 with ``crop``
    :301-324, :449-491 and the samplers :85-252) restated in oracle/pipeline_oracle.py against batches produced by the
    reference's OWN crop / process_img / augment methods and sampler functions (oracle/make_golden_pipeline_hm.py), bit-exact;
    and the product package's host-side plans draw the same numbers in the same order."""
    import random
    import pipeline_oracle as po
    import seunet_amd as A
    g = _load(golden_dir, "pipeline_hm_known.npz")
    img, label, skeleton, cube, b = g["img"], g["label"], g["skeleton"], int(g["cube"]), int(g["batch"])
    loc_skel, loc_small, loc_break = _hm_locs(g)
    assert len(loc_skel[0]) > 0 and len(loc_small[0]) > 0 and len(loc_break[0]) > 0
    kinds = set()
    for seed in (1, 2):
        for stage in (2, 3):
            def plan_of(mod):
                random.seed(300 + seed); np.random.seed(400 + seed)
                if stage == 2:
                    return mod.draw_stage2_plan(img.shape, b, loc_skel, loc_small, cube)
                return mod.draw_stage3_plan(img.shape, b, loc_skel, loc_small, loc_break, cube)
            plan = plan_of(po)
            w = g["weight16"] if stage == 2 else g["weight3"]
            out = po.crop_batch(img, plan["starts"], plan["codes"], cube, label, w, skeleton if stage == 3 else None, plan["u"])
            for k in ("data", "label", "weight") + (("skel",) if stage == 3 else ()):
                np.testing.assert_array_equal(out[k], g[f"s{stage}_{seed}_{k}"], err_msg=f"stage {stage} seed {seed} {k}")
            mine = plan_of(A)
            assert mine["u"] == plan["u"] and mine["starts"] == plan["starts"] and mine["codes"] == plan["codes"]
            kinds |= set(mine["kinds"])
    assert {"random", "small", "skeleton", "break"} <= kinds          # every branch of both samplers was taken
    # empty candidate lists: stage 2 falls back to the small-airway list, then to a uniform origin (hard_sample, data.py:124-136)
    empty = (np.array([], dtype=np.int64),) * 3
    random.seed(1); np.random.seed(2)
    a = po.draw_stage2_plan(img.shape, 8, empty, empty, cube, hard_ratio=1.0)
    random.seed(1); np.random.seed(2)
    m = A.draw_stage2_plan(img.shape, 8, empty, empty, cube, hard_ratio=1.0)
    assert m["starts"] == a["starts"] and set(m["kinds"]) == {"random"}


def test_metrics_oracle_matches_reference_metrics_module_fixture(golden_dir):
    """oracle/components_oracle.py's metric functions against values produced by the reference's own metrics.py
    (oracle/make_golden_components.py): identical rounded percentages and branch counts."""
    import components_oracle as co
    g = _load(golden_dir, "metrics_known.npz")
    for c in range(int(g["n"])):
        pred, label, skel, parsing = g[f"pred_{c}"], g[f"label_{c}"], g[f"skel_{c}"], g[f"parsing_{c}"].astype(np.int32)
        tot, det, bd = co.branch_detected_calculation(pred, parsing, skel)
        vals = [bd, co.dice_coefficient_score_calculation(pred, label), co.tree_length_calculation(pred, skel),
                co.false_positive_rate_calculation(pred, label), co.false_negative_rate_calculation(pred, label),
                co.sensitivity_calculation(pred, label), co.specificity_calculation(pred, label), co.precision_calculation(pred, label)]
        assert [tot, det] == list(g[f"branches_{c}"])
        assert vals == list(g[f"values_{c}"]), (c, vals, list(g[f"values_{c}"]))


def test_component_oracle_known_answers():
    """maximum_3d (util.py:58-75) / evaluation_case's component rule on cases whose answers follow from the definitions."""
    import components_oracle as co
    v = np.zeros((12, 12, 12), dtype=np.uint8)
    v[1:4, 1:4, 1:4] = 1                       # 27 voxels, reaches no test slice (z = 6, 4, 8)
    v[6:9, 6:9, 3:9] = 1                       # 54 voxels, reaches them
    v[4, 4, 4] = 1                             # 26-connected bridge? (3,3,3)-(4,4,4)-(5,5,5): corner contacts only
    v[5, 5, 4] = 1                             # (4,4,4)-(5,5,4) face-diagonal contact; (5,5,4)-(6,6,3): corner contact
    big = co.largest_component(v)
    assert big.sum() == 27 + 54 + 2            # everything hangs together through corner / edge contacts (26-connectivity)
    v2 = np.zeros((12, 12, 12), dtype=np.uint8)
    v2[0:5, 0:5, 0:3] = 1                      # 75 voxels, z in 0..2: misses z = 6, 4, 8
    v2[7:10, 7:10, 3:9] = 1                    # 54 voxels
    assert co.largest_component(v2).sum() == 75
    m = co.maximum_3d(v2)
    assert m.dtype == bool and m.sum() == 54 and m[8, 8, 5]          # the second largest is taken
    shell = np.zeros((9, 9, 9), dtype=np.uint8)
    shell[1:8, 1:8, 1:8] = 1
    shell[3:6, 3:6, 3:6] = 0                   # enclosed cavity of 27 voxels
    assert co.largest_component(shell).sum() == 343 - 27 and co.maximum_3d(shell).sum() == 343
    shell[4, 4, 0:4] = 0                       # a 6-connected channel to the border: no longer a hole
    assert co.maximum_3d(shell).sum() == 343 - 27 - 2
    tie = np.zeros((6, 6, 12), dtype=np.uint8)
    tie[0, 0, 0:3] = 1
    tie[5, 5, 5:8] = 1                         # equal sizes: the reference takes the HIGHER label = later in raster order
    assert co.largest_component(tie)[5, 5, 6] == 1 and co.largest_component(tie)[0, 0, 1] == 0
    with pytest.raises(IndexError):
        co.maximum_3d(np.zeros((4, 4, 4), dtype=np.uint8))
    assert co.largest_component(np.zeros((4, 4, 4), dtype=np.uint8)).sum() == 0
