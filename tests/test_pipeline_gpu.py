"""GPU input pipeline (csrc/pipeline.hip through seunet_crop_batch / seunet_hu_two_channel) against the
fixture generated from the reference's own data.py helpers and against the numpy oracle.  Index work and IEEE divisions
are bit-exact; the weight power (float16 / float32 / float64 `**`) is compared bit-exactly on the fixture and to <= 1 ulp
of the array's dtype on random data (libm's pow vs the device's)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import seunet_amd
    seunet_amd._lib.load()
    return seunet_amd


@pytest.fixture(scope="module")
def po():
    import pipeline_oracle
    return pipeline_oracle


def _case(seed, shape):
    rng = np.random.default_rng(seed)
    img = rng.integers(-1500, 1700, shape).astype(np.int16)
    label = (rng.random(shape) < 0.2).astype(np.uint8)
    w16 = (rng.random(shape) * 2.6).astype(np.float16)
    skel = (label * (rng.random(shape) < 0.3)).astype(np.uint8)
    return img, label, w16, skel


def test_reference_fixture_stage1_batches_bit_exact(A, golden_dir):
    """Seeded CropSegData batches (data.py:689-715 from the reference's own helpers): same draws, same tensors."""
    g = np.load(os.path.join(golden_dir, "pipeline_known.npz"))
    img, label, w16, cube, b = g["img"], g["label"], g["weight16"], int(g["cube"]), int(g["batch"])
    ds = A.CropSegDataGPU(torch.from_numpy(img).cuda(), torch.from_numpy(label).cuda(), torch.from_numpy(w16).cuda(), b, cube=cube)
    for seed in (1, 2):
        random.seed(100 + seed)
        np.random.seed(200 + seed)
        out = ds.sample()
        for k in ("data", "label", "weight"):
            got, want = out[k].cpu().numpy(), g[f"s1_{seed}_{k}"]
            assert got.dtype == np.float32 and got.shape == want.shape
            np.testing.assert_array_equal(got, want, err_msg=f"seed {seed} {k}")
    # int16 crops, float64 division (AirwayHMData.process_img, data.py:286-299)
    out = A.crop_batch(torch.from_numpy(img).cuda(), [tuple(int(v) for v in s) for s in g["s2_starts"]], cube)
    np.testing.assert_array_equal(out["data"].cpu().numpy(), g["s2_data"])


def test_reference_fixture_stage2_stage3_batches_bit_exact(A, golden_dir):
    """Seeded AirwayHMData / AirwayHMData3 batches (data.py:359-408, 546-584) produced by the reference's own crop / process_img /
    augment methods and sampler functions: the GPU samplers take the same draws and build the same tensors, bit for bit (int16
    crops normalised in float64, float16 weight power, label / skeleton crops, flips and axis swaps)."""
    from scipy import ndimage
    g = np.load(os.path.join(golden_dir, "pipeline_hm_known.npz"))
    img, label, skeleton, pred, cube, b = g["img"], g["label"], g["skeleton"], g["pred"], int(g["cube"]), int(g["batch"])
    dis = ndimage.distance_transform_edt(label)
    loc_skel, loc_small, loc_break = np.where(skeleton * (1 - pred)), np.where((dis * skeleton) < 2), tuple(g["br_skel"])
    dev = lambda a: torch.from_numpy(a).cuda()
    ds2 = A.AirwayHMDataGPU(dev(img), dev(label), dev(g["weight16"]), loc_skel, loc_small, b, cube=cube)
    # stage 3's weight volume = LIB + 0.6 * break weight, formed in float16 like data.py:553-557
    ds3 = A.AirwayHMData3GPU(dev(img), dev(label), dev(g["weight3"]), dev(skeleton), loc_skel, loc_small, loc_break, b, cube=cube)
    for seed in (1, 2):
        for stage, ds in ((2, ds2), (3, ds3)):
            random.seed(300 + seed)
            np.random.seed(400 + seed)
            out = ds.sample()
            for k in ("data", "label", "weight") + (("skel",) if stage == 3 else ()):
                got, want = out[k].cpu().numpy(), g[f"s{stage}_{seed}_{k}"]
                assert got.dtype == np.float32 and got.shape == want.shape
                np.testing.assert_array_equal(got, want, err_msg=f"stage {stage} seed {seed} {k}")


@pytest.mark.parametrize("code", list(range(16)))
def test_every_axis_map_against_oracle(A, po, code):
    img, label, w16, skel = _case(code, (70, 66, 100))
    starts = [(3, 1, 36), (6, 2, 0)]
    t = lambda a: torch.from_numpy(a).cuda()
    got = A.crop_batch(t(img), starts, 64, t(label), t(w16), t(skel), [code, 15 - code], u=0.37)
    want = po.crop_batch(img, starts, [code, 15 - code], 64, label, w16, skel, 0.37)
    for k in ("data", "label", "skel"):
        np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=k)
    # float16 power: numpy rounds powf to half; allow one half ulp step (measured: identical)
    gw, ww = got["weight"].cpu().numpy(), want["weight"]
    assert np.array_equal(gw, ww) or np.abs(gw - ww).max() <= np.spacing(np.float16(ww.max())).astype(np.float32)
    print("weight mismatches:", int((gw != ww).sum()), "of", gw.size)


@pytest.mark.parametrize("wdtype", [np.float32, np.float64])
def test_weight_power_in_wider_dtypes(A, po, wdtype):
    img, label, w16, _ = _case(5, (40, 40, 40))
    w = (w16.astype(np.float64) * 1.01).astype(wdtype)
    t = lambda a: torch.from_numpy(a).cuda()
    got = A.crop_batch(t(img), [(4, 4, 4)], 32, t(label), t(w), None, None, u=0.811)["weight"].cpu().numpy()
    want = po.crop_batch(img, [(4, 4, 4)], [0], 32, label, w, None, 0.811)["weight"]
    ulp = np.spacing(np.abs(want).astype(np.float32))       # numpy's float32 power is its own SIMD powf (not correctly rounded):
    assert np.all(np.abs(got - want) <= ulp), float(np.abs(got - want).max())     # measured 4 % of the voxels off by one ulp
    print("f32-rounded mismatches:", int((got != want).sum()), "of", got.size)


@pytest.mark.parametrize("dtype,f64", [(np.int16, True), (np.int16, False), (np.float32, False), (np.float32, True)])
def test_two_channel_volume(A, po, dtype, f64):
    """prediction.py:39-49,74 (float64 math then astype(float32)) and data.py:775-784 (float32 math): bit-exact."""
    rng = np.random.default_rng(11)
    hu = rng.integers(-1500, 1700, (37, 41, 53)).astype(dtype)
    got = A.two_channel_volume(torch.from_numpy(hu).cuda(), f64_math=f64).cpu().numpy()
    if f64:
        c0, c1 = po.two_channel(hu)
    else:
        c0, c1 = po.process_imgmsk(hu)
    want = np.stack([c0, c1])[None].astype(np.float32)
    assert got.shape == (1, 2, 37, 41, 53)
    np.testing.assert_array_equal(got, want)
    a0, a1 = A.two_channel(hu.astype(np.float64))            # the numpy helper of the package (prediction.py:39-49)
    if f64:
        np.testing.assert_array_equal(got[0, 0], a0.astype(np.float32))
        np.testing.assert_array_equal(got[0, 1], a1.astype(np.float32))


def test_full_size_batch_feeds_the_network(A, po):
    """8 crops of 128^3 from a 300 x 320 x 340 case (the reference's batch, train.py:141): properties at full size
    (ranges, label in {0,1}, weight == 1 off the mask, crop k of a batch == the same crop alone), and the tensors go
    straight into SE_UNet + the stage-2 loss."""
    img, label, w16, _ = _case(21, (300, 320, 340))
    t = lambda a: torch.from_numpy(a).cuda()
    ds = A.CropSegDataGPU(t(img), t(label), t(w16), batch_size=8)
    random.seed(5); np.random.seed(6)
    plan = A.draw_stage1_plan(img.shape, 8)
    out = A.crop_batch(ds.img, plan["starts"], 128, ds.label, ds.weight, None, plan["codes"], plan["u"], f64_math=False)
    d, l, w = out["data"], out["label"], out["weight"]
    assert d.shape == (8, 2, 128, 128, 128) and l.shape == w.shape == (8, 1, 128, 128, 128)
    assert float(d.min()) >= 0.0 and float(d.max()) <= 1.0
    assert set(torch.unique(l).tolist()) <= {0.0, 1.0} and bool((w[l == 0] == 1).all())
    one = A.crop_batch(ds.img, plan["starts"][5:6], 128, ds.label, ds.weight, None, plan["codes"][5:6], plan["u"], f64_math=False)
    assert torch.equal(one["data"][0], d[5]) and torch.equal(one["weight"][0], w[5])
    want = po.crop_batch(img.astype(np.float32), plan["starts"][5:6], plan["codes"][5:6], 128, label, w16, None, plan["u"], f64_math=False)
    np.testing.assert_array_equal(d[5].cpu().numpy(), want["data"][0])
    m = A.SE_UNet(2, 1).cuda()
    e, dd = m(d[:2])
    A.fused_stage_loss(2, e, dd, l[:2], w[:2]).backward()
    assert torch.isfinite(m.ec1.conv1.weight.grad).all()


def test_bad_arguments_fail_loudly(A):
    img = torch.zeros((40, 40, 40), dtype=torch.int16, device="cuda")
    with pytest.raises(RuntimeError, match="leaves"):
        A.crop_batch(img, [(9, 0, 0)], 32)
    with pytest.raises(RuntimeError, match="multiple of 32"):
        A.crop_batch(img, [(0, 0, 0)], 24)
    with pytest.raises(RuntimeError, match="no CPU path"):
        A.crop_batch(img.cpu(), [(0, 0, 0)], 32)
    with pytest.raises(TypeError):
        A.crop_batch(img.double(), [(0, 0, 0)], 32)
