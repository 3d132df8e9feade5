"""GPU double_threshold_iteration (csrc/dti.hip through seunet_dti) against the reference-generated fixture and the C
oracle: bit-exact (integer / bit work)."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _A():
    import seunet_amd as A
    return A


def test_fixture_parity(golden_dir):
    A = _A()
    d = np.load(os.path.join(golden_dir, "dti_known.npz"))
    for c in range(int(d["n"])):
        got = A.double_threshold_iteration(d[f"pred_{c}"], float(d[f"h_{c}"]), float(d[f"l_{c}"]))
        assert isinstance(got, np.ndarray) and got.dtype == np.float64           # what the reference returns
        np.testing.assert_array_equal(got.astype(np.uint8), d[f"out_{c}"])


def test_fixture_parity_float32_copies(golden_dir):
    """train.py:25-49 / test.py:18-42 (pred*255 rounded to float32): fixture from those copies, including volumes with
    voxels within a float32 ulp of the thresholds, on which the float64 copy (prediction.py) gives a different answer."""
    A = _A()
    d = np.load(os.path.join(golden_dir, "dti_known_f32.npz"))
    differ = 0
    for c in range(int(d["n"])):
        v, h, l = d[f"pred_{c}"], float(d[f"h_{c}"]), float(d[f"l_{c}"])
        got32 = A.double_threshold_iteration(v, h, l, pred_dtype="float32").astype(np.uint8)
        got64 = A.double_threshold_iteration(v, h, l, pred_dtype="float64").astype(np.uint8)
        np.testing.assert_array_equal(got32, d[f"out_{c}"])
        np.testing.assert_array_equal(got64, d[f"out64_{c}"])
        differ += int((got32 != got64).sum())
    assert differ > 0


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 1, 200), (5, 7, 63), (5, 7, 64), (5, 7, 65), (17, 3, 129), (3, 40, 191),
                                   (40, 3, 64), (33, 31, 257), (64, 64, 64), (128, 128, 128)])
@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_against_c_oracle(shape, kind):
    import dti_oracle as do
    A = _A()
    rng = np.random.default_rng(hash((shape, kind)) % (2 ** 32))
    v = rng.random(shape)
    if kind == "smooth":
        for ax in range(3):
            v = (v + np.roll(v, 1, ax) + np.roll(v, -1, ax)) / 3.0
        v = (v - v.min()) / max(v.max() - v.min(), 1e-9)
    for h, l in ((0.5, 0.4), (0.62, 0.37)):
        for pd in ("float64", "float32"):
            want = do.double_threshold_iteration(v, h, l, pd)
            got = A.double_threshold_iteration(torch.from_numpy(v).cuda(), h, l, pred_dtype=pd)
            assert got.dtype == torch.uint8 and got.is_cuda
            np.testing.assert_array_equal(got.cpu().numpy(), want.astype(np.uint8))


def test_thresholds_exactly_on_the_boundary():
    """pred*255 >= h*255 is evaluated in float64 like the reference: values equal to the thresholds are strong / weak."""
    import dti_oracle as do
    A = _A()
    v = np.zeros((2, 2, 8))
    v[0, 0, :] = [0.5, 0.4, 0.39999999999999997, 0.49999999999999994, 0.4, 0.4, 0.5000000000000001, 0.0]
    want = do.double_threshold_iteration(v, 0.5, 0.4)
    np.testing.assert_array_equal(A.double_threshold_iteration(v, 0.5, 0.4), want)


def test_full_size_volume_properties_and_speed():
    """512^3 (BASELINE configs[3] volume): strong <= result <= strong | weak, equal to the oracle on a 512x512x16 slab, timing."""
    import dti_oracle as do
    A = _A()
    g = torch.Generator(device="cuda").manual_seed(5)
    v = torch.rand((512, 512, 512), generator=g, device="cuda", dtype=torch.float64)
    v = (v + v.roll(1, 0) + v.roll(1, 1) + v.roll(1, 2)) / 4.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = A.double_threshold_iteration(v, 0.5, 0.4)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    strong, weak = v * 255.0 >= 0.5 * 255, (v * 255.0 >= 0.4 * 255) & (v * 255.0 < 0.5 * 255)
    assert bool((out[strong] == 1).all()) and bool((out[~(strong | weak)] == 0).all())
    assert int(out.sum()) > int(strong.sum())
    slab = v[:, :, :16].contiguous()
    want = do.double_threshold_iteration(slab.cpu().numpy(), 0.5, 0.4)
    np.testing.assert_array_equal(A.double_threshold_iteration(slab, 0.5, 0.4).cpu().numpy(), want.astype(np.uint8))
    print(f"\ndouble_threshold_iteration 512^3 on the GPU: {dt * 1e3:.1f} ms ({512 ** 3 / dt / 1e9:.2f} Gvoxel/s)")


def test_postprocess_prediction_pipeline():
    import dti_oracle as do
    A = _A()
    rng = np.random.default_rng(9)
    v = rng.random((40, 40, 70))
    want = do.double_threshold_iteration(v, 0.5, 0.4)
    want[0:6] = 0; want[34:] = 0; want[:, 0:6] = 0; want[:, 34:] = 0          # int(0.15*40) = 6, int(0.85*40) = 34
    np.testing.assert_array_equal(A.postprocess_prediction(v), want)


def test_bad_arguments_report_errors():
    A = _A()
    lib = A._lib.load()
    assert lib.seunet_dti_workspace_bytes(0, 4, 4) == 0
    out = torch.empty(8, dtype=torch.uint8, device="cuda")
    p = torch.zeros(8, dtype=torch.float64, device="cuda")
    assert lib.seunet_dti(p.data_ptr(), 2, 2, 2, 0.5, 0.4, 0, out.data_ptr(), out.data_ptr(), 1, None) != 0   # workspace too small
    assert "workspace" in A._lib.last_error()
