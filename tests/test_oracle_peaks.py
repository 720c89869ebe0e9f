"""CPU: pin oracle.peaks against (a) the reference's own KAT (test/test_image_proc.py:94-120),
(b) outputs of the real reference functions stored in tests/golden/peaks_golden.npz,
(c) the installed scipy / numpy for the two third-party pieces it restates."""
import os

import numpy as np
import pytest
import scipy.ndimage

import cases
from oracle import peaks as op


def test_reference_kat_belief_maps():
    # same construction as the reference's test_belief_maps
    res = (80, 60)
    kp = np.array([65.0, 20.0])
    kp_out = np.array([res[0] + 20.0, res[1] + 20.0])
    maps = op.create_belief_map(res, [kp, kp_out]).astype(np.float32)
    peaks = op.peaks_from_belief_maps(maps, 0.0)
    assert len(peaks[0]) == 1
    assert np.linalg.norm(kp - np.array(peaks[0][0][:2])) < 1.0e-3
    assert len(peaks[1]) == 0


@pytest.mark.parametrize("name", sorted(cases.belief_map_cases()))
def test_create_belief_map_matches_reference_outputs(name, golden_dir):
    """oracle.peaks.create_belief_map (the yardstick of the on-device renderer) pinned to the reference's own outputs:
    float64, bit for bit, incl. border rejection and int() truncation of float64 coordinates."""
    res, pts, sigma = cases.belief_map_cases()[name]
    gold = np.load(os.path.join(golden_dir, "belief_map_golden.npz"))[name]
    got = op.create_belief_map(res, [tuple(p) for p in pts], sigma=sigma)
    assert got.dtype == np.float64 and np.array_equal(got, gold)


@pytest.mark.parametrize("shape", [(25, 33), (100, 100), (60, 80), (208, 208), (5, 7), (1, 40), (13, 2)])
def test_gaussian_restatement_bit_exact_vs_scipy(shape):
    rs = np.random.RandomState(shape[0] * 1000 + shape[1])
    m = rs.normal(0, 1, shape).astype(np.float32)
    assert np.array_equal(op.gaussian_filter_sigma3(m), scipy.ndimage.gaussian_filter(m, sigma=3))


def test_gaussian_weights_kat():
    w = op.gaussian_weights()
    assert w.shape == (25,) and abs(w.sum() - 1.0) < 1e-15 and np.array_equal(w, w[::-1])
    from dream_amd import image_proc
    # the product hard-codes the 13 distinct taps; they must be exactly what scipy would compute
    assert [float(v).hex() for v in w[:13]] == list(image_proc.GAUSS_SIGMA3_HALF_TAPS_HEX)


def test_pairwise_sum_matches_numpy():
    rs = np.random.RandomState(3)
    for _ in range(500):
        a = rs.normal(0, 1, (5, 5)) * rs.uniform(1e-3, 1e3)
        assert a.sum() == op.numpy_pairwise_sum25(a.ravel())


def _golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "peaks_golden.npz"))


@pytest.mark.parametrize("name", sorted(cases.peak_cases().keys()))
def test_peaks_match_reference_outputs(name):
    g = _golden()
    maps, off = cases.peak_cases()[name]
    pk = op.peaks_from_belief_maps(maps, off)
    counts = np.array([len(p) for p in pk])
    assert np.array_equal(counts, g[name + "/counts"])
    flat = [q for p in pk for q in p]
    if flat:
        xy = np.array([[q[0], q[1]] for q in flat], np.float64)
        assert np.array_equal(xy, g[name + "/xy"])                      # float64, bit-exact
        assert np.array_equal(np.array([q[2] for q in flat], np.float32), g[name + "/score"])
        assert np.array_equal(np.array([q[3] for q in flat]), g[name + "/id"])
    kps = op.keypoints_from_belief_maps(maps[None], off)
    assert kps.dtype == np.float32
    assert np.array_equal(kps, g[name + "/keypoints"])                  # float32, bit-exact


def test_selection_rule_cases_present():
    g = _golden()
    k = g["two_blobs_gap/keypoints"][0]
    # gaps 0, .1, .2499 -> rejected; .25 (float32 difference), .2501, .4 -> accepted; .9 -> see counts
    assert (k[:3, 0] == np.float32(-999.999)).all()
    assert (k[4:6, 0] > 0).all()
    assert (g["negative_zero/keypoints"][0][2] == np.float32(-999.999)).all()


@pytest.mark.parametrize("name", sorted(cases.peak_cases().keys()))
def test_c_restatement_matches_reference_outputs(name):
    import __graft_entry__
    __graft_entry__.build_oracle(verbose=False)
    g = _golden()
    maps, off = cases.peak_cases()[name]
    kps, counts = op.c_keypoints_from_belief_maps(maps[None], off)
    assert np.array_equal(kps, g[name + "/keypoints"])
    assert np.array_equal(counts[0], g[name + "/counts"])
