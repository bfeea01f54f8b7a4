"""The MathWorks toolbox functions the oracle restates from their documentation (imresize, quantile, medfilt2, imdilate / imerode / imopen,
bwlabel) against INDEPENDENT implementations of the same published definitions (Pillow, NumPy, SciPy).  None of these is MATLAB output -- the pin
route for that is tests/test_matlab_fixtures.py -- but each removes the possibility that the restatement misreads its own definition."""
import numpy as np
import pytest

import cnmfe_oracle as orc


def _pil_resize(img, out_hw, resample):
    from PIL import Image
    im = Image.fromarray(np.asarray(img, dtype=np.float32), mode="F")
    return np.asarray(im.resize((out_hw[1], out_hw[0]), resample=resample), dtype=np.float64)


@pytest.mark.parametrize("scale", [0.5, 1 / 3])
def test_imresize_bicubic_antialiased_matches_pillow_in_the_interior(scale):
    """Pillow's BICUBIC resize is the same construction MATLAB documents for imresize: Keys kernel a = -0.5, centre-aligned sample positions, kernel
    stretched by 1/scale when shrinking, weights normalised.  The two differ only in how the window is closed at the image border (MATLAB mirrors,
    Pillow clips and renormalises), so the comparison leaves a rim out."""
    from PIL import Image
    rng = np.random.default_rng(4)
    img = rng.standard_normal((48, 60)).cumsum(axis=0).cumsum(axis=1)
    got = orc.imresize_scale(img, scale)
    ref = _pil_resize(img, got.shape, Image.BICUBIC)
    rim = 3
    e = np.abs(got - ref)[rim:-rim, rim:-rim].max() / np.abs(ref).max()
    assert got.shape == (int(np.ceil(48 * scale)), int(np.ceil(60 * scale))) and e <= 2e-6, e      # Pillow computes in float32


def test_imresize_bicubic_upsampling_matches_pillow_in_the_interior():
    from PIL import Image
    rng = np.random.default_rng(5)
    img = rng.standard_normal((20, 24)).cumsum(axis=0)
    got = orc.imresize_size(img, [40, 47])
    ref = _pil_resize(img, (40, 47), Image.BICUBIC)
    e = np.abs(got - ref)[5:-5, 5:-5].max() / np.abs(ref).max()
    assert e <= 2e-6, e


def test_imresize_nearest_is_a_pixel_selection_with_matlabs_rounding():
    """'nearest' = box kernel on u = x/scale + 0.5*(1 - 1/scale): for integer factors s the kept index is s*x - floor((s - 1)/2) (1-based): 2 -> 2x,
    3 -> 3x - 1, 4 -> 4x - 1 -- MATLAB's documented behaviour of picking the pixel whose centre is nearest, ties towards the larger index."""
    for s in (2, 3, 4, 5):
        n = 37
        M = orc.imresize_weights(n, -(-n // s), 1.0 / s, "nearest")
        sel = M.argmax(axis=1) + 1
        x = np.arange(1, M.shape[0] + 1)
        expect = np.minimum(np.floor(x * s + 0.5 * (1 - s) + 0.5).astype(int), 2 * n + 1 - np.floor(x * s + 0.5 * (1 - s) + 0.5).astype(int))
        assert np.array_equal(M.sum(axis=1), np.ones(M.shape[0])) and np.array_equal(sel, expect), s


def test_quantile_is_the_hazen_definition():
    """quantile(x, p): sorted values at the (i - 0.5)/n quantiles, linear interpolation, extremes outside == numpy's method='hazen'"""
    rng = np.random.default_rng(6)
    for n in (1, 2, 7, 100, 3001):
        x = rng.integers(0, 50, n).astype(float)
        for p in (0.0, 0.01, 0.2, 0.5, 0.8, 0.999, 1.0):
            assert abs(orc.matlab_quantile(x, p) - np.quantile(x, p, method="hazen")) <= 1e-12, (n, p)


def test_morphology_restatements_match_scipy():
    import scipy.ndimage as ndi
    rng = np.random.default_rng(7)
    img = rng.random((23, 31)) * (rng.random((23, 31)) > 0.3)
    assert np.array_equal(orc._medfilt3(img), ndi.median_filter(img, size=3, mode="constant", cval=0.0))
    bw = img > 0.5
    for se in (np.ones((3, 3), bool), orc.strel_disk(2), orc.strel_disk(4)):
        assert np.array_equal(orc._bw_dilate(bw, se), ndi.binary_dilation(bw, structure=se, border_value=0))
        assert np.array_equal(orc._bw_erode(bw, se), ndi.binary_erosion(bw, structure=se, border_value=1))
    # grey opening with a flat 5 x 5 square, -Inf / +Inf outside (connectivity_constraint.m:12-13)
    er = ndi.grey_erosion(img, size=(5, 5), mode="constant", cval=np.inf)
    assert np.allclose(orc._imopen_square(img, 5), ndi.grey_dilation(er, size=(5, 5), mode="constant", cval=-np.inf))
    for conn, st in ((4, [[0, 1, 0], [1, 1, 1], [0, 1, 0]]), (8, np.ones((3, 3)))):
        lab, n = orc._flood_label(bw, conn)
        ref, nref = ndi.label(bw, structure=st)
        assert n == nref
        # same partition (label numbering may differ: MATLAB numbers by column-major first pixel, SciPy by row-major)
        pairs = set(zip(lab[bw].tolist(), ref[bw].tolist()))
        assert len(pairs) == n
