"""The pin route of the oracle: tests/golden/matlab_outputs.mat holds what the REFERENCE returns on tests/golden/matlab_inputs.mat
(oracle/matlab/make_fixtures.m, run by someone with MATLAB).  While that file is absent -- the build image has no MATLAB / Octave -- the
comparisons are skipped and only the inputs are checked; once it is committed, every function of the oracle is held against MATLAB's own
numbers and DESIGN.md's "parity unpinned" can go."""
import os

import numpy as np
import pytest
import scipy.io as sio
import scipy.sparse as sp

import cnmfe_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IN = os.path.join(GOLD, "matlab_inputs.mat")
OUT = os.path.join(GOLD, "matlab_outputs.mat")
needs_matlab = pytest.mark.skipif(not os.path.exists(OUT), reason="tests/golden/matlab_outputs.mat not generated (oracle/matlab/make_fixtures.m needs MATLAB or Octave)")

# what every field of matlab_outputs.mat pins: field (or prefix) -> (oracle function, reference file:line, produced by a MathWorks-only toolbox function?)
PINS = {
    "nhood_r": ("cnmfe_oracle.get_nhood", "ca_source_extraction/endoscope/get_nhood.m:1-25", False),
    "fit_W": ("cnmfe_oracle.fit_ring_model (W: first run, second run, no projection, outlier branch)", "endoscope/fit_ring_model.m:1-127", False),
    "fit_b0": ("cnmfe_oracle.fit_ring_model (b0)", "endoscope/fit_ring_model.m:44", False),
    "Ysig": ("cnmfe_oracle.residual_ysig", "@Sources2D/update_spatial_parallel.m:162-166", False),
    "IND_ellipse": ("cnmfe_oracle.determine_search_location", "utilities/determine_search_location.m:57-89", False),
    "A_hals": ("cnmfe_oracle.HALS_spatial", "utilities/HALS_spatial.m:27-44", False),
    "A_thresh": ("cnmfe_oracle.HALS_spatial_thresh", "utilities/HALS_spatial_thresh.m:30-53", False),
    "A_nnls": ("cnmfe_oracle.nnls_spatial / nnls", "endoscope/nnls_spatial.m:26-109", False),
    "cm": ("cnmfe_oracle.com", "utilities/com.m:20-28", False),
    "C_hals": ("cnmfe_oracle.HALS_temporal (C)", "utilities/HALS_temporal.m:47-68", False),
    "Craw_hals": ("cnmfe_oracle.HALS_temporal (C_raw)", "utilities/HALS_temporal.m:62-68", False),
    "conn": ("cnmfe_oracle.connectivity_constraint (imopen, bwlabel restated)", "endoscope/connectivity_constraint.m:12-18", True),
    "circ": ("cnmfe_oracle.circular_constraints (medfilt2, imdilate restated)", "endoscope/circular_constraints.m:1-55", True),
    "thr_comp": ("cnmfe_oracle.threshold_components", "utilities/threshold_components.m", True),
    "IND_dilate": ("cnmfe_oracle.determine_search_location_dilate", "utilities/determine_search_location.m:89-98", True),
    "resize_": ("cnmfe_oracle.imresize_scale / imresize_size (MathWorks imresize restated)", "bg_ssub > 1: update_background_parallel.m:137,224", True),
    "quant": ("cnmfe_oracle.matlab_quantile", "endoscope/fit_ring_model.m:64; OASIS foopsi_oasisAR1.m:93", True),
    "medfilt": ("cnmfe_oracle._medfilt3", "endoscope/circular_constraints.m:33", True),
    "sn_tr": ("oasis_oracle.GetSn (pwelch restated)", "OASIS_matlab/functions/GetSn.m:33-47", True),
    "g0_tr": ("oasis_oracle.estimate_time_constant_ar1", "OASIS_matlab/functions/estimate_time_constant.m:35-66", False),
    "g_tr": ("oasis_oracle.deconvolveCa_ar1_foopsi (gamma; fminbnd restated)", "OASIS_matlab/packages/oasis/foopsi_oasisAR1.m:122-179", True),
    "b_tr": ("oasis_oracle.deconvolveCa_ar1_foopsi (baseline)", "foopsi_oasisAR1.m:93-98", True),
    "c_tr": ("oasis_oracle.oasisAR1 via deconvolveCa (denoised trace)", "OASIS_matlab/packages/oasis/oasisAR1.m:30-109", True),
    "s_tr": ("oasis_oracle.oasisAR1 via deconvolveCa (spikes)", "oasisAR1.m:109", True),
    "engine": ("(which interpreter wrote the file)", "-", False),
    "skipped": ("(groups the interpreter could not run)", "-", False),
}


def _pin_of(field):
    return next((v for k, v in PINS.items() if field == k or field.startswith(k)), None)


def test_pin_map_covers_every_field_the_script_writes(capsys):
    """every `out.<field>` of oracle/matlab/make_fixtures.m is named in PINS -- the table below is what a committed matlab_outputs.mat pins
    (run with -s to read it)"""
    import re
    src = open(os.path.join(os.path.dirname(GOLD), "..", "oracle", "matlab", "make_fixtures.m")).read()
    fields = sorted(set(re.findall(r"out\.([A-Za-z_0-9]+)\b", src)) | {"nhood_r15"})
    missing = [f for f in fields if _pin_of(f) is None]
    assert not missing, missing
    have = sio.loadmat(OUT) if os.path.exists(OUT) else {}
    with capsys.disabled():
        print("\nmatlab_outputs.mat: %s" % ("present, written by %s" % "".join(np.ravel(have.get("engine", ["?"]))) if have else "ABSENT -- nothing below is pinned yet"))
        for f in fields:
            fn, ref, toolbox = _pin_of(f)
            print("  %-22s %-7s pins %-70s <- %s%s" % (f, "[have]" if any(k.startswith(f) for k in have) else "[none]", fn, ref, "   (MathWorks-toolbox semantics: MATLAB only)" if toolbox else ""))


def _engine_is_matlab(o):
    return "".join(np.ravel(o.get("engine", ["matlab"]))).lower().startswith("matlab")


def _need(o, *keys):
    miss = [k for k in keys if k not in o]
    if miss:
        pytest.skip("matlab_outputs.mat has no %s (group skipped by the interpreter that wrote it: %s)" % (miss, "".join(np.ravel(o.get("skipped", [""])))))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="module")
def case():
    z = sio.loadmat(IN)
    d1, d2, T = int(z["d1"].item()), int(z["d2"].item()), int(z["T"].item())
    block = z["block"].ravel().astype(int)
    mask = np.zeros((d1, d2), bool); mask[block[0] - 1:block[1], block[2] - 1:block[3]] = True
    mask = mask.reshape(-1, order="F")
    c = dict(z=z, d1=d1, d2=d2, T=T, mask=mask, ip=z["ind_patch"].ravel().astype(bool), Yb=z["Y"][mask], Ab=sp.csc_matrix(z["A"])[mask],
             C=z["C"], sn=z["sn"].ravel(), W0=sp.csr_matrix(z["W0"]))
    c["out"] = sio.loadmat(OUT) if os.path.exists(OUT) else None
    return c


def test_inputs_are_the_seeded_ones(case):
    """the committed inputs are what tests/golden/make_matlab_inputs.py builds (so the MATLAB run and this test see the same data)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mmi", os.path.join(GOLD, "make_matlab_inputs.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    ref = m.build()
    z = case["z"]
    assert np.array_equal(z["Y"], ref["Y"]) and np.array_equal(z["C"], ref["C"]) and np.array_equal(sp.csc_matrix(z["A"]).toarray(), ref["A"].toarray())
    assert np.array_equal(sp.csc_matrix(z["W0"]).toarray(), ref["W0"].toarray()) and np.array_equal(z["imgs"], ref["imgs"])
    # the oracle runs on them (so a MATLAB run has something to be compared with)
    W1, b0 = orc.fit_ring_model(case["Yb"], case["Ab"], case["C"], case["W0"], np.nan, case["sn"][case["mask"]][case["ip"]], case["ip"], True)
    assert np.isfinite(W1.data).all() and b0.shape == (int(case["ip"].sum()),)


@needs_matlab
def test_ring_geometry(case):
    o = case["out"]
    _need(o, "nhood_r15", "nhood_r15_k40")
    for r in (3, 5, 15, 18):
        rs, cs = orc.get_nhood(r)
        assert np.array_equal(np.c_[np.ravel(rs), np.ravel(cs)], o["nhood_r%d" % r])
    rs, cs = orc.get_nhood(15, 40)
    assert np.array_equal(np.c_[np.ravel(rs), np.ravel(cs)], o["nhood_r15_k40"])


@needs_matlab
def test_fit_ring_model_and_residual(case):
    o = case["out"]; c = case
    _need(o, "fit_W1", "fit_W4", "Ysig")
    snp = c["sn"][c["mask"]][c["ip"]]
    W1, b01 = orc.fit_ring_model(c["Yb"], c["Ab"], c["C"], c["W0"], np.nan, snp, c["ip"], True)
    W2, b02 = orc.fit_ring_model(c["Yb"], c["Ab"], c["C"], W1, np.nan, snp, c["ip"], True)
    W3, _ = orc.fit_ring_model(c["Yb"], c["Ab"], c["C"], c["W0"], np.nan, snp, c["ip"], False)
    W4, b04 = orc.fit_ring_model(c["Yb"], c["Ab"], c["C"], c["W0"], float(c["z"]["thresh_outlier"].item()), snp, c["ip"], True)
    for got, key in ((W1, "fit_W1"), (W2, "fit_W2"), (W3, "fit_W3"), (W4, "fit_W4")):
        assert rel(got.toarray(), o[key]) <= 1e-8, key
    assert rel(b01, o["fit_b01"].ravel()) <= 1e-12 and rel(b04, o["fit_b04"].ravel()) <= 1e-12
    Ysig = orc.residual_ysig(c["Yb"], c["Ab"], c["C"], W1, b01, c["ip"])
    assert rel(Ysig, o["Ysig"]) <= 1e-9


@needs_matlab
def test_spatial_temporal(case):
    o = case["out"]; c = case
    _need(o, "IND_ellipse", "A_hals", "A_nnls", "C_hals", "Ysig")
    IND = orc.determine_search_location(c["z"]["A"], c["d1"], c["d2"])
    assert np.array_equal(IND, o["IND_ellipse"].astype(bool))
    INDp = sp.csc_matrix(IND[c["mask"]][c["ip"]])
    Ap = c["Ab"].toarray()[c["ip"]]; snp = c["sn"][c["mask"]][c["ip"]]
    Ysig = o["Ysig"]
    assert rel(orc.HALS_spatial(Ysig, Ap, c["C"], INDp, 3), o["A_hals"]) <= 1e-9
    assert rel(orc.HALS_spatial_thresh(Ysig, Ap, c["C"], INDp, 3, snp), o["A_thresh"]) <= 1e-9
    assert rel(orc.nnls_spatial(Ysig, Ap, c["C"], INDp, int(c["z"]["maxN"].item())), o["A_nnls"]) <= 1e-8
    assert rel(orc.com(c["z"]["A"], c["d1"], c["d2"]), o["cm"]) <= 1e-12
    Ct, Crawt, _ = orc.HALS_temporal(Ysig, o["A_hals"], c["C"], 5)
    assert rel(Ct, o["C_hals"]) <= 1e-9 and rel(Crawt, o["Craw_hals"]) <= 1e-9


@needs_matlab
def test_post_processing_and_optional_branches(case):
    o = case["out"]; c = case
    _need(o, "conn", "circ", "thr_comp", "IND_dilate")
    if not _engine_is_matlab(o):
        pytest.skip("written by Octave: imopen / bwlabel / medfilt2 / imdilate there are not the MathWorks implementations the restatement follows")
    imgs = c["z"]["imgs"]
    for k in range(imgs.shape[2]):
        assert rel(orc.connectivity_constraint(imgs[:, :, k]), o["conn"][:, :, k]) <= 1e-12, k
        assert rel(orc.circular_constraints(imgs[:, :, k]), o["circ"][:, :, k]) <= 1e-12, k
    A = imgs.reshape(c["d1"] * c["d2"], -1, order="F")
    assert rel(orc.threshold_components(A, c["d1"], c["d2"], nb=1, nrgthr=0.99), o["thr_comp"]) <= 1e-12
    assert np.array_equal(orc.determine_search_location_dilate(A, c["d1"], c["d2"], orc.strel_disk(4), nb=1, nrgthr=0.99), o["IND_dilate"].astype(bool))


@needs_matlab
def test_toolbox_restatements(case):
    o = case["out"]; z = case["z"]
    _need(o, "resize_half", "quant", "medfilt")
    if not _engine_is_matlab(o):
        pytest.skip("written by Octave: imresize / quantile / medfilt2 there are not the MathWorks implementations the restatement follows")
    img = z["resize_img"]
    assert rel(orc.imresize_scale(img, 1 / 2), o["resize_half"]) <= 1e-10
    assert rel(orc.imresize_scale(img, 1 / 3), o["resize_third"]) <= 1e-10
    assert rel(orc.imresize_scale(img, 1 / 2, "nearest"), o["resize_half_nearest"]) == 0
    assert rel(orc.imresize_size(o["resize_half"], img.shape[:2]), o["resize_up"]) <= 1e-10
    assert rel(orc.imresize_size(o["resize_half_nearest"], img.shape[:2], "nearest"), o["resize_up_nearest"]) == 0
    assert rel([orc.matlab_quantile(z["quant_x"].ravel(), p) for p in z["quant_p"].ravel()], o["quant"].ravel()) <= 1e-14
    assert rel(orc._medfilt3(z["imgs"][:, :, 0]), o["medfilt"]) == 0


@needs_matlab
def test_oasis(case):
    import oasis_oracle as oo
    o = case["out"]; tr = case["z"]["traces"]
    _need(o, "sn_tr", "g_tr", "c_tr")
    if not _engine_is_matlab(o):
        pytest.skip("written by Octave: pwelch and fminbnd there are not the MathWorks implementations the restatement follows")
    for i in range(tr.shape[0]):
        sn = oo.GetSn(tr[i])
        assert abs(sn - float(o["sn_tr"][i])) <= 1e-9 * sn
        c, s, b, g = oo.deconvolveCa_ar1_foopsi(tr[i], sn, smin=-5.0, optimize_pars=True, optimize_b=True)
        assert abs(g - float(o["g_tr"][i])) <= 2e-4 and abs(b - float(o["b_tr"][i])) <= 1e-3 * max(1.0, abs(b))     # fminbnd TolX = 1e-4
        assert rel(c, o["c_tr"][i]) <= 5e-3
