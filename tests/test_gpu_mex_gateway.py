"""The MEX gateway (cnmf_e_amd/csrc/matlab/cnmfe_mex.cpp) RUN end to end: linked against libcnmfe_hip.so and a mock of the mx* / mex* runtime
(tests/mex_stub/mx_mock.cpp: this image has no MATLAB), driven with the call sequence of the three .m twins
(cnmf_e_amd/csrc/matlab/@Sources2D/update_{background,spatial,temporal}_parallel.m) -- including the commands of round 6 that the measured host path
uses (spatial_queue / spatial_collect, temporal_job / temporal_jobs_sweep / stitch_add_job, stitch_finish_async / stitch_collect, set_option, synchronize) --
on a 2 x 2-patch video, and compared with the Python mirror (cnmf_e_amd/sources2d.py) running the same iteration on the same library.
What this checks is the gateway's own work: argument marshalling (double / logical sparse -> int64 / int32 / float CSC, 1-based row lists, single / double
dense), the row-major <-> column-major conversions of its outputs, the pending-ticket bookkeeping, and that the .m twins' sequence of commands is one the
engine accepts.  What it cannot check is MathWorks' own mex.h / libmx."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_util import rel

pytestmark = pytest.mark.gpu
CLS = {np.dtype(np.float64): 6, np.dtype(np.float32): 7, np.dtype(np.int32): 12, np.dtype(np.bool_): 3, np.dtype(np.uint16): 11, np.dtype(np.uint8): 9}


class Mex:
    """cnmfe_mex(cmd, ...) through the mock runtime: numpy / scipy arguments in, numpy / scipy results out"""

    def __init__(self):
        stub = os.path.join(ROOT, "tests", "mex_stub")
        out = os.path.join(stub, "_build")
        os.makedirs(out, exist_ok=True)
        self.path = os.path.join(out, "libmex_mock.so")
        cmd = ["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wall", "-Wextra", os.path.join(stub, "mx_mock.cpp"),
               os.path.join(ROOT, "cnmf_e_amd", "csrc", "matlab", "cnmfe_mex.cpp"), "-I", stub, "-I", os.path.join(ROOT, "include"),
               "-L", os.path.join(ROOT, "cnmf_e_amd"), "-lcnmfe_hip", "-Wl,-rpath," + os.path.join(ROOT, "cnmf_e_amd"), "-o", self.path]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        L = self.L = C.CDLL(self.path)
        L.mock_dense.restype = C.c_void_p; L.mock_dense.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
        L.mock_string.restype = C.c_void_p; L.mock_string.argtypes = [C.c_char_p]
        L.mock_sparse.restype = C.c_void_p; L.mock_sparse.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mock_call.restype = C.c_int; L.mock_call.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        for f in ("mxGetM", "mxGetN", "mock_nnz"):
            getattr(L, f).restype = C.c_size_t; getattr(L, f).argtypes = [C.c_void_p]
        for f in ("mxGetData", "mxGetJc", "mxGetIr"):
            getattr(L, f).restype = C.c_void_p; getattr(L, f).argtypes = [C.c_void_p]
        for f in ("mock_class", "mock_is_sparse"):
            getattr(L, f).restype = C.c_int; getattr(L, f).argtypes = [C.c_void_p]
        L.mock_free.argtypes = [C.c_void_p]

    def _to_mx(self, a):
        L = self.L
        if isinstance(a, str):
            return L.mock_string(a.encode())
        if sp.issparse(a):
            a = a.tocsc(); a.sort_indices()
            jc = np.ascontiguousarray(a.indptr, dtype=np.int64); ir = np.ascontiguousarray(a.indices, dtype=np.int64)
            logical = a.dtype == np.bool_
            pr = np.ascontiguousarray(a.data, dtype=np.float64)
            return L.mock_sparse(a.shape[0], a.shape[1], jc.ctypes.data, ir.ctypes.data, pr.ctypes.data, int(logical))
        a = np.asarray(a)
        if a.dtype not in CLS:
            a = a.astype(np.float64)                             # MATLAB's default class
        a2 = np.atleast_2d(a) if a.ndim != 1 else a.reshape(-1, 1)   # vectors: columns
        f = np.asfortranarray(a2)
        return L.mock_dense(CLS[f.dtype], f.shape[0], f.shape[1], f.ctypes.data)

    def _from_mx(self, p):
        L = self.L
        m, n = L.mxGetM(p), L.mxGetN(p)
        cls = L.mock_class(p)
        dt = {v: k for k, v in CLS.items()}[cls]
        if L.mock_is_sparse(p):
            nnz = L.mock_nnz(p)
            jc = np.ctypeslib.as_array(C.cast(L.mxGetJc(p), C.POINTER(C.c_size_t)), (n + 1,)).astype(np.int64)
            ir = np.ctypeslib.as_array(C.cast(L.mxGetIr(p), C.POINTER(C.c_size_t)), (max(nnz, 1),))[:nnz].astype(np.int64)
            pr = np.ctypeslib.as_array(C.cast(L.mxGetData(p), C.POINTER(C.c_double)), (max(nnz, 1),))[:nnz].copy()
            return sp.csc_matrix((pr, ir, jc), shape=(m, n))
        if m * n == 0:
            return np.zeros((m, n), dtype=dt)
        buf = (C.c_char * (m * n * dt.itemsize)).from_address(L.mxGetData(p))
        return np.frombuffer(buf, dtype=dt).reshape((m, n), order="F").copy()

    def __call__(self, cmd, *args, nout=0):
        pin = [self._to_mx(cmd)] + [self._to_mx(a) for a in args]
        arr_in = (C.c_void_p * len(pin))(*pin)
        arr_out = (C.c_void_p * max(nout, 1))()
        err = C.create_string_buffer(2048)
        rc = self.L.mock_call(nout, arr_out, len(pin), arr_in, err, 2048)
        for p in pin:
            self.L.mock_free(p)
        if rc:
            raise RuntimeError("cnmfe_mex('%s'): %s" % (cmd, err.value.decode()))
        outs = []
        for i in range(max(nout, 1)):
            if arr_out[i]:
                if i < nout:
                    outs.append(self._from_mx(arr_out[i]))
                self.L.mock_free(arr_out[i])
        return outs[0] if nout == 1 else outs


class _Geometry:
    def create_patch(self, *a):
        pass


def _case():
    from cnmf_e_amd import synth
    d1, d2, T, K, r = 48, 40, 160, 5, 5
    f = synth.make_factors(d1, d2, T, K, 3, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)                                  # (T, d)
    return d1, d2, T, K, r, f, Y


def _python_iteration(d1, d2, T, r, f, Y, pdims):
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    eng = Engine(0)
    try:
        video = PatchedVideo(d1, d2, T, pdims, r, eng)
        video.upload_from_full(Y)
        s = Sources2D(video, Options(ring_radius=r, maxIter=3), f.A_init, f.C_init, f.sn)
        s.update_background_parallel()
        s.update_spatial_parallel()
        A = s.A.toarray().astype(np.float64)
        s.update_temporal_parallel()
        return A, np.asarray(s.C, dtype=np.float64).copy()
    finally:
        eng.close()


def test_the_gateway_runs_the_twins_sequence_and_agrees_with_the_python_host():
    from cnmf_e_amd.sources2d import PatchedVideo, determine_search_location
    d1, d2, T, K, r, f, Y = _case()
    pdims = [24, 20]
    A_ref, C_ref = _python_iteration(d1, d2, T, r, f, Y, pdims)
    geo = PatchedVideo(d1, d2, T, pdims, r, _Geometry())
    assert len(geo.order) == 4
    mex = Mex()
    h = float(mex("create", 0, nout=1)[0, 0])
    try:
        mex("set_option", h, "prealloc", 0)
        mex("set_option", h, "lanes", 3)                                 # cnmfe_handle.m: several patches per context -> execution lanes (the Python reference run uses one)
        with pytest.raises(RuntimeError, match="unknown option"):
            mex("set_option", h, "no_such_option", 1)
        pid = {idx: float(i + 1) for i, idx in enumerate(geo.order)}
        for idx in geo.order:                                            # cnmfe_handle.m: patches, blocks of the video (uint16 for one of them: widened on the device), rings
            mex("patch", h, pid[idx], geo.patch_pos[idx].astype(np.float64), geo.block_pos[idx].astype(np.float64), d1, d2, T)
            blk = np.ascontiguousarray(Y[:, geo.block_pix[idx]].T)       # d_b x T
            mex("upload", h, pid[idx], blk, 0)
            mex("ring_init", h, pid[idx], r, np.zeros((0, 0)))
        A = sp.csc_matrix(f.A_init, dtype=np.float64)
        Cm = np.asarray(f.C_init, dtype=np.float64)
        # ---- update_background_parallel.m ----
        mex("bind_traces", h, Cm)
        assert bool(mex("first_run", h, pid[geo.order[0]], nout=1)[0, 0])
        for idx in geo.order:
            Ab = A[geo.block_pix[idx]]
            ind = np.nonzero(np.asarray(abs(Ab).sum(axis=0)).ravel() > 0)[0]
            info = mex("fit_ring", h, pid[idx], Ab[:, ind], (ind + 1).astype(np.int32), 1, nout=1)
            assert info.shape == (1, 4) and info[0, 0] == 1              # first run
        A_prev, C_prev = A, Cm
        # ---- update_spatial_parallel.m: queued updates, collected two patches late ----
        IND = sp.csc_matrix(determine_search_location(A, d1, d2, 3.0, 8.0, 3.0), dtype=bool)
        rows, cols, vals = [], [], []
        pending = []

        def collect(q, pp, ind):
            An = mex("spatial_collect", h, q, nout=1).tocoo()
            rows.append(pp[An.row]); cols.append(ind[An.col]); vals.append(An.data)

        for idx in geo.order:
            pp, bp = geo.patch_pix[idx], geo.block_pix[idx]
            ind = np.nonzero(np.asarray(IND[pp].sum(axis=0)).ravel() > 0)[0]
            if ind.size == 0:
                continue
            halo = geo.halo_pix(idx)
            indp = np.nonzero(np.asarray(abs(A_prev[halo]).sum(axis=0)).ravel() > 0)[0] if halo.size else np.zeros(0, dtype=np.int64)
            mex("residual", h, pid[idx], A_prev[bp][:, indp], (indp + 1).astype(np.int32))
            q = float(mex("spatial_queue", h, pid[idx], "hals", A[pp][:, ind], (ind + 1).astype(np.int32), IND[pp][:, ind], f.sn[pp].astype(np.float64), 3, nout=1)[0, 0])
            pending.append((q, pp, ind))
            while len(pending) > 2:
                collect(*pending.pop(0))
        while pending:
            collect(*pending.pop(0))
        mex("synchronize", h)
        A_new = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(d1 * d2, K))
        A_new.eliminate_zeros(); A_new.sort_indices()
        keep = mex("postprocess", h, A_new, d1, d2, nout=1).ravel().astype(bool)
        coo = A_new.tocoo()                                              # (find(A_new): column-major order = the order of the CSC entries)
        order = np.lexsort((coo.row, coo.col))
        A_pp = sp.csc_matrix((coo.data[order][keep], (coo.row[order][keep], coo.col[order][keep])), shape=A_new.shape)
        assert rel(A_pp.toarray(), A_ref) <= 2e-6, rel(A_pp.toarray(), A_ref)
        # ---- update_temporal_parallel.m: jobs, one sweep over all of them ----
        A = A_pp
        mex("bind_traces", h, C_prev)
        mex("stitch_begin", h, K, T)
        jobs = []
        for idx in geo.order:
            pp, bp = geo.patch_pix[idx], geo.block_pix[idx]
            ind = np.nonzero(np.asarray(abs(A[bp]).sum(axis=0)).ravel() > 0)[0]
            if ind.size == 0:
                continue
            indp = np.nonzero(np.asarray(abs(A_prev[bp]).sum(axis=0)).ravel() > 0)[0]
            mex("residual", h, pid[idx], A_prev[bp][:, indp], (indp + 1).astype(np.int32))
            job = float(mex("temporal_job", h, pid[idx], A[pp][:, ind], (ind + 1).astype(np.int32), 3, nout=1)[0, 0])
            jobs.append((job, ind))
        mex("temporal_jobs_sweep", h)
        for job, ind in jobs:
            mex("stitch_add_job", h, job, (ind + 1).astype(np.float64))
        mex("stitch_finish_async", h, 1, K, T)
        C_raw = mex("stitch_collect", h, nout=1)
        assert C_raw.dtype == np.float32 and C_raw.shape == (K, T)
        assert rel(C_raw, C_ref) <= 2e-6, rel(C_raw, C_ref)
        with pytest.raises(RuntimeError, match="no stitch_finish_async outstanding"):
            mex("stitch_collect", h, nout=1)
        with pytest.raises(RuntimeError, match="no such queued update"):
            mex("spatial_collect", h, 1, nout=1)
    finally:
        mex("destroy", h)
