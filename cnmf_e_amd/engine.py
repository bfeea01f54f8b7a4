"""Thin numpy-facing wrapper over the C ABI (one Engine == one cnmfe_ctx == one GPU).

Mirrors the reference's per-patch kernel functions (SURVEY.md section 8(b) level 2):
  fit_ring_model, the residual expression, HALS_spatial(_thresh), nnls_spatial,
  HALS_temporal, post_process_spatial.
All arrays cross the boundary as plain pointers; sparse matrices are scipy CSC
(MATLAB's layout).  Trace matrices are numpy (K, T) C-order == CNMFE_ROWMAJOR.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _lib as L

_DT = {np.dtype(np.float32): L.F32, np.dtype(np.float64): L.F64, np.dtype(np.uint16): L.U16,
       np.dtype(np.uint8): L.U8, np.dtype(np.float16): L.F16}


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None and a.size else C.cast(None, t)


def _csc(A, nrow):
    """-> (K, colptr int64, rowidx int32, val float32) with sorted, de-duplicated columns."""
    A = sp.csc_matrix(A)
    if A.shape[0] != nrow:
        raise ValueError("sparse matrix has %d rows, expected %d" % (A.shape[0], nrow))
    if not A.has_canonical_format:
        A = A.copy(); A.sum_duplicates(); A.sort_indices()
    return (A.shape[1], np.ascontiguousarray(A.indptr, dtype=np.int64),
            np.ascontiguousarray(A.indices, dtype=np.int32), np.ascontiguousarray(A.data, dtype=np.float32))


def _traces(Cm, K, T):
    Cm = np.ascontiguousarray(Cm, dtype=np.float32)
    if Cm.shape != (K, T):
        raise ValueError("trace matrix has shape %s, expected %s" % (Cm.shape, (K, T)))
    return Cm


class BoundRows:
    """C(ind, :) of a trace matrix, kept as (matrix, row indices): when the matrix is the one bound on the engine the rows are gathered
    on the device (CNMFE_BOUND_ROWS); anywhere else it behaves like the array C[ind] (np.asarray)"""
    def __init__(self, parent, ind):
        self.parent = parent
        self.ind = np.ascontiguousarray(ind, dtype=np.int32)
        self.shape = (int(self.ind.size), int(parent.shape[1]))
        self.dtype = parent.dtype
        self.ndim = 2
    def __array__(self, dtype=None, copy=None):
        a = self.parent[self.ind]
        return a if dtype is None else a.astype(dtype, copy=False)
    def __len__(self):
        return self.shape[0]
    def __getitem__(self, key):
        return np.asarray(self)[key]
    def mean(self, *a, **k):
        return np.asarray(self).mean(*a, **k)


class DeviceTraces:
    """A K x T fp32 trace matrix that lives in a torch DEVICE tensor (what the sharded temporal update's all-reduce produces).  It is bound on
    the engine device-to-device, row subsets go through BoundRows, row means are taken on the device; a host copy is only made when
    somebody actually reads the values (np.asarray), and then kept."""
    def __init__(self, tensor):
        self.tensor = tensor.contiguous()
        self.shape = tuple(int(x) for x in tensor.shape)
        self.dtype = np.dtype(np.float32)
        self.ndim = 2
        self.flags = {"C_CONTIGUOUS": True}
        self._host = None
    def host(self):
        if self._host is None:
            self._host = self.tensor.cpu().numpy()
        return self._host
    def __array__(self, dtype=None, copy=None):
        a = self.host()
        return a if dtype is None else a.astype(dtype, copy=False)
    def __getitem__(self, key):
        return self.host()[key]
    def __len__(self):
        return self.shape[0]
    def mean(self, axis=None, dtype=None, **k):
        import torch
        if axis == 1 and self._host is None:
            return self.tensor.to(torch.float64).mean(dim=1).cpu().numpy().astype(dtype or np.float64)
        return self.host().mean(axis=axis, dtype=dtype, **k)
    def copy(self):
        return self.host().copy()
    def data_ptr(self):
        return self.tensor.data_ptr()


class BoundOnly:
    """identity of a trace matrix that exists only as the engine's bound device matrix (stitch_finish(want="bound")): it can be passed where the bound
    matrix is meant; it has no host values"""
    def __init__(self, K, T):
        self.shape = (int(K), int(T)); self.dtype = np.dtype(np.float32); self.ndim = 2
        self.flags = {"C_CONTIGUOUS": True}
    def __array__(self, dtype=None, copy=None):
        raise RuntimeError("this trace matrix was left on the device only (stitch_finish(want='bound')): deconvolve or download it through the engine")


class _PinnedBlock:
    """Owner of one pinned host buffer of a LazyHostTraces.  The ctypes array every ndarray view of the buffer is built on holds the only strong
    reference to this object, so the block goes back to the engine's pool -- or is freed, once the engine is closed or gone -- when the LAST view
    dies: an array taken from `s.C` (np.asarray, a slice, .T) stays valid for as long as somebody holds it, whatever happens to `s` or the engine."""
    def __init__(self, eng, nbytes):
        import weakref
        self.eng = weakref.ref(eng); self.nbytes = nbytes; self.ready = False
        self.gen = None                                   # the download batch that fills the buffer (Engine._mark_batch): what wait() waits for
        self.ptr = eng._pinned_take(nbytes)
    def _wait_ctx(self, ctx, check):
        # this block's batch only: the copy stream may already hold the NEXT iteration's downloads, which a release of this buffer must not wait for
        rc = L.lib.cnmfe_stitch_wait(ctx) if self.gen is None else L.lib.cnmfe_copy_wait(ctx, self.gen)
        if check:
            L.check(rc)
    def wait(self):
        if not self.ready:
            eng = self.eng()
            if eng is not None and getattr(eng, "_ctx", None):
                self._wait_ctx(eng._ctx, True)                   # cnmfe_destroy drains the copy stream itself
            self.ready = True
    def __del__(self):
        try:
            eng = self.eng()
            if eng is not None and getattr(eng, "_ctx", None):
                if not self.ready:
                    self._wait_ctx(eng._ctx, False)              # the copy may still be writing into the buffer
                eng._pinned_give(self.ptr, self.nbytes)
            else:
                L.lib.cnmfe_host_free(self.ptr)
        except Exception:
            pass


class LazyHostTraces:
    """The K x T result of the temporal update as the host sees it: the engine keeps the matrix bound on the device (that is what the next
    background / spatial / temporal calls read) and streams a copy into pinned host memory on a second stream; the first time somebody
    READS the values (np.asarray, indexing, mean) this object waits for that copy -- never for the compute stream.  Identity of the bound
    matrix like any array returned by stitch_finish.  Arrays handed out are views of the pinned buffer and keep it alive (_PinnedBlock);
    np.array(x) / x.copy() / x.astype(...) are copies as for an ndarray."""
    def __init__(self, eng, K, T):
        self._eng = eng
        self.shape = (int(K), int(T)); self.dtype = np.dtype(np.float32); self.ndim = 2
        self.flags = {"C_CONTIGUOUS": True}
        n = max(1, K * T)
        blk = _PinnedBlock(eng, n * 4)
        buf = (C.c_float * n).from_address(blk.ptr)
        buf._owner = blk                                  # buf <- memoryview <- ndarray (and every view of it): the block lives as long as they do
        self._ptr = blk.ptr
        self._blk = __import__("weakref").ref(blk)
        self._arr = np.frombuffer(buf, dtype=np.float32, count=K * T).reshape(K, T)
    @property
    def _ready(self):
        return self._blk().ready
    def host(self):
        self._blk().wait()
        return self._arr
    def __array__(self, dtype=None, copy=None):
        a = self.host()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            if copy is False:
                raise ValueError("a float32 trace matrix cannot be viewed as %s without a copy" % np.dtype(dtype))
            return a.astype(dtype)
        return a.copy() if copy else a
    def __getitem__(self, key):
        return self.host()[key]
    def __len__(self):
        return self.shape[0]
    def mean(self, *a, **k):
        return self.host().mean(*a, **k)
    def copy(self):
        return self.host().copy()
    def astype(self, *a, **k):
        return self.host().astype(*a, **k)
    def __getattr__(self, name):                      # everything else an ndarray has (.T, .sum, .max, ...): the host copy's
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.host(), name)


for _op in ("add", "sub", "mul", "truediv", "radd", "rsub", "rmul", "rtruediv", "lt", "le", "gt", "ge", "eq", "ne", "neg", "abs", "matmul", "rmatmul"):
    setattr(LazyHostTraces, "__%s__" % _op, (lambda op: lambda self, *a: getattr(self.host(), "__%s__" % op)(*a))(_op))
LazyHostTraces.__hash__ = object.__hash__


class Engine:
    supports_lazy_traces = True
    # pinned buffers of LazyHostTraces, recycled by size (page-locking 20 MB costs milliseconds)
    def _pinned_take(self, nbytes):
        pool = self.__dict__.setdefault("_pinned_pool", {})
        lst = pool.get(nbytes)
        if lst:
            return lst.pop()
        p = L.lib.cnmfe_host_alloc(nbytes)
        if not p:
            raise L.CnmfeError(L.lib.cnmfe_last_error().decode("utf-8", "replace"))
        return p
    def _pinned_give(self, ptr, nbytes):
        pool = self.__dict__.get("_pinned_pool")
        if pool is None or getattr(self, "_ctx", None) is None:
            L.lib.cnmfe_host_free(ptr)
        else:
            pool.setdefault(nbytes, []).append(ptr)

    # -- bound traces: obj.C is the same matrix for several calls of one iteration; bind_traces uploads it once and every
    # call that is handed THAT array object afterwards passes (NULL, CNMFE_BOUND).  The array must not be mutated in place
    # while it is bound (Sources2D replaces C, it never writes into it).
    def bind_traces(self, Cm, device_ptr=None):
        """device_ptr: address of a row-major fp32 DEVICE copy of Cm (e.g. the tensor an all-reduce just produced): the engine then binds
        with a device-to-device copy; Cm stays the host-side identity of the bound matrix"""
        Cm = None if Cm is None or Cm.shape[0] == 0 else Cm
        if isinstance(Cm, DeviceTraces):
            device_ptr = Cm.data_ptr()
        if Cm is None or Cm.dtype != np.float32 or not Cm.flags["C_CONTIGUOUS"]:
            self._bound = None
            L.check(L.lib.cnmfe_traces_bind(self._ctx, 0, 0, None, L.ROWMAJOR))
            return
        src = _p(Cm, L.f32p) if device_ptr is None else C.cast(int(device_ptr), L.f32p)
        L.check(L.lib.cnmfe_traces_bind(self._ctx, Cm.shape[0], Cm.shape[1], src, L.ROWMAJOR))
        self._bound = Cm

    def _targs(self, Cm, K, T):
        """(pointer, c_order, keep-alive) for a K x T trace argument"""
        if K == 0 or Cm is None:
            return None, L.ROWMAJOR, None
        if Cm is getattr(self, "_bound", None) and Cm.shape == (K, T):
            return None, L.BOUND, None
        if isinstance(Cm, BoundRows) and Cm.parent is getattr(self, "_bound", None) and Cm.shape == (K, T):
            return C.cast(Cm.ind.ctypes.data, L.f32p), L.BOUND_ROWS, Cm.ind
        a = _traces(Cm, K, T)
        return _p(a, L.f32p), L.ROWMAJOR, a

    def __init__(self, device: int = 0):
        self._opts_set = {}
        self._ctx = L.lib.cnmfe_create(int(device))
        if not self._ctx:
            raise L.CnmfeError("cnmfe_create failed: " + L.lib.cnmfe_last_error().decode())
        self.device = device
        self._patch = {}

    def close(self):
        if getattr(self, "_ctx", None):
            L.lib.cnmfe_destroy(self._ctx)
            self._ctx = None
            for lst in self.__dict__.pop("_pinned_pool", {}).values():
                for ptr in lst:
                    L.lib.cnmfe_host_free(ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data plane ---------------------------------------------------------
    def create_patch(self, pid, patch_rect, block_rect, d1, d2, T):
        pr = np.asarray(patch_rect, dtype=np.int32); br = np.asarray(block_rect, dtype=np.int32)
        L.check(L.lib.cnmfe_patch_create(self._ctx, pid, _p(pr, L.i32p), _p(br, L.i32p), d1, d2, T))
        nr, nc = int(pr[1] - pr[0] + 1), int(pr[3] - pr[2] + 1)
        nrb, ncb = int(br[1] - br[0] + 1), int(br[3] - br[2] + 1)
        self._patch[pid] = dict(d=nr * nc, d_b=nrb * ncb, T=int(T), nr=nr, nc=nc, nr_b=nrb, nc_b=ncb)

    def upload_block(self, pid, Y, t0=0):
        """Y: (nt, d_b) host array (float32/float64/uint16/uint8/float16), frames t0..t0+nt."""
        Y = np.ascontiguousarray(Y)
        if Y.dtype not in _DT:
            # element types the ABI has no code for (int16 TIFF / HDF5 recordings, int32, bool): widened on the host to the narrowest float that holds
            # them exactly -- the device stores fp32 either way
            if Y.dtype.kind not in "iub":
                raise TypeError("cannot upload a block of %s" % Y.dtype)
            Y = Y.astype(np.float32 if Y.dtype.itemsize <= 2 else np.float64)
        info = self._patch[pid]
        if Y.ndim != 2 or Y.shape[1] != info["d_b"]:
            raise ValueError("block must be (frames, %d), got %s" % (info["d_b"], Y.shape))
        L.check(L.lib.cnmfe_upload_block(self._ctx, pid, Y.ctypes.data_as(C.c_void_p), _DT[Y.dtype], L.HOST, t0, Y.shape[0]))

    def upload_block_device(self, pid, dev_ptr, nt, t0=0, dtype=L.F32):
        """dev_ptr: raw device address of an (nt, d_b) array already in HBM (e.g. torch tensor .data_ptr())."""
        L.check(L.lib.cnmfe_upload_block(self._ctx, pid, C.c_void_p(int(dev_ptr)), dtype, L.DEVICE, t0, nt))

    def ymean(self, pid):
        out = np.empty(self._patch[pid]["d_b"], dtype=np.float64)
        L.check(L.lib.cnmfe_get_ymean(self._ctx, pid, _p(out, L.f64p)))
        return out

    # ---- ring -----------------------------------------------------------------
    def ring_init(self, pid, radius, num_neighbors=None):
        L.check(L.lib.cnmfe_ring_init(self._ctx, pid, int(radius), int(num_neighbors or 0)))

    def fit_reserve(self, pid):
        """allocate the ring fit's large device buffers of this patch now (cnmfe_fit_reserve): the first fit then queues its kernels without a hipMalloc"""
        L.check(L.lib.cnmfe_fit_reserve(self._ctx, pid))

    def ring_csr(self, pid):
        nnz = C.c_int64(); p = C.c_int32()
        L.check(L.lib.cnmfe_ring_nnz(self._ctx, pid, C.byref(nnz), C.byref(p)))
        info = self._patch[pid]
        rowptr = np.empty(info["d"] + 1, dtype=np.int64)
        col = np.empty(nnz.value, dtype=np.int32); val = np.empty(nnz.value, dtype=np.float32)
        L.check(L.lib.cnmfe_ring_get_csr(self._ctx, pid, _p(rowptr, L.i64p), _p(col, L.i32p), _p(val, L.f32p)))
        return sp.csr_matrix((val, col, rowptr), shape=(info["d"], info["d_b"]))

    def ring_first_run(self, pid):
        f = C.c_int()
        L.check(L.lib.cnmfe_ring_first_run(self._ctx, pid, C.byref(f)))
        return bool(f.value)

    def ring_set_values(self, pid, val):
        val = np.ascontiguousarray(val, dtype=np.float32)
        L.check(L.lib.cnmfe_ring_set_values(self._ctx, pid, _p(val, L.f32p)))

    def b0(self, pid):
        out = np.empty(self._patch[pid]["d"], dtype=np.float32)
        L.check(L.lib.cnmfe_b0_get(self._ctx, pid, _p(out, L.f32p)))
        return out

    def set_b0(self, pid, b0):
        b0 = np.ascontiguousarray(b0, dtype=np.float32)
        L.check(L.lib.cnmfe_b0_set(self._ctx, pid, _p(b0, L.f32p)))

    def estimate_noise(self, pid, nframes=None):
        """GetSn of the first `nframes` (default min(T, 3000), Sources2D.m:333-335) frames of the raw video, per block pixel"""
        info = self._patch[pid]
        n = min(info["T"], 3000) if nframes is None else int(nframes)
        out = np.empty(info["d_b"], dtype=np.float32)
        L.check(L.lib.cnmfe_estimate_noise(self._ctx, pid, n, _p(out, L.f32p)))
        return out

    def set_noise(self, pid, sn_block):
        """sn of the block pixels (update_background_parallel.m:131,137); read by the outlier branch of fit_ring_model only"""
        sn_block = np.ascontiguousarray(sn_block, dtype=np.float32).ravel()
        assert sn_block.size == self._patch[pid]["d_b"], (sn_block.size, self._patch[pid]["d_b"])
        L.check(L.lib.cnmfe_set_noise(self._ctx, pid, _p(sn_block, L.f32p)))

    # ---- kernels ----------------------------------------------------------------
    def fit_ring_model(self, pid, A_block, C_block, thresh_outlier=float("nan"), with_projection=True, want_b0=True):
        """[W, b0] = fit_ring_model(Y, A, C, W_old, thresh_outlier, sn, ind_patch, with_projection); W stays resident."""
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_block, info["d_b"]) if A_block is not None else (0, None, None, None)
        cptr, cord, _keep = self._targs(C_block, K, info["T"])
        b0 = np.empty(info["d"], dtype=np.float32) if want_b0 else None
        inf = np.zeros(4, dtype=np.int64)
        L.check(L.lib.cnmfe_fit_ring_model(self._ctx, pid, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr,
                                           cord, float(thresh_outlier), int(bool(with_projection)), _p(b0, L.f32p), _p(inf, L.i64p)))
        return b0, dict(first_run=bool(inf[0]), frame_stride=int(inf[1]), n_active=int(inf[2]), pmax=int(inf[3]))

    def ring_solve_stats(self, pid):
        """diagnostics of the last fit's ring solve out of the cached inverses (cnmfe_ring_solve_stats): pixels left to the factorising kernel, ridge-series terms,
        pixels with at least one term, inverses rebuilt; all -1 without inverses"""
        out = np.zeros(4, dtype=np.int64)
        L.check(L.lib.cnmfe_ring_solve_stats(self._ctx, pid, _p(out, L.i64p)))
        return dict(left_over=int(out[0]), series_terms=int(out[1]), pixels_with_terms=int(out[2]), rebuilt=int(out[3]))

    # -- bg_ssub > 1 ---------------------------------------------------------------------------------------------
    def patch_derive(self, src_pid, new_pid, ssub, mode):
        """low-resolution patch of src_pid (mode 'nearest' | 'bicubic'); its FOV is the ceil(nr_b/s) x ceil(nc_b/s) grid"""
        src = self._patch[src_pid]
        L.check(L.lib.cnmfe_patch_derive(self._ctx, src_pid, new_pid, int(ssub), 0 if mode == "nearest" else 1))
        d1s, d2s = -(-src["nr_b"] // ssub), -(-src["nc_b"] // ssub)
        self._patch[new_pid] = dict(d=d1s * d2s, d_b=d1s * d2s, T=src["T"], nr=d1s, nc=d2s, nr_b=d1s, nc_b=d2s)

    def fit_ring_model_ssub(self, pid, fit_pid, res_pid, ssub, A_block, C_block, thresh_outlier=float("nan"), with_projection=True):
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_block, info["d_b"]) if A_block is not None else (0, None, None, None)
        cptr, cord, _keep = self._targs(C_block, K, info["T"])
        inf = np.zeros(4, dtype=np.int64)
        L.check(L.lib.cnmfe_fit_ring_model_ssub(self._ctx, pid, fit_pid, res_pid, int(ssub), K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p),
                                                cptr, cord, float(thresh_outlier), int(bool(with_projection)), _p(inf, L.i64p)))
        return None, dict(first_run=bool(inf[0]), frame_stride=int(inf[1]), n_active=int(inf[2]), pmax=int(inf[3]))

    def residual_ssub(self, pid, res_pid, ssub, A_prev_block=None, C_prev=None, want=False):
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_prev_block, info["d_b"]) if A_prev_block is not None and A_prev_block.shape[1] else (0, None, None, None)
        cptr, cord, _keep = self._targs(C_prev, K, info["T"])
        out = np.empty((info["T"], info["d"]), dtype=np.float32) if want else None
        dst = out.ctypes.data_as(C.c_void_p) if want else C.c_void_p(None)
        L.check(L.lib.cnmfe_residual_ssub(self._ctx, pid, res_pid, int(ssub), K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr,
                                          cord, dst, L.HOST))
        return out

    def residual(self, pid, A_prev_block=None, C_prev=None, want=False, out_dev_ptr=None):
        """want=True returns Ysig as a (T, d) host array; out_dev_ptr: raw device address of a (T, d) fp32 buffer that
        receives a device-to-device copy instead."""
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_prev_block, info["d_b"]) if A_prev_block is not None and A_prev_block.shape[1] else (0, None, None, None)
        cptr, cord, _keep = self._targs(C_prev, K, info["T"])
        out = np.empty((info["T"], info["d"]), dtype=np.float32) if want else None
        if out_dev_ptr is not None:
            dst, space = C.c_void_p(int(out_dev_ptr)), L.DEVICE
        else:
            dst, space = (out.ctypes.data_as(C.c_void_p) if want else C.c_void_p(None)), L.HOST
        L.check(L.lib.cnmfe_residual(self._ctx, pid, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr, cord, dst, space))
        return out

    def get_sn(self, pid):
        """sn = GetSn(Ysig) per patch pixel (update_spatial_parallel.m:191-194); needs residual() first"""
        out = np.empty(self._patch[pid]["d"], dtype=np.float32)
        L.check(L.lib.cnmfe_get_sn(self._ctx, pid, _p(out, L.f32p)))
        return out

    def update_spatial(self, pid, algorithm, A_patch, C_patch, IND_patch, sn=None, param=3, defer=False):
        """Returns the updated A as a CSC matrix with exactly IND's pattern (explicit zeros kept).  defer=True returns a callable instead that
        fetches it: the sweeps are queued, the caller does other host work under them and calls it afterwards (before the next spatial update)."""
        info = self._patch[pid]
        alg = {"hals": L.SPATIAL_HALS, "hals_thresh": L.SPATIAL_HALS_THRESH, "nnls": L.SPATIAL_NNLS}[algorithm]
        K, cp, ri, va = _csc(A_patch, info["d"])
        IND = sp.csc_matrix(IND_patch).astype(np.float32)
        IND.sort_indices()
        K2, icp, iri, _ = _csc(IND, info["d"])
        if K2 != K:
            raise ValueError("A and IND disagree on K")
        cptr, cord, _keep = self._targs(C_patch, K, info["T"])
        snf = np.ascontiguousarray(sn, dtype=np.float32).ravel() if sn is not None else None
        out = np.zeros(icp[-1], dtype=np.float32)
        L.check(L.lib.cnmfe_update_spatial(self._ctx, pid, alg, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr, cord,
                                           _p(icp, L.i64p), _p(iri, L.i32p), _p(snf, L.f32p), int(param), None if defer else _p(out, L.f32p)))
        if defer:
            pend = {}
            def start():
                """queue the download into pinned memory right behind the sweeps (cnmfe_update_spatial_fetch_async): fetch() then waits for that point of
                the stream only, so the caller may queue the next patch's kernels first"""
                if out.size == 0 or "t" in pend:
                    return
                nb = 1 << max(12, int(out.size * 4 - 1).bit_length())          # pooled by size class
                ptr = self._pinned_take(nb)
                tk = C.c_int64(0)
                try:
                    L.check(L.lib.cnmfe_update_spatial_fetch_async(self._ctx, ptr, int(out.size), C.byref(tk)))
                except Exception:
                    self._pinned_give(ptr, nb)
                    raise
                pend["t"] = (tk.value, ptr, nb)
            def compacted(vals, keep=None):
                """CSC of the non-zero values (keep: and flagged entries) on IND's pattern, rows sorted: one pass in the library's host helper --
                scipy's eliminate_zeros() on the 250 k-entry mask pattern was the longest host step between the spatial and the temporal update"""
                optr = np.empty(K + 1, dtype=np.int64); orow = np.empty(max(1, vals.size), dtype=np.int32); oval = np.empty(max(1, vals.size), dtype=np.float32)
                n = C.c_int64(0)
                L.check(L.lib.cnmfe_csc_drop_zeros(K, icp.ctypes.data, iri.ctypes.data, vals.ctypes.data, None if keep is None else keep.ctypes.data,
                                                   optr.ctypes.data, orow.ctypes.data, oval.ctypes.data, C.byref(n)))
                M = sp.csc_matrix((oval[:n.value], orow[:n.value], optr), shape=(info["d"], K))
                M.has_canonical_format = True                            # (sorted rows, no duplicates: the mask's pattern was canonical)
                return M
            def start_connected(connected_fov):
                """queue the connectivity constraint + the downloads of (values, keep flags) into pinned memory now; fetch(connected_fov=...) then only waits for them"""
                if out.size == 0 or "c" in pend or "t" in pend:
                    return
                nb = 1 << max(12, int(out.size * 5 - 1).bit_length())
                ptr = self._pinned_take(nb)
                tk = C.c_int64(0)
                try:
                    L.check(L.lib.cnmfe_update_spatial_fetch_connected_async(self._ctx, int(connected_fov[0]), int(connected_fov[1]), K, _p(icp, L.i64p), _p(iri, L.i32p),
                                                                             ptr, ptr + out.size * 4, C.byref(tk)))
                except Exception:
                    self._pinned_give(ptr, nb)
                    raise
                pend["c"] = (tk.value, ptr, nb)
            def fetch(connected_fov=None, compact=False):
                """connected_fov = (d1, d2): the patch is the whole field of view -- also apply the connectivity constraint on the device and
                return (A_raw, A) instead of A_raw.  compact: without the stored zeros of the mask pattern (rows sorted); with connected_fov, A_raw then comes as a
                callable that builds it on first use"""
                if connected_fov is None:
                    if "t" in pend:
                        tk, ptr, nb = pend.pop("t")
                        try:
                            L.check(L.lib.cnmfe_ticket_wait(self._ctx, tk))
                            C.memmove(out.ctypes.data, ptr, out.size * 4)
                        finally:
                            self._pinned_give(ptr, nb)
                    else:
                        L.check(L.lib.cnmfe_update_spatial_fetch(self._ctx, _p(out, L.f32p), int(out.size)))
                    return compacted(out) if compact else sp.csc_matrix((out, iri.copy(), icp.copy()), shape=(info["d"], K))
                keep = np.zeros(out.size, dtype=np.uint8)
                if "c" in pend:
                    tk, ptr, nb = pend.pop("c")
                    try:
                        L.check(L.lib.cnmfe_ticket_wait(self._ctx, tk))
                        C.memmove(out.ctypes.data, ptr, out.size * 4)
                        C.memmove(keep.ctypes.data, ptr + out.size * 4, out.size)
                    finally:
                        self._pinned_give(ptr, nb)
                else:
                    L.check(L.lib.cnmfe_update_spatial_fetch_connected(self._ctx, int(connected_fov[0]), int(connected_fov[1]), K, _p(icp, L.i64p), _p(iri, L.i32p),
                                                                       _p(out, L.f32p), _p(keep, L.u8p)))
                if compact:                                  # the raw update (obj.A before post-processing) is compacted when somebody reads it: nothing in the iteration does
                    return (lambda: compacted(out)), compacted(out, keep)
                A_raw = sp.csc_matrix((out, iri.copy(), icp.copy()), shape=(info["d"], K))
                A_pp = sp.csc_matrix((out * keep, iri.copy(), icp.copy()), shape=(info["d"], K))
                return A_raw, A_pp
            fetch.start = start
            fetch.start_connected = start_connected
            return fetch
        return sp.csc_matrix((out, iri.copy(), icp.copy()), shape=(info["d"], K))

    def hals_temporal(self, pid, A_patch, C_patch, maxIter=5, want_C=True, want_raw=True):
        """[C, C_raw] = HALS_temporal(Ysig, A, C, maxIter); want_C=False skips the download of C (the caller of
        update_temporal_parallel.m:180 only keeps C_raw), want_raw=False that of C_raw too: it stays on the device for stitch_add."""
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_patch, info["d"])
        T = info["T"]
        cptr, cord, _keep = self._targs(C_patch, K, T)
        Cout = np.empty((K, T), dtype=np.float32) if want_C else None
        Craw = np.empty((K, T), dtype=np.float32) if want_raw else None
        aa = np.empty(K, dtype=np.float32)
        L.check(L.lib.cnmfe_hals_temporal(self._ctx, pid, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr, cord,
                                          int(maxIter), _p(Cout, L.f32p), _p(Craw, L.f32p), _p(aa, L.f32p)))
        return Cout, Craw, aa

    def fast_temporal(self, pid, A_patch, want_raw=True):
        """[aa, C_raw] = fast_temporal(Ysig, A) (update_temporal_parallel.m:314-337); returns (C_raw, aa)"""
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_patch, info["d"])
        Craw = np.empty((K, info["T"]), dtype=np.float32) if want_raw else None
        aa = np.empty(K, dtype=np.float32)
        L.check(L.lib.cnmfe_fast_temporal(self._ctx, pid, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), L.ROWMAJOR,
                                          _p(Craw, L.f32p), _p(aa, L.f32p)))
        return Craw, aa

    def reconstruct_background(self, pid, b0_block, b0_new_patch, frame0=0, nframes=None):
        """Ybg of one patch (Sources2D.m:1247-1355), frames [frame0, frame0 + nframes): (nframes, d) array; needs the resident residual of (A_prev, C_prev)"""
        info = self._patch[pid]
        nframes = info["T"] - frame0 if nframes is None else int(nframes)
        bb = np.ascontiguousarray(b0_block, dtype=np.float32).ravel(); bn = np.ascontiguousarray(b0_new_patch, dtype=np.float32).ravel()
        out = np.empty((nframes, info["d"]), dtype=np.float32)
        L.check(L.lib.cnmfe_reconstruct_background(self._ctx, pid, _p(bb, L.f32p), _p(bn, L.f32p), int(frame0), nframes, _p(out, L.f32p), L.HOST))
        return out

    def compute_rss(self, pid, A_patch, C_patch, b0_block, b0_new_patch):
        """RSS of one patch, compute_RSS (Sources2D.m:1358-1510): needs the resident residual of (A_prev, C_prev) on the block"""
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_patch, info["d"]) if A_patch is not None and A_patch.shape[1] else (0, np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32))
        cptr, cord, _keep = self._targs(C_patch, K, info["T"])
        bb = np.ascontiguousarray(b0_block, dtype=np.float32).ravel(); bn = np.ascontiguousarray(b0_new_patch, dtype=np.float32).ravel()
        if bb.size != info["d_b"] or bn.size != info["d"]:
            raise ValueError("b0_block / b0_new have %d / %d entries, expected %d / %d" % (bb.size, bn.size, info["d_b"], info["d"]))
        out = C.c_double(0.0)
        L.check(L.lib.cnmfe_compute_rss(self._ctx, pid, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr, cord, _p(bb, L.f32p), _p(bn, L.f32p), C.byref(out)))
        return float(out.value)

    # -- the same two with bg_ssub > 1 ('nearest' resizes, Sources2D.m:1325-1334,1479-1486) --
    def background_ssub(self, pid, fit_pid, ssub, A_prev_block, C_prev, b0_block):
        """W * imresize(Y_block - b0_block - A_prev*C_prev, 1/s, 'nearest') on the fit patch, kept on the device for the two readers below"""
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_prev_block, info["d_b"]) if A_prev_block is not None and A_prev_block.shape[1] else (0, None, None, None)
        cptr, cord, _keep = self._targs(C_prev, K, info["T"])
        bb = np.ascontiguousarray(b0_block, dtype=np.float32).ravel()
        if bb.size != info["d_b"]:
            raise ValueError("b0_block has %d entries, expected %d" % (bb.size, info["d_b"]))
        L.check(L.lib.cnmfe_background_ssub(self._ctx, pid, fit_pid, int(ssub), K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr, cord, _p(bb, L.f32p)))

    def reconstruct_background_ssub(self, pid, b0_new_patch, frame0=0, nframes=None):
        info = self._patch[pid]
        nframes = info["T"] - frame0 if nframes is None else int(nframes)
        bn = np.ascontiguousarray(b0_new_patch, dtype=np.float32).ravel()
        out = np.empty((nframes, info["d"]), dtype=np.float32)
        L.check(L.lib.cnmfe_reconstruct_background_ssub(self._ctx, pid, _p(bn, L.f32p), int(frame0), nframes, _p(out, L.f32p), L.HOST))
        return out

    def compute_rss_ssub(self, pid, A_patch, C_patch, b0_new_patch):
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_patch, info["d"]) if A_patch is not None and A_patch.shape[1] else (0, None, None, None)
        cptr, cord, _keep = self._targs(C_patch, K, info["T"])
        bn = np.ascontiguousarray(b0_new_patch, dtype=np.float32).ravel()
        out = C.c_double(0.0)
        L.check(L.lib.cnmfe_compute_rss_ssub(self._ctx, pid, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr, cord, _p(bn, L.f32p), C.byref(out)))
        return float(out.value)

    @staticmethod
    def _dopts(deconv_options, maxIter=10):
        """deconv_options struct of demo_large_data_1p.m:38-43 -> cnmfe_deconv_opts"""
        o = dict(type="ar1", method="foopsi", smin=-5.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
        o.update(deconv_options or {})
        if str(o["type"]).lower() != "ar1" or str(o["method"]).lower() != "foopsi":
            raise NotImplementedError("only deconv_options type='ar1', method='foopsi' is built")
        return L.DeconvOpts(1, 1, float(o["smin"]), float(o.get("lambda", 0.0)), float(o["max_tau"]),
                            int(bool(o["optimize_b"])), int(bool(o["optimize_pars"])), int(o.get("maxIter", maxIter)))

    # ---- the overlap-region stitch on the device (update_temporal_parallel.m:264-286; cnmfe_stitch_* in include/cnmfe.h) ----
    def stitch_begin(self, K, T):
        L.check(L.lib.cnmfe_stitch_begin(self._ctx, int(K), int(T)))
        self._stitch_shape = (int(K), int(T))

    # ---- several patches per context: set the patches' temporal updates up, sweep them together, add each to the stitch (cnmfe_hals_temporal_job) ----
    supports_temporal_jobs = True

    def hals_temporal_job(self, pid, A_patch, C_patch, maxIter, deconv_options=None, kernel_pars=None):
        """everything of hals_temporal / hals_temporal_deconv up to the Gauss-Seidel sweeps; returns the job number (for stitch_add_job)"""
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_patch, info["d"])
        cptr, cord, _keep = self._targs(C_patch, K, info["T"])
        opts = pars = None
        if deconv_options is not None:
            opts = self._dopts(deconv_options)
            pars = np.zeros(K, dtype=np.float32) if kernel_pars is None else np.ascontiguousarray(kernel_pars, dtype=np.float32)
        job = C.c_int32(-1)
        L.check(L.lib.cnmfe_hals_temporal_job(self._ctx, pid, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr, cord, int(maxIter),
                                              None if opts is None else C.byref(opts), _p(pars, L.f32p), C.byref(job)))
        return job.value

    def temporal_jobs_sweep(self):
        """level l of EVERY job set up since stitch_begin in one launch (the patches are independent)"""
        L.check(L.lib.cnmfe_temporal_jobs_sweep(self._ctx))

    def stitch_add_job(self, job, ind):
        ind = np.ascontiguousarray(ind, dtype=np.int32)
        L.check(L.lib.cnmfe_stitch_add_job(self._ctx, int(job), ind.size, _p(ind, L.i32p)))

    def stitch_add(self, ind):
        """rows `ind` of the accumulator += aa .* C_raw of the temporal call just made (its result is still on the device)"""
        ind = np.ascontiguousarray(ind, dtype=np.int32)
        L.check(L.lib.cnmfe_stitch_add(self._ctx, ind.size, _p(ind, L.i32p)))

    def stitch_allreduce(self, group):
        """one process per GPU: the all-reduce of the accumulators over the torch.distributed group, in place on the device buffer
        (nccl == RCCL over xGMI).  gloo (CPU test hook with every rank on one device) goes through a host copy."""
        import torch
        import torch.distributed as td
        ptr = L.f32p(); ld = C.c_int64(); sp_ = C.c_void_p()
        nccl = td.get_backend(group) == "nccl"
        K, _ = self._stitch_shape
        if nccl:
            # RCCL: the collective is enqueued with the ENGINE's stream as torch's current stream -- the process group orders its own stream behind and in front
            # of it with events, so neither the additions before it nor the finish after it need the host (two drains of the stream per update until round 3)
            L.check(L.lib.cnmfe_stitch_buffer_stream(self._ctx, C.byref(ptr), C.byref(ld), C.byref(sp_)))
        else:
            L.check(L.lib.cnmfe_stitch_buffer(self._ctx, C.byref(ptr), C.byref(ld)))      # (gloo goes through the host: everything added so far must have landed)
        n = K * ld.value
        if n == 0:
            return

        class _View:                                                    # zero-copy torch view of the engine's accumulator
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (C.cast(ptr, C.c_void_p).value, False), "version": 2}
        # ONE tensor per (address, length), kept: a tensor over a Python object's memory is released through that object's deleter, which needs the GIL -- and the
        # process group's watchdog thread, which drops the last reference of a finished collective's operands, then took the GIL for ~50 ms at a time, every
        # ~100 ms, wherever the main thread happened to be (profiles/r04/forced_collectives.txt: c4 on one rank of RCCL 26 -> 50-58 ms per iteration)
        key = (C.cast(ptr, C.c_void_p).value, n)
        cached = getattr(self, "_stitch_view", None)
        if cached is None or cached[0] != key:
            cached = self._stitch_view = (key, torch.as_tensor(_View(), device="cuda"))
        t = cached[1]
        if nccl:
            ext = torch.cuda.ExternalStream(int(sp_.value), device=torch.device("cuda", torch.cuda.current_device()))
            with torch.cuda.stream(ext):
                # (test hook of the one-rank measurements: honoured on a ONE-rank group only -- with more ranks skipping the reduction would silently return wrong traces)
                if not (os.environ.get("CNMFE_SKIP_STITCH_ALLREDUCE") == "1" and td.get_world_size(group) == 1):
                    td.all_reduce(t, group=group)
            return
        h = t.cpu(); td.all_reduce(h, group=group); t.copy_(h)
        torch.cuda.current_stream().synchronize()                       # the engine continues on its own stream

    def stitch_finish(self, subtract_min, want=True):
        """C_raw = acc ./ aa (aa == 0 -> 1), minus the row minima without deconvolution (:279-286); the result becomes the engine's bound
        trace matrix, and the returned host copy its identity: passing THAT array to later calls costs no upload"""
        K, T = self._stitch_shape
        if want == "lazy" and K > 0:
            out = LazyHostTraces(self, K, T)
            L.check(L.lib.cnmfe_stitch_finish_async(self._ctx, int(bool(subtract_min)), C.cast(out._ptr, L.f32p)))
            self._mark_batch(out)
            self._bound = out
            return out
        if want == "bound" and K > 0:                    # no host copy at all: the caller goes on with the bound matrix (deconv_temporal_bound)
            L.check(L.lib.cnmfe_stitch_finish(self._ctx, int(bool(subtract_min)), None, L.ROWMAJOR))
            self._bound = BoundOnly(K, T)
            return self._bound
        out = np.empty((K, T), dtype=np.float32) if want else None
        L.check(L.lib.cnmfe_stitch_finish(self._ctx, int(bool(subtract_min)), _p(out, L.f32p), L.ROWMAJOR))
        self._bound = out if (want and K > 0) else None
        return out

    def hals_temporal_deconv(self, pid, A_patch, C_patch, maxIter, deconv_options, kernel_pars=None, want_all=True):
        """[C, C_raw, results_deconv] = HALS_temporal(Y, A, C, maxIter, deconv_options): returns
        (C, C_raw, S, sn, kernel_pars, aa).  want_all=False skips the download of C and S (update_temporal_parallel.m:106-110
        only keeps C_raw and aa; two K x T copies over PCIe otherwise)."""
        info = self._patch[pid]
        K, cp, ri, va = _csc(A_patch, info["d"])
        T = info["T"]
        cptr, cord, _keep = self._targs(C_patch, K, T)                   # (the bound matrix or rows of it: no K x T upload, as in hals_temporal)
        Craw = np.empty((K, T), dtype=np.float32) if want_all is not None else None      # want_all=None: nothing but aa comes back (C_raw stays on the device for stitch_add)
        Cout = np.empty((K, T), dtype=np.float32) if want_all else None
        S = np.empty((K, T), dtype=np.float32) if want_all else None
        aa = np.empty(K, dtype=np.float32); sn = np.zeros(K, dtype=np.float32) if want_all is not None else None   # want_all=None: the call returns with the sweeps in flight
        pars = np.zeros(K, dtype=np.float32) if kernel_pars is None else np.ascontiguousarray(kernel_pars, dtype=np.float32).copy()
        opts = self._dopts(deconv_options)
        L.check(L.lib.cnmfe_hals_temporal_deconv(self._ctx, pid, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), cptr, cord,
                                                 int(maxIter), C.byref(opts), _p(pars, L.f32p), _p(Cout, L.f32p), _p(Craw, L.f32p),
                                                 _p(S, L.f32p), _p(sn, L.f32p), _p(aa, L.f32p)))
        if want_all is None:                                 # nothing was copied back (the ABI treats kernel_pars as input only then): no stale copies of the inputs
            return None, None, None, None, None, aa
        return Cout, Craw, S, sn, pars, aa

    def deconv_temporal(self, C_raw, deconv_options, overwrite=False):
        """obj.deconvTemporal(): returns (C, C_raw, S, kernel_pars, sn).  The ABI updates C_raw in place (ck_raw - b); overwrite=True
        lets it do that to the caller's array (when that is a C-contiguous float32 one) instead of to a copy."""
        Craw = np.ascontiguousarray(C_raw, dtype=np.float32)
        if not overwrite or not isinstance(C_raw, np.ndarray) or not np.shares_memory(Craw, C_raw):
            # in-place only on the caller's own plain array when it asked for it; anything else (a DeviceTraces' cached host copy, a view
            # np.ascontiguousarray passed through) gets a private copy, so the write never lands in memory somebody else still reads
            if np.shares_memory(Craw, np.asarray(C_raw)):
                Craw = Craw.copy()
        K, T = Craw.shape
        Cout = np.empty_like(Craw); S = np.empty_like(Craw)
        pars = np.zeros(K, dtype=np.float32); sn = np.zeros(K, dtype=np.float32)
        opts = self._dopts(deconv_options)
        L.check(L.lib.cnmfe_deconv_temporal(self._ctx, K, T, _p(Craw, L.f32p), L.ROWMAJOR, C.byref(opts), _p(Cout, L.f32p), _p(S, L.f32p),
                                            _p(pars, L.f32p), _p(sn, L.f32p)))
        return Cout, Craw, S, pars, sn

    def deconv_temporal_bound(self, deconv_options):
        """obj.deconvTemporal() on the engine's BOUND trace matrix (the C_raw that stitch_finish left on the device): returns (C, C_raw, S, kernel_pars, sn) as
        lazy host views (LazyHostTraces: the values arrive in pinned memory behind the kernels, the first read waits for that copy only); C is the
        engine's bound matrix from here on.  No K x T array crosses PCIe before the call returns."""
        K, T = self._bound.shape
        Cout, Craw, S = LazyHostTraces(self, K, T), LazyHostTraces(self, K, T), LazyHostTraces(self, K, T)
        pars, sn = LazyHostTraces(self, 1, K), LazyHostTraces(self, 1, K)
        opts = self._dopts(deconv_options)
        L.check(L.lib.cnmfe_deconv_temporal_bound(self._ctx, C.byref(opts), C.cast(Cout._ptr, L.f32p), C.cast(Craw._ptr, L.f32p), C.cast(S._ptr, L.f32p),
                                                  C.cast(pars._ptr, L.f32p), C.cast(sn._ptr, L.f32p)))
        self._mark_batch(Cout, Craw, S, pars, sn)
        self._bound = Cout
        return Cout, Craw, S, pars, sn

    def _mark_batch(self, *lazies):
        """the asynchronous call just made queued one batch of downloads: its generation number goes to the buffers it fills"""
        g = C.c_int64(0)
        L.check(L.lib.cnmfe_copy_generation(self._ctx, C.byref(g)))
        for x in lazies:
            blk = x._blk()
            if blk is not None:
                blk.gen = g.value

    def post_process_spatial(self, A_full, d1, d2):
        K, cp, ri, va = _csc(A_full, d1 * d2)
        keep = np.zeros(cp[-1], dtype=np.uint8)
        L.check(L.lib.cnmfe_post_process_spatial(self._ctx, d1, d2, K, _p(cp, L.i64p), _p(ri, L.i32p), _p(va, L.f32p), _p(keep, L.u8p)))
        out = sp.csc_matrix((va * keep, ri.copy(), cp.copy()), shape=(d1 * d2, K))
        out.eliminate_zeros()
        return out

    # ---- measurement ---------------------------------------------------------------
    def profile(self, on=True):
        """True / 1: events around every kernel; 2: only around the roofline kernels (cnmfe.h); False: off"""
        L.check(L.lib.cnmfe_profile_enable(self._ctx, int(on)))

    def profile_reset(self):
        L.check(L.lib.cnmfe_profile_reset(self._ctx))

    def profile_table(self):
        n = L.lib.cnmfe_profile_count(self._ctx)
        out = {}
        for i in range(n):
            name = C.create_string_buffer(128); ms = C.c_double(); calls = C.c_int64()
            L.check(L.lib.cnmfe_profile_get(self._ctx, i, name, 128, C.byref(ms), C.byref(calls)))
            out[name.value.decode()] = dict(total_ms=ms.value, calls=calls.value)
        return out

    def synchronize(self):
        L.check(L.lib.cnmfe_synchronize(self._ctx))

    def set_option(self, name, value):
        L.check(L.lib.cnmfe_set_option(self._ctx, name.encode(), int(value)))
        self._opts_set[name] = int(value)

    def get_option(self, name, default):
        """the value this engine's option has: what set_option gave it last, else what CNMFE_OPTS preset at cnmfe_create, else `default` (the library's own)"""
        if name in self._opts_set:
            return self._opts_set[name]
        for kv in os.environ.get("CNMFE_OPTS", "").split(","):
            k, _, v = kv.partition("=")
            if k == name and v:
                try:
                    return int(v)
                except ValueError:
                    pass
        return default
