"""TEST DOUBLE ONLY: an Engine-shaped object whose kernels are the float64 oracle functions.

It exists so that the host-side logic of cnmf_e_amd.sources2d (patch sharding over ranks, slicing,
stitching, the all-reduce / all-gather of the N>1 path) can be exercised on CPU under gloo.  It is
never importable from the product package and never used on a GPU box for parity claims."""
import numpy as np
import scipy.sparse as sp

import cnmfe_oracle as orc


class FakeEngine:
    def __init__(self):
        self.p = {}

    def create_patch(self, pid, patch_rect, block_rect, d1, d2, T):
        pr = np.asarray(patch_rect); br = np.asarray(block_rect)
        nrb, ncb = int(br[1] - br[0] + 1), int(br[3] - br[2] + 1)
        self.p[pid] = dict(pr=pr, br=br, d1=d1, d2=d2, T=T, Y=np.zeros((T, nrb * ncb), np.float64), ip=orc.ind_patch_mask(pr, br))

    def upload_block(self, pid, Y, t0=0):
        self.p[pid]["Y"][t0:t0 + Y.shape[0]] = Y

    def ymean(self, pid):
        return self.p[pid]["Y"].mean(axis=0)

    def ring_init(self, pid, radius, num_neighbors=None):
        q = self.p[pid]
        rs, cs = orc.get_nhood(radius, num_neighbors)
        q["W"] = orc.build_ring_W(q["pr"], q["br"], q["d1"], q["d2"], rs, cs).tocsr()
        q["b0"] = np.zeros(int(q["ip"].sum()))

    def ring_csr(self, pid):
        return self.p[pid]["W"]

    def ring_first_run(self, pid):
        row = self.p[pid]["W"].getrow(0)
        vals = set(np.unique(row.data).tolist())
        if row.nnz < row.shape[1]:
            vals.add(0.0)
        return len(vals) == 2

    def b0(self, pid):
        return self.p[pid]["b0"]

    def set_noise(self, pid, sn_block):
        self.p[pid]["sn_b"] = np.asarray(sn_block, dtype=np.float64).ravel()

    def fit_ring_model(self, pid, A_block, C_block, thresh_outlier=float("nan"), with_projection=True, want_b0=True):
        q = self.p[pid]
        A = None if A_block is None else sp.csc_matrix(A_block).astype(np.float64)
        W, b0 = orc.fit_ring_model(q["Y"].T, A, C_block, q["W"], thresh_outlier,
                                   q["sn_b"][q["ip"]] if "sn_b" in q else None, q["ip"], with_projection)
        q["W"], q["b0"] = W.tocsr(), b0
        return b0, {}

    def residual(self, pid, A_prev_block=None, C_prev=None, want=False):
        q = self.p[pid]
        A = None if A_prev_block is None else sp.csc_matrix(A_prev_block).astype(np.float64)
        q["Ysig"] = orc.residual_ysig(q["Y"].T, A, C_prev, q["W"], q["b0"], q["ip"])
        q["res_AC"] = (A, None if C_prev is None else np.asarray(C_prev, dtype=np.float64))
        return q["Ysig"].T if want else None

    @staticmethod
    def _dopt(deconv_options):
        o = dict(smin=-5.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
        o.update({k: v for k, v in (deconv_options or {}).items() if k in o})
        return o

    def hals_temporal_deconv(self, pid, A_patch, C_patch, maxIter, deconv_options, kernel_pars=None, want_all=True):
        import oasis_oracle as oo
        A = sp.csc_matrix(A_patch).astype(np.float64)
        Cn, Craw, S, sn, kp = oo.HALS_temporal_deconv(self.p[pid]["Ysig"], A, np.asarray(C_patch, dtype=np.float64), maxIter, **self._dopt(deconv_options))
        aa = np.asarray(A.multiply(A).sum(axis=0)).ravel()
        self._last = (Craw, aa)
        return Cn, Craw, S, sn, np.array([0.0 if g is None else g for g in kp]), aa

    def deconv_temporal(self, C_raw, deconv_options, overwrite=False):
        import oasis_oracle as oo
        return oo.deconvTemporal(np.asarray(C_raw, dtype=np.float64), **self._dopt(deconv_options))

    def get_sn(self, pid):
        import oasis_oracle as oo
        return np.array([oo.GetSn(row) for row in self.p[pid]["Ysig"]], dtype=np.float32)

    def update_spatial(self, pid, algorithm, A_patch, C_patch, IND_patch, sn=None, param=3):
        Y = self.p[pid]["Ysig"]
        A = sp.csc_matrix(A_patch).astype(np.float64)
        IND = sp.csc_matrix(IND_patch).toarray().astype(bool)
        if algorithm == "hals":
            out = orc.HALS_spatial(Y, A, C_patch, IND, param)
        elif algorithm == "hals_thresh":
            out = orc.HALS_spatial_thresh(Y, A, C_patch, IND, param, sn)
        else:
            out = orc.nnls_spatial(Y, A, C_patch, IND, param)
        return sp.csc_matrix(out)

    def compute_rss(self, pid, A_patch, C_patch, b0_block, b0_new_patch):
        q = self.p[pid]
        ip = q["ip"]
        Yb = q["Y"].T.astype(np.float64)
        YmAC = Yb[ip] - (sp.csc_matrix(A_patch).astype(np.float64) @ np.asarray(C_patch, dtype=np.float64) if A_patch is not None and A_patch.shape[1] else 0.0)
        R = Yb - np.asarray(b0_block, dtype=np.float64)[:, None]
        Ap, Cp = q.get("res_AC", (None, None))
        if Ap is not None and Ap.shape[1]:
            R = R - Ap @ Cp
        return float(np.sum((YmAC - (q["W"] @ R + np.asarray(b0_new_patch, dtype=np.float64)[:, None])) ** 2))

    def fast_temporal(self, pid, A_patch, want_raw=True):
        aa, C_raw = orc.fast_temporal(self.p[pid]["Ysig"], sp.csc_matrix(A_patch).astype(np.float64))
        self._last = (C_raw, aa)
        return C_raw, aa

    def hals_temporal(self, pid, A_patch, C_patch, maxIter=5, want_C=True, want_raw=True):
        A = sp.csc_matrix(A_patch).astype(np.float64)
        C, C_raw, _ = orc.HALS_temporal(self.p[pid]["Ysig"], A, C_patch, maxIter, None)
        aa = np.asarray(A.multiply(A).sum(axis=0)).ravel()
        self._last = (C_raw, aa)
        return C, C_raw, aa

    # the stitch accumulator of the real engine (cnmfe_stitch_*), on the host
    def stitch_begin(self, K, T):
        self._acc = np.zeros((K, T + 1))

    def stitch_add(self, ind):
        C_raw, aa = self._last
        self._acc[ind, :-1] += C_raw * aa[:, None]                       # update_temporal_parallel.m:274
        self._acc[ind, -1] += aa                                        # :275

    def stitch_allreduce(self, group):
        import torch
        import torch.distributed as td
        t = torch.from_numpy(self._acc)
        td.all_reduce(t, group=group)

    def stitch_finish(self, subtract_min, want=True):
        aa = self._acc[:, -1:].copy()
        aa[aa == 0] = 1                                                  # :279
        C_raw = self._acc[:, :-1] / aa                                   # :280
        if subtract_min:
            C_raw = C_raw - C_raw.min(axis=1, keepdims=True)             # :285
        return np.ascontiguousarray(C_raw, dtype=np.float32)

    def post_process_spatial(self, A_full, d1, d2):
        A = sp.csc_matrix(A_full).toarray().astype(np.float64)
        return sp.csc_matrix(orc.post_process_spatial(A.reshape(d1, d2, A.shape[1], order="F")))


class LazyFakeEngine(FakeEngine):
    """the same double with the asynchronous surface of the real engine (Engine.supports_lazy_traces): update_spatial(defer=True) returns a fetch
    closure and stitch_finish(want="lazy") hands out the traces -- so the orderings sources2d only takes with such an engine (the temporal update's
    residual requested under the spatial sweeps, the next patch's slices cut before this patch's fetch, the connectivity constraint applied with the
    fetch of a whole-FOV patch) run on CPU.  The spatial result is computed when the call is made, from the residual resident at that moment: the
    real engine has the sweeps queued by then and only records a later residual request as pending."""
    supports_lazy_traces = True

    def __init__(self):
        super().__init__()
        self.calls = []

    def residual(self, pid, A_prev_block=None, C_prev=None, want=False):
        self.calls.append(("residual", pid))
        return super().residual(pid, A_prev_block, C_prev, want)

    def update_spatial(self, pid, algorithm, A_patch, C_patch, IND_patch, sn=None, param=3, defer=False):
        self.calls.append(("update_spatial", pid))
        out = super().update_spatial(pid, algorithm, A_patch, C_patch, IND_patch, sn, param)
        if not defer:
            return out

        def fetch(connected_fov=None):
            self.calls.append(("fetch", pid))
            if connected_fov is None:
                return out
            return out, self.post_process_spatial(out, *connected_fov)
        return fetch

    def stitch_finish(self, subtract_min, want=True):
        return super().stitch_finish(subtract_min)


class LateFakeEngine(LazyFakeEngine):
    """... and with the real engine's queued download: fetch.start() (cnmfe_update_spatial_fetch_async) and fetch(compact=True) (no stored zeros; with
    connected_fov the raw update comes as a recipe) -- the surface behind which sources2d collects a patch's result one patch late"""
    def update_spatial(self, pid, algorithm, A_patch, C_patch, IND_patch, sn=None, param=3, defer=False):
        inner = super().update_spatial(pid, algorithm, A_patch, C_patch, IND_patch, sn, param, defer)
        if not defer:
            return inner

        def comp(M):
            M = sp.csc_matrix(M).copy(); M.eliminate_zeros(); M.sort_indices()
            return M

        def fetch(connected_fov=None, compact=False):
            r = inner(connected_fov)
            if not compact:
                return r
            if connected_fov is None:
                return comp(r)
            raw, pp = r
            return (lambda: comp(raw)), comp(pp)

        def start():
            self.calls.append(("start", pid))
        fetch.start = start
        return fetch

    # ... and the temporal jobs of the real engine (cnmfe_hals_temporal_job / cnmfe_temporal_jobs_sweep / cnmfe_stitch_add_job): every patch is set up first,
    # the sweeps of all patches run together, each job is then added to the stitch.  Here a job is the per-patch update evaluated at the sweep.
    supports_temporal_jobs = True

    def stitch_begin(self, K, T):
        super().stitch_begin(K, T)
        self._jobs = []

    def hals_temporal_job(self, pid, A_patch, C_patch, maxIter, deconv_options=None, kernel_pars=None):
        self.calls.append(("job", pid))
        self._jobs.append(dict(args=(pid, A_patch, np.array(C_patch, dtype=np.float64, copy=True), maxIter, deconv_options), done=None))
        return len(self._jobs) - 1

    def temporal_jobs_sweep(self):
        self.calls.append(("sweep", len(self._jobs)))
        for j in self._jobs:
            if j["done"] is None:
                pid, A, Cm, maxIter, dopt = j["args"]
                if dopt is None:
                    self.hals_temporal(pid, A, Cm, maxIter, want_C=False, want_raw=False)
                else:
                    self.hals_temporal_deconv(pid, A, Cm, maxIter, dopt, want_all=None)
                j["done"] = self._last

    def stitch_add_job(self, job, ind):
        self.calls.append(("add_job", job))
        assert self._jobs[job]["done"] is not None, "job %d added before the sweep" % job
        self._last = self._jobs[job]["done"]
        self.stitch_add(ind)
