"""Pure-Python cost of one iteration of the three update methods (no GPU: a null engine answers every call at once with data of the right shape):
   python scripts/host_python_profile.py [--patch 128] [--world 8 --rank 0]
what the host spends on slicing, search masks and assembling A per iteration -- the floor under a rank's step time in the sharded configuration."""
import argparse, cProfile, os, pstats, sys, time
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--patch", type=int, default=128); ap.add_argument("--world", type=int, default=1); ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
from cnmf_e_amd import synth
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options


class NullEngine:
    supports_lazy_traces = True
    def __init__(self): self.K = self.T = 0; self.bound = None
    def create_patch(self, *x, **k): pass
    def ring_init(self, *x, **k): pass
    def ymean(self, pid): return self._ym[pid]
    def ring_first_run(self, pid): return False
    def bind_traces(self, C, device_ptr=None): self.bound = C
    def fit_ring_model(self, *x, **k): return None, dict(first_run=False, frame_stride=1, n_active=-1, pmax=96)
    def residual(self, *x, **k): return None
    def update_spatial(self, pid, alg, A_patch, C_patch, IND_patch, sn=None, param=3, defer=False):
        out = sp.csc_matrix(A_patch).astype(np.float32).copy()            # (keeps the footprints' real sparsity from iteration to iteration)
        if defer:
            return lambda connected_fov=None: out if connected_fov is None else (out, out.copy())
        return out
    def post_process_spatial(self, A_full, d1, d2): return sp.csc_matrix(A_full)
    def hals_temporal(self, pid, A, C, maxIter=5, want_C=True, want_raw=True): return None, None, np.ones(A.shape[1], np.float32)
    def stitch_begin(self, K, T): self.K, self.T = K, T
    def stitch_add(self, ind): pass
    def stitch_allreduce(self, group): pass
    def stitch_finish(self, subtract_min, want=True):
        self.bound = np.zeros((self.K, self.T), np.float32); return self.bound
    def b0(self, pid): return np.zeros(self._d[pid], np.float32)


d1 = d2 = 512; T = 400; K = 500; r = 15
f = synth.make_factors(d1, d2, T, K, 2)
eng = NullEngine()
video = PatchedVideo(d1, d2, T, [a.patch, a.patch], r, eng, rank=a.rank, world_size=a.world)
eng._ym = {video.pid[i]: np.zeros(video.block_pix[i].size) for i in video.owned}
eng._d = {video.pid[i]: video.patch_pix[i].size for i in video.owned}
s = Sources2D(video, Options(ring_radius=r), f.A_init, f.C_init, f.sn)
def it():
    s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
it(); it()
t = time.perf_counter()
for _ in range(a.iters): it()
print("%d owned patches: %.2f ms of Python per iteration" % (len(video.owned), (time.perf_counter() - t) / a.iters * 1e3))
pr = cProfile.Profile(); pr.enable(); it(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
