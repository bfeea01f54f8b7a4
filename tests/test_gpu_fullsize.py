"""Size-independent properties at BASELINE.json's headline size (512x512x10000 fp32, K=500, r=15, one patch).
Everything is checked on the device against torch float64 so that the 10.5 GB video never crosses PCIe."""
import numpy as np
import pytest

from cnmf_e_amd import synth

pytestmark = pytest.mark.gpu
D1, D2, T, K, R, SEED = 512, 512, 10000, 500, 15, 2


@pytest.fixture(scope="module")
def big():
    import torch
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo
    if torch.cuda.get_device_properties(0).total_memory < 80e9:
        pytest.skip("needs an 80+ GB GPU")
    f = synth.make_factors(D1, D2, T, K, SEED)
    Yd = synth.make_video_device(f, "cuda:0")
    torch.cuda.synchronize()
    eng = Engine(0)
    video = PatchedVideo(D1, D2, T, [D1, D2], R, eng)
    video.upload_block_device((0, 0), Yd.data_ptr())
    eng.ring_init(0, R)
    yield dict(f=f, Y=Yd, eng=eng, video=video, torch=torch)
    eng.close()


def test_r1_zero_ring_is_plain_subtraction(big):
    """W = 0  =>  Ysig = Y - b0 exactly (the residual expression with no background fluctuation term)."""
    torch, eng, Y = big["torch"], big["eng"], big["Y"]
    W = eng.ring_csr(0)
    eng.ring_set_values(0, np.zeros(W.nnz, dtype=np.float32))
    b0 = np.linspace(900, 1100, D1 * D2).astype(np.float32)
    eng.set_b0(0, b0)
    out = torch.empty((T, D1 * D2), dtype=torch.float32, device="cuda")
    eng.residual(0, None, None, out_dev_ptr=out.data_ptr())
    ref = Y - torch.from_numpy(b0).cuda()[None, :]
    assert float((out - ref).abs().max()) <= 2e-3          # |Y| ~ 2e3: one fp32 ulp of the operands
    del out, ref


def test_r1_uniform_ring_on_spatially_constant_video(big):
    """With the initial ring (rows sum to 1) and a video that is constant in space, W*(R - mean_t R) equals the
    fluctuation itself, so Ysig = mean_t(Y) - b0 at every frame: a checksum that touches every weight and pixel."""
    torch, eng = big["torch"], big["eng"]
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo
    Tn = 256
    f_t = torch.linspace(-50, 50, Tn, device="cuda") ** 2 / 50.0 + 1000.0
    Yc = f_t[:, None].expand(Tn, D1 * D2).contiguous()
    e2 = Engine(0)
    v2 = PatchedVideo(D1, D2, Tn, [D1, D2], R, e2)
    v2.upload_block_device((0, 0), Yc.data_ptr())
    e2.ring_init(0, R)
    e2.set_b0(0, np.full(D1 * D2, 123.0, dtype=np.float32))
    out = torch.empty((Tn, D1 * D2), dtype=torch.float32, device="cuda")
    e2.residual(0, None, None, out_dev_ptr=out.data_ptr())
    expect = float(f_t.double().mean()) - 123.0
    assert float((out - expect).abs().max()) <= 5e-3
    e2.close()


def test_ring_fit_satisfies_normal_equations_at_full_size(big):
    """After fit_ring_model at full size, sampled rows of W solve the reference's ridge system
    (XX' + 1e-5*tr(XX')*I) w = X y' (fit_ring_model.m:103-106), evaluated in float64 on the device."""
    torch, eng, Y, f = big["torch"], big["eng"], big["Y"], big["f"]
    eng.ring_init(0, R)
    A = f.A_init.astype(np.float32)
    _, info = eng.fit_ring_model(0, A, f.C_init)
    assert info["first_run"] and info["frame_stride"] == 1 and info["n_active"] == D1 * D2
    W = eng.ring_csr(0)
    Cc = torch.from_numpy(f.C_init - f.C_init.mean(axis=1, keepdims=True)).cuda().double()
    Acsr = A.tocsr()
    rng = np.random.default_rng(0)
    for m in list(rng.integers(0, D1 * D2, 6)) + [0, D1 * D2 - 1, 17 * D1 + 3]:
        ring = W.indices[W.indptr[m]:W.indptr[m + 1]]
        w_gpu = W.data[W.indptr[m]:W.indptr[m + 1]].astype(np.float64)
        px = np.concatenate([ring, [m]])
        Yp = Y[:, torch.from_numpy(px).cuda()].double().T                          # (p+1) x T
        Bf = Yp - Yp.mean(dim=1, keepdim=True)
        Bf = Bf - torch.from_numpy(Acsr[px].toarray().astype(np.float64)).cuda() @ Cc
        X = torch.cat([Bf[:-1], torch.ones((1, T), dtype=torch.float64, device="cuda")], dim=0)
        XX = X @ X.T
        w = torch.linalg.solve(XX + torch.eye(XX.shape[0], dtype=torch.float64, device="cuda") * torch.trace(XX) * 1e-5, X @ Bf[-1])
        w = w[:-1].cpu().numpy()
        # (default path: fp64 table of the video + fp64 footprint corrections + fp64 Cholesky; what is left is the fp32 storage of W)
        assert np.linalg.norm(w_gpu - w) / np.linalg.norm(w) <= 2e-5, (m, np.linalg.norm(w_gpu - w) / np.linalg.norm(w))


def test_full_iteration_recovers_planted_model_and_reduces_rss(big):
    torch, eng, Y, f, video = big["torch"], big["eng"], big["Y"], big["f"], big["video"]
    from cnmf_e_amd.sources2d import Sources2D, Options
    s = Sources2D(video, Options(ring_radius=R, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)

    def rss():
        # || Ysig - A*C ||^2 with the current background (the residual kernel) and factors, on the device
        out = torch.empty((T, D1 * D2), dtype=torch.float32, device="cuda")
        eng.residual(0, s.A_prev.astype(np.float32), s.C_prev, out_dev_ptr=out.data_ptr())
        At = torch.sparse_coo_tensor(np.vstack(s.A.tocoo().coords), s.A.tocoo().data.astype(np.float32), s.A.shape, device="cuda")
        # constant per-pixel offsets are excluded: the residual kernel uses b0 of the last BACKGROUND update, while the
        # temporal update shifts every trace to min 0 (b0_new absorbs that only at the next background update)
        Ct = torch.from_numpy(np.asarray(s.C)).cuda()
        s1 = torch.zeros(D1 * D2, dtype=torch.float64, device="cuda"); s2 = 0.0
        for t0 in range(0, T, 1000):
            res = (out[t0:t0 + 1000] - torch.sparse.mm(At, Ct[:, t0:t0 + 1000]).T).double()
            s1 += res.sum(dim=0); s2 += float((res ** 2).sum())
        return s2 - float((s1 ** 2).sum()) / T

    s.update_background_parallel()
    r0 = rss()
    s.update_spatial_parallel()
    s.update_temporal_parallel()
    r1 = rss()
    # A_init / C_init are a mild perturbation of the truth, so both sit near the noise floor d*T*sn^2 = 2.6e9;
    # one iteration must not move away from it and must end within 3 % of it
    assert r1 < 1.005 * r0, (r0, r1)
    assert r1 < 1.03 * D1 * D2 * T, r1
    assert np.all(s.C >= 0) and np.allclose(s.C.min(axis=1), 0)
    assert s.A.min() >= 0
    cors = [np.corrcoef(s.A[:, k].toarray().ravel(), f.A_true[:, k].toarray().ravel())[0, 1] for k in range(0, K, 25)]
    assert np.median(cors) > 0.97
    assert np.median([np.corrcoef(s.C[k], f.C_true[k])[0, 1] for k in range(0, K, 25)]) > 0.97


def test_compute_rss_equals_the_literal_objective_at_full_size(big):
    """compute_RSS (Sources2D.m:1358-1510) at full size: the engine's one-read evaluation (pending footprint term + per-pixel constant +
    footprint rows) against the literal expression  sum((Y - A C - (W (Y - b0 - A_prev C_prev) + b0_new))^2)  evaluated with torch sparse
    products in float64, frames in chunks -- this exercises the ring sweep, the pending-term fold and the RSS kernel together."""
    torch, eng, Y, f, video = big["torch"], big["eng"], big["Y"], big["f"], big["video"]
    from cnmf_e_amd.sources2d import Sources2D, Options
    s = Sources2D(video, Options(ring_radius=R, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
    s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
    got, _ = s.compute_RSS()
    W = s.get_W((0, 0))
    dev = "cuda"
    def sp_t(M):
        M = M.tocoo()
        return torch.sparse_coo_tensor(np.vstack([M.row, M.col]), M.data.astype(np.float64), M.shape, device=dev).coalesce()
    Wt, At, Apt = sp_t(W), sp_t(s.A), sp_t(s.A_prev)
    Ct = torch.from_numpy(np.asarray(s.C, dtype=np.float64)).to(dev); Cpt = torch.from_numpy(np.asarray(s.C_prev, dtype=np.float64)).to(dev)
    b0 = torch.from_numpy(s.reconstruct_b0().reshape(-1, order="F").astype(np.float64)).to(dev)
    b0n = torch.from_numpy(np.asarray(s.b0_new, dtype=np.float64).reshape(-1, order="F")).to(dev)
    ref = 0.0
    for t0 in range(0, T, 500):
        Yc = Y[t0:t0 + 500].double().T                                           # d x chunk
        Rm = Yc - b0[:, None] - torch.sparse.mm(Apt, Cpt[:, t0:t0 + 500])
        E = Yc - torch.sparse.mm(At, Ct[:, t0:t0 + 500]) - (torch.sparse.mm(Wt, Rm) + b0n[:, None])
        ref += float((E ** 2).sum())
        del Yc, Rm, E
    assert abs(got - ref) <= 2e-6 * ref, (got, ref)
