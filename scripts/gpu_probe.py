"""GPU probe: per-kernel timing of one CNMF-E iteration at a named config; R1 tile-variant sweep.
usage: python scripts/gpu_probe.py --cfg c2 [--variants 0,1,2,3,4] [--iters 2]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = {"tiny": (64, 64, 400, 10, 15, 1), "c2": (256, 256, 3000, 200, 15, 1), "c3": (512, 512, 10000, 500, 15, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="c2")
    ap.add_argument("--variants", default="0,1,2,3,4")
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--alg", default="hals")
    ap.add_argument("--deconv", action="store_true")
    a = ap.parse_args()
    import torch
    from cnmf_e_amd import synth
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r, seed = CFG[a.cfg]
    t0 = time.time()
    f = synth.make_factors(d1, d2, T, K, seed)
    print("factors %.1fs" % (time.time() - t0), flush=True)
    t0 = time.time()
    Yd = synth.make_video_device(f, "cuda:0")
    torch.cuda.synchronize()
    print("video on device %.1fs  %.2f GB" % (time.time() - t0, Yd.numel() * 4 / 1e9), flush=True)
    eng = Engine(0)
    video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
    video.upload_block_device((0, 0), Yd.data_ptr())
    del Yd
    torch.cuda.empty_cache()
    s = Sources2D(video, Options(ring_radius=r, spatial_algorithm=a.alg, maxIter=5, deconv_flag=a.deconv), f.A_init, f.C_init, f.sn)
    eng.profile(True); eng.set_option("r1_delta", 0)   # variant timings: always the full ring sweep
    # R1 variant sweep (spatial-call flavour: no A_prev; temporal-call flavour: all neurons)
    bytes_r1 = 4.0 * d1 * d2 * T * 2
    A_b = f.A_init.astype(np.float32)
    for v in [int(x) for x in a.variants.split(",")]:
        eng.set_option("r1_variant", v)
        for flavour, (Ab, Cb) in (("noAC", (None, None)), ("AC", (A_b, f.C_init))):
            eng.residual(0, Ab, Cb)             # warm
            eng.profile_reset()
            for _ in range(3):
                eng.residual(0, Ab, Cb)
            tab = eng.profile_table()
            key = "residual_r1" if "residual_r1" in tab and tab["residual_r1"]["calls"] else "residual_r1_generic"
            ms = tab[key]["total_ms"] / max(1, tab[key]["calls"])
            print("R1 variant %2d %-4s: %8.3f ms  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (v, flavour, ms, bytes_r1 / ms / 1e6, bytes_r1 / ms / 1e6 / 80), flush=True)
    eng.set_option("r1_variant", 11); eng.set_option("r1_delta", 1)
    for it in range(a.iters):
        eng.profile_reset()
        torch.cuda.synchronize(); t0 = time.time()
        info = s.update_background_parallel()
        torch.cuda.synchronize(); t1 = time.time()
        s.update_spatial_parallel()
        torch.cuda.synchronize(); t2 = time.time()
        s.update_temporal_parallel()
        torch.cuda.synchronize(); t3 = time.time()
        print("iter %d: bg %.3fs spatial %.3fs temporal %.3fs total %.3fs  info=%s" % (it, t1 - t0, t2 - t1, t3 - t2, t3 - t0, info), flush=True)
        tab = eng.profile_table()
        for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["total_ms"]):
            if v["calls"]:
                print("   %-24s %9.3f ms  x%d" % (k, v["total_ms"], v["calls"]))
        print("   K=%d nnz(A)=%d  C range %.3g..%.3g" % (s.A.shape[1], s.A.nnz, s.C.min(), s.C.max()), flush=True)
    if a.iters:
        import cProfile, pstats, io
        pr = cProfile.Profile(); pr.enable()
        s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
        pr.disable()
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28)
        print("\n".join(l[:150] for l in st.getvalue().splitlines()[:48]))
    # planted-model sanity
    A = s.A
    cors = []
    for k in range(min(K, 50)):
        a_t = f.A_true[:, k].toarray().ravel(); a_e = A[:, k].toarray().ravel()
        if a_e.std() > 0:
            cors.append(np.corrcoef(a_t, a_e)[0, 1])
    print("median corr(A, A_true) over first 50: %.3f ; corr(C,C_true) median %.3f" % (
        np.median(cors), np.median([np.corrcoef(s.C[k], f.C_true[k])[0, 1] for k in range(min(K, 50))])))
    eng.close()


if __name__ == "__main__":
    main()
