// B2b, round 6 (second half): the packed ring solve with a SHORT set-up.
//
// A pixel of k_ring_solve6 (ring_solve_packed.hpp) spends a third of its time before its first useful FMA: geometry -> CSR row of every ring pixel -> column
// entries -> slot in the centre block's list -> (mask) -> trace number -> slot in the ring pixel's block -> U~: six dependent gathers, ~16 us of the 46 us a
// pixel takes at two waves per SIMD (profiles/r06/solve_f32_call2*.txt; the phase probe of ring_solve_inv.hpp: 3.0 ms of set-up + loads).  What the chain
// looks up is, per neuron, two IMAGES sampled on the pixel's ring -- U~(:, k) and A(:, k) -- so they are laid out as images:
//   * per fit, k_nwin_build writes U~ and A of every neuron densely over its footprint's bounding box dilated by TWICE the ring radius (U~ is non-zero wherever
//     the video is, and a ring that touches the footprint at one end reaches two radii from it at the other: ~6 k entries per neuron, ~35 MB at K = 500), and the host adds to the block lists, per list position, the neuron's window (origin, size, offset);
//   * the solve reads its centre block's list (one load per lane), keeps the neurons whose bounding box, dilated by one radius, contains the centre (the only ones whose footprint a ring
//     of this radius can reach: ~4 of the ~18 on a block's list) and samples their two images at its ring pixels with plain index arithmetic: three dependent
//     loads instead of six, all candidates' samples in flight at once.  A candidate whose samples are all zero on ring and centre is skipped.
// Same neurons in the same (ascending) order as before, the same rank-2 corrections: the weights are bit-identical to k_ring_solve6's (tests/test_gpu_solve_r6.py).
#pragma once
#include "ring_solve_packed.hpp"

namespace cnmfe {

constexpr int NWIN_MAX = 32768;                      // entries of one neuron's window; a larger footprint sends the fit to k_ring_solve6

struct StageArgs {
    const int *lst_ptr;                              // per 16x16 block: its list of traces
    const int *lmeta;                                // per list position 8 ints: k, r0, c0, h (rows), w (columns), offset (the window, block-region coordinates),
                                                     // first | last << 16 row and column of the box a centre must lie in for its ring to reach the footprint
    const double *nwu; const float *nwa;             // the windows: U~ and A, column-major inside a window
};

// U~ and A of neuron k = blockIdx.x over its window
__global__ void __launch_bounds__(256) k_nwin_build(const int *__restrict__ nmeta, BgGeom g, int K, const int *__restrict__ arow, const int *__restrict__ acol,
                                                    const float *__restrict__ aval, const int *__restrict__ lst_ptr, const short *__restrict__ slot_of,
                                                    const double *__restrict__ Ut, double *__restrict__ nwu, float *__restrict__ nwa, int *__restrict__ errflag) {
    const int k = blockIdx.x;
    const int r0 = nmeta[8 * k], c0 = nmeta[8 * k + 1], h = nmeta[8 * k + 2], w = nmeta[8 * k + 3], off = nmeta[8 * k + 4];
    int bad = 0;
    for (int idx = threadIdx.x; idx < h * w; idx += 256) {
        const int r = r0 + idx % h, c = c0 + idx / h;
        const int q = c * g.nr_b + r, blk = (c >> 4) * g.nbr + (r >> 4), lp = lp_of(r & 15, c & 15);
        const int sl = slot_of[(int64_t)blk * K + k];
        float av = 0.f;
        for (int e = arow[q]; e < arow[q + 1]; ++e) if (acol[e] == k) av = aval[e];
        bad |= (sl < 0 && av != 0.f);                        // (a footprint pixel in a block that does not list the neuron: the lists are broken)
        nwu[off + idx] = sl >= 0 ? Ut[((int64_t)lst_ptr[blk] + sl) * 256 + lp] : 0.0;
        nwa[off + idx] = av;
    }
    if (bad) atomicOr(errflag, 1);
}

template <int NT>
__global__ void __launch_bounds__(64, (NT <= 2 ? 4 : (NT <= 3 ? 3 : (NT <= 6 ? 2 : 1))))
k_ring_solve8(const double *__restrict__ sys, StageArgs sa_, BgGeom g, const int *__restrict__ dr, const int *__restrict__ dc, const double *__restrict__ rowsum,
              const unsigned char *__restrict__ active, float *__restrict__ W, int *__restrict__ errflag, int probe, const int *__restrict__ pix,
              double *__restrict__ lam_out) {
    constexpr int N = 16 * NT, NTILE = (NT * (NT + 1)) / 2;
    __shared__ int s_q[N + 1];                                          // block-region pixel of ring neighbour a ([N]: the centre), -1: outside the field of view
    __shared__ int s_rc[N + 1];                                         // its block-region row | column << 16
    __shared__ __attribute__((aligned(16))) double s_u[RSP_NS][N + 2];
    __shared__ __attribute__((aligned(16))) double s_vec[3][N];
    __shared__ __attribute__((aligned(16))) double s_core[16 * RS_DS + 4 * 64];
    double *s_blk = s_core;
    double (*s_part)[64] = reinterpret_cast<double (*)[64]>(s_core + 16 * RS_DS);
    float (*s_a)[N + 2] = reinterpret_cast<float (*)[N + 2]>(s_core);  // (shares the factorisation's exchange buffers, as in k_ring_solve6)
    static_assert(sizeof(float) * RSP_NS * (N + 2) <= sizeof(double) * (16 * RS_DS + 4 * 64), "staged A values do not fit the exchange buffers");
    const int64_t m = pix ? pix[blockIdx.x] : (int)blockIdx.x;
    if (active && !active[m]) return;
    const int lane = threadIdx.x, c = lane & 15, rq = lane >> 4;
    const int p = g.p;
    const double *sp = sys + m * (int64_t)(NTILE * 256 + N);
    const int mi = (int)m;
    const int rbm = mi % g.nr + g.roff, cbm = mi / g.nr + g.coff;
    const int blkm = (cbm >> 4) * g.nbr + (rbm >> 4);
    const bool corr = sa_.lmeta != nullptr && !(probe & 8);
    // (in-kernel laps of one wave, scripts/probes/solve_r6/laps.py: a build with -DRSP_DEBUG and solve_probe = 16384)
#ifdef RSP_DEBUG
    long long lap[8]; int nlap = 0;
#define RSP_LAP() do { lap[nlap++] = wall_clock64(); } while (0)
    RSP_LAP();
#else
#define RSP_LAP() do {} while (0)
#endif
    // ---- the centre block's list: one position per lane; the candidates are the neurons whose window holds the centre ----
    unsigned long long cmask = 0;
    int mr0 = 0, mc0 = 0, mh = 0, mw = 0, moff = 0;
    if (corr) {
        const int l0 = sa_.lst_ptr[blkm], nl = sa_.lst_ptr[blkm + 1] - l0;
        bool cand = false;
        if (lane < nl) {
            const int4 m0 = reinterpret_cast<const int4 *>(sa_.lmeta)[2 * (l0 + lane)];
            const int4 m1 = reinterpret_cast<const int4 *>(sa_.lmeta)[2 * (l0 + lane) + 1];
            mr0 = m0.y; mc0 = m0.z; mh = m0.w; mw = m1.x; moff = m1.y;
            cand = rbm >= (m1.z & 0xffff) && rbm <= (m1.z >> 16) && cbm >= (m1.w & 0xffff) && cbm <= (m1.w >> 16);
        }
        cmask = __ballot(cand);
    }
    // ---- the system: 2 NTILE coalesced 16-byte loads, all in flight at once -- issued FIRST: the set-up below (lists, geometry, the windows' samples) runs under their latency ----
    double4_t T[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const double2 v0 = reinterpret_cast<const double2 *>(sp)[(t * 2) * 64 + lane], v1 = reinterpret_cast<const double2 *>(sp)[(t * 2 + 1) * 64 + lane];
        T[t] = (double4_t){v0.x, v0.y, v1.x, v1.y};
    }
    // ---- geometry ----
#pragma unroll 1
    for (int a = lane; a <= N; a += 64) {
        int q = -1, rc = 0;
        if (a < p || a == N) {
            const int rb = a < p ? rbm + dr[a] : rbm, cb = a < p ? cbm + dc[a] : cbm;
            const int ra = g.r0_abs + rb, ca = g.c0_abs + cb;
            if (ra >= 1 && ra <= g.d1 && ca >= 1 && ca <= g.d2) { q = cb * g.nr_b + rb; rc = rb | (cb << 16); }
        }
        s_q[a] = q; s_rc[a] = rc;
    }
    __syncthreads();
    for (int a = lane; a < N; a += 64) {
        const int q = s_q[a], rc = s_rc[a];
        const int rb = rc & 0xffff, cb = rc >> 16;
        s_vec[0][a] = q >= 0 ? rowsum[((cb >> 4) * g.nbr + (rb >> 4)) * 256 + lp_of(rb & 15, cb & 15)] : 0.0;
        s_vec[1][a] = q >= 0 ? sp[NTILE * 256 + a] : 0.0;
    }
    const double sc = rowsum[blkm * 256 + lp_of(rbm & 15, cbm & 15)];
    RSP_LAP();                                              // 1: list, geometry, border vectors issued
    // ---- staging: up to RSP_NS candidates per round, every sample of the round in flight at once; `live`: the slots with a non-zero A on ring or centre ----
    unsigned live = 0;
    int nst = 0;
    auto stage = [&]() {
        nst = 0; live = 0;
        int r0[RSP_NS], c0[RSP_NS], hh[RSP_NS], ww[RSP_NS], of[RSP_NS];
#pragma unroll
        for (int i = 0; i < RSP_NS; ++i) {
            r0[i] = 0; c0[i] = 0; hh[i] = 0; ww[i] = 0; of[i] = 0;
            if (cmask) {
                const int s = __builtin_amdgcn_readfirstlane(__builtin_ctzll(cmask));
                cmask &= cmask - 1;
                r0[i] = __builtin_amdgcn_readlane(mr0, s); c0[i] = __builtin_amdgcn_readlane(mc0, s); hh[i] = __builtin_amdgcn_readlane(mh, s);
                ww[i] = __builtin_amdgcn_readlane(mw, s); of[i] = __builtin_amdgcn_readlane(moff, s);
                nst = i + 1;
            }
        }
#pragma unroll 1
        for (int a = lane; a <= N; a += 64) {
            const int q = s_q[a], rc = s_rc[a];
            const int rb = rc & 0xffff, cb = rc >> 16;
            double uu[RSP_NS]; float av[RSP_NS];
#pragma unroll
            for (int i = 0; i < RSP_NS; ++i) {
                const int rr = rb - r0[i], cc = cb - c0[i];
                const bool in = q >= 0 && i < nst && rr >= 0 && rr < hh[i] && cc >= 0 && cc < ww[i];
                const int idx = of[i] + cc * hh[i] + rr;
                uu[i] = in ? sa_.nwu[idx] : 0.0; av[i] = in ? sa_.nwa[idx] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < RSP_NS; ++i) {
                s_u[i][a] = uu[i]; s_a[i][a] = av[i];
                if (__ballot(av[i] != 0.f) != 0ull) live |= 1u << i;
            }
        }
        live = (unsigned)__builtin_amdgcn_readfirstlane((int)live);
    };
    auto apply = [&](double4_t (&T)[NTILE]) {
        __syncthreads();
        for (int i = 0; i < nst; ++i) {
            if (!((live >> i) & 1u)) continue;
            rsp_rank2<NT>(T, s_u[i], s_a[i], c, rq);
            const double uN = s_u[i][N], aN = (double)s_a[i][N];
            for (int a = lane; a < N; a += 64) s_vec[1][a] -= fma(s_u[i][a], aN, (double)s_a[i][a] * uN);
        }
        __syncthreads();
    };
    if (cmask) stage();
    RSP_LAP();                                              // 2: staged
#ifdef RSP_DEBUG
    if (probe & 4096) {
        __syncthreads();
        for (int a = lane; a < p; a += 64) { double v = 0.0; for (int i = 0; i < nst; ++i) if ((live >> i) & 1u) v += s_u[i][a]; W[(int64_t)a * g.d + m] = (float)v; }
        return;
    }
#endif
    if (probe & 2048) {                                     // (diagnostic: the staged A summed over the slots, per ring pixel; row p - 1: candidates staged + 100 x live slots)
        __syncthreads();
        for (int a = lane; a < p; a += 64) {
            float v = 0.f;
            for (int i = 0; i < nst; ++i) v += s_a[i][a];
            if (a == p - 1) v = (float)(nst + 100 * __builtin_popcount(live));
            W[(int64_t)a * g.d + m] = v;
        }
        return;
    }
    if (nst) apply(T);
    RSP_LAP();                                              // 3: the system has arrived, corrections applied
    while (cmask) {                                         // more than RSP_NS candidates around one pixel: further rounds under the live tiles (rare)
        stage();
        apply(T);
    }
    // ---- trace, ridge (fit_ring_model.m:106) ----
    double tr = 0.0;
    bool rowex[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) rowex[I] = s_q[16 * I + c] >= 0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c == rq + 4 * r && rowex[I]) tr += T[rs_tix(I, I)][r];
    tr = rs_wave_sum(tr);
    const double lam = (tr + (double)g.Tp) * 1e-5;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c == rq + 4 * r && rowex[I]) T[rs_tix(I, I)][r] += lam;
    if (lam_out && lane == 0) lam_out[m] = lam;
    __syncthreads();
    double wc[NT];
    RSP_LAP();                                              // 4: trace, ridge
    if (!(probe & 2)) rs_factor<NT>(T, s_blk, lane, c, rq);
    __syncthreads();
    RSP_LAP();                                              // 5: factorisation
    rs_solve_core<NT>(T, s_vec, s_blk, s_part, sc, lam, (double)g.Tp, lane, probe | 2, wc);
    RSP_LAP();                                              // 6: substitutions
#ifdef RSP_DEBUG
    if ((probe & 16384) && (m % 509) == 7) {
        if (lane == 0) for (int i = 1; i < nlap; ++i) W[(int64_t)(i - 1) * g.d + m] = (float)(lap[i] - lap[i - 1]);
        return;
    }
#endif
    if (rq == 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int a = 16 * k + c;
            if (a < p) W[(int64_t)a * g.d + m] = s_q[a] >= 0 ? (float)wc[k] : 0.f;
        }
    }
}

}  // namespace cnmfe
