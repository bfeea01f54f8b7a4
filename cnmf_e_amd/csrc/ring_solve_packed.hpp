// B2b, round 4 (second half): the per-pixel ridge solve out of PACKED systems.
//
// What bound k_ring_solve5 (ring_solve.hpp) was its gather: a pixel's (p+1)^2 normal equations are a one-pixel-wide curve through the block-pair covariance
// table, 3.6 useful doubles per 128-byte line, 46 GB of fabric traffic per launch for 11 GB of entries (profiles/r04/bench_c3_v1.json) -- and the table it
// gathered from had just been rewritten by a 7.7 GB sweep (k_cov_correct: base - footprint corrections).  Both go away when the order of the two steps is
// swapped:
//   * ONCE per recording (and frame stride) the systems of the VIDEO's table are gathered by the old code and written out per pixel in the solve's own
//     register-tile order (k_sys_pack = ring_solve_kernel.inc with RS_PACK: [pixel][tile][r / 2][lane][2] doubles + the border vector g): 43 KB per pixel at
//     p = 96, 11.5 GB at 512 x 512;
//   * every fit loads a pixel's system with 42 coalesced 16-byte loads per lane (a wave load = 1 KB of consecutive addresses, all in flight at once, no
//     address tables, no window codes) and applies the footprints' corrections IN REGISTERS:
//         G(a, b) -= sum_k  U~(n_a, k) A(n_b, k) + A(n_a, k) U~(n_b, k)          (bg.hip: the incremental table, U~ = Yc Cc' - A (Cc Cc') / 2)
//     is a symmetric rank-2 update per neuron k that has a pixel on the ring (or under the centre); those neurons are a subset of the centre block's list
//     (<= 64, typically 3-8 of them are present), U~ comes from the window projection's table, A from the CSR rows of the ring pixels.  2 x 84 fp64 FMAs per
//     neuron and lane -- nothing against the 26 us the gather took per pixel.
// Same arithmetic as before up to the order of the corrections' sum (1e-16 relative); everything behind the gather is rs_solve_core, unchanged.
#pragma once
#include "ring_solve_core.hpp"

namespace cnmfe {

// the gather of ring_solve_kernel.inc, writing instead of solving
#define RS_PACK
#define RS_KNAME k_sys_pack
#define RS_EXTRA , const double *__restrict__ fillg, double *__restrict__ sys
#define RS_FILLP fillg
#include "ring_solve_kernel.inc"
#undef RS_KNAME
#undef RS_EXTRA
#undef RS_FILLP
#undef RS_PACK

struct PackArgs {                                   // what the corrections read (arow == nullptr: no footprints, the video's systems as they are)
    const int *arow, *acol; const float *aval;      // CSR rows of A over the block region (pixel q = cb * nr_b + rb)
    const int *lst_ptr, *lst_k; const short *slot_of; int K;   // per 16x16 block: its list of traces, slot of trace k in block b = slot_of[b * K + k]
    const double *Ut;                               // U~[(lst_ptr[b] + slot) * 256 + local pixel]
};
constexpr int RSP_CAP = 4;                          // footprints over one pixel whose (trace, value) entries are cached in LDS; longer rows are read where they lie
constexpr int RSP_CH = 4;                           // neurons per staging round
constexpr int RSP_NS = 8;                           // neurons staged before the system is loaded (two rounds)

// T(a, b) -= u(a) alpha(b) + alpha(a) u(b) on every tile.  A tile (four registers = one vector value) is updated as a whole: element-wise updates interleaved
// across tiles made the compiler copy half-updated vectors around and spill a third of them.
template <int NT>
__device__ __forceinline__ void rsp_rank2(double4_t (&T)[(NT * (NT + 1)) / 2], const double *su, const float *sa, int c, int rq) {
    double ua[NT], aa[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) { ua[I] = -su[16 * I + c]; aa[I] = -(double)sa[16 * I + c]; }
#pragma unroll
    for (int J = 0; J < NT; ++J) {
        double4_t ub, ab;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ub[r] = su[16 * J + rq + 4 * r]; ab[r] = (double)sa[16 * J + rq + 4 * r]; }
#pragma unroll
        for (int I = J; I < NT; ++I) {
            double4_t t = T[rs_tix(I, J)];
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = fma(aa[I], ub[r], fma(ua[I], ab[r], t[r]));
            asm volatile("" : "+v"(t));                     // the FMAs stay HERE: sunk below the loop behind them (with every column's LDS reads hoisted above it) they cost 70 registers
            T[rs_tix(I, J)] = t;
        }
        // (the LDS reads of all NT columns hoisted to the top keep 70 registers alive that the tiles need: the clobber pins the loads at IR level,
        //  the scheduling barrier the FMAs in the machine scheduler)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NT>
__global__ void __launch_bounds__(64, (NT <= 2 ? 4 : (NT <= 3 ? 3 : (NT <= 6 ? 2 : 1))))
k_ring_solve6(const double *__restrict__ sys, PackArgs pa, BgGeom g, const int *__restrict__ dr, const int *__restrict__ dc, const double *__restrict__ rowsum,
              const unsigned char *__restrict__ active, float *__restrict__ W, int *__restrict__ errflag, int probe, const int *__restrict__ pix, const int *__restrict__ npix,
              double *__restrict__ lam_out) {
    constexpr int N = 16 * NT, NTILE = (NT * (NT + 1)) / 2;
    __shared__ int s_q[N + 1];                                          // block-region pixel of ring neighbour a ([N]: the centre), -1: outside the field of view
    __shared__ int s_rs[N + 1];                                         // block * 256 + local pixel (row sums)
    __shared__ int s_bk[N + 1];                                         // block * K
    __shared__ int s_ulp[N + 1];                                        // lst_ptr[block] * 256 + local pixel
    __shared__ int s_en[N + 1];
    __shared__ int s_e0[N + 1];
    __shared__ int s_ec[N + 1][RSP_CAP];
    __shared__ float s_ev[N + 1][RSP_CAP];
    __shared__ unsigned s_mask[2];
    __shared__ __attribute__((aligned(16))) double s_u[RSP_NS][N + 2];
    __shared__ __attribute__((aligned(16))) double s_vec[3][N];
    // the staged A values share their memory with the factorisation's exchange buffers (used only behind the corrections; barriers in between):
    // eight one-wave workgroups per CU must fit 160 KB
    __shared__ __attribute__((aligned(16))) double s_core[16 * RS_DS + 4 * 64];
    double *s_blk = s_core;
    double (*s_part)[64] = reinterpret_cast<double (*)[64]>(s_core + 16 * RS_DS);
    float (*s_a)[N + 2] = reinterpret_cast<float (*)[N + 2]>(s_core);
    static_assert(sizeof(float) * RSP_NS * (N + 2) <= sizeof(double) * (16 * RS_DS + 4 * 64), "staged A values do not fit the exchange buffers");
    if (npix && (int)blockIdx.x >= *npix) return;                       // (a device-side list: ring_solve_inv.hpp -- what its fast path left over; the grid covers the list's capacity)
    const int64_t m = pix ? pix[blockIdx.x] : (int)blockIdx.x;
    if (active && !active[m]) return;
    const int lane = threadIdx.x, c = lane & 15, rq = lane >> 4;
    const int p = g.p;
    const double *sp = sys + m * (int64_t)(NTILE * 256 + N);
    const int mi = (int)m;
    const int rbm = mi % g.nr + g.roff, cbm = mi / g.nr + g.coff;
    const int blkm = (cbm >> 4) * g.nbr + (rbm >> 4);
    const bool corr = pa.arow != nullptr && !(probe & 8);
    if (lane < 2) s_mask[lane] = 0;
    __syncthreads();
    int bad = 0;
#pragma unroll 1
    for (int a = lane; a <= N; a += 64) {
        int q = -1, rs = 0, bk = 0, ulp = 0, en = 0, e0s = 0;
        if (a < p || a == N) {
            const int rb = a < p ? rbm + dr[a] : rbm, cb = a < p ? cbm + dc[a] : cbm;
            const int ra = g.r0_abs + rb, ca = g.c0_abs + cb;
            if (ra >= 1 && ra <= g.d1 && ca >= 1 && ca <= g.d2) {
                q = cb * g.nr_b + rb;
                const int blk = (cb >> 4) * g.nbr + (rb >> 4), lp = lp_of(rb & 15, cb & 15);
                rs = blk * 256 + lp;
                if (corr) {
                    bk = blk * pa.K; ulp = pa.lst_ptr[blk] * 256 + lp;
                    const int e0 = pa.arow[q];
                    en = pa.arow[q + 1] - e0; e0s = e0;
#pragma unroll
                    for (int j = 0; j < RSP_CAP; ++j)
                        if (j < en) {
                            const int col = pa.acol[e0 + j];
                            s_ec[a][j] = col; s_ev[a][j] = pa.aval[e0 + j];
                            const int sl = pa.slot_of[(int64_t)blkm * pa.K + col];
                            if (sl < 0) bad = 1; else atomicOr(&s_mask[sl >> 5], 1u << (sl & 31));
                        }
                    for (int j = RSP_CAP; j < en; ++j) {               // (a pixel under more footprints than the cache holds: rare)
                        const int sl = pa.slot_of[(int64_t)blkm * pa.K + pa.acol[e0 + j]];
                        if (sl < 0) bad = 1; else atomicOr(&s_mask[sl >> 5], 1u << (sl & 31));
                    }
                }
            }
        }
        s_q[a] = q; s_rs[a] = rs; s_bk[a] = bk; s_ulp[a] = ulp; s_en[a] = en; s_e0[a] = e0s;
    }
    __syncthreads();
    // ---- border vectors u (row sums of Bf: corrected by k_rowsum_correct), g (the video's, corrected below) and the scalar s ----
    for (int a = lane; a < N; a += 64) {
        const bool ex = s_q[a] >= 0;
        s_vec[0][a] = ex ? rowsum[s_rs[a]] : 0.0;
        s_vec[1][a] = ex ? sp[NTILE * 256 + a] : 0.0;
    }
    const double sc = rowsum[s_rs[N]];
    // ---- the footprints' corrections: one symmetric rank-2 update per neuron with a pixel on the ring or under the centre.  A round stages U~ and A of up to
    // RSP_CH neurons for every ring pixel in LDS (two dependent gathers: slot, value); the FIRST round runs before the system is loaded -- with the 168 tile
    // registers live the compiler spilled a third of them around this phase --, later rounds (more than RSP_CH neurons on one ring: rare) run under them
    unsigned long long mask = 0;
    int lbm = 0, nst = 0;
    if (corr) {
        mask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[0]) |
               ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[1]) << 32);
        lbm = pa.lst_ptr[blkm];
    }
    auto stage = [&](int slot0) {
        int ks[RSP_CH];
#pragma unroll
        for (int i = 0; i < RSP_CH; ++i) {
            ks[i] = -1;
            if (mask) { const int s = __builtin_ctzll(mask); mask &= mask - 1; ks[i] = pa.lst_k[lbm + s]; ++nst; }
        }
#pragma unroll 1
        for (int a = lane; a <= N; a += 64) {
            const int q = s_q[a];
            int sl[RSP_CH];
#pragma unroll
            for (int i = 0; i < RSP_CH; ++i) sl[i] = (q >= 0 && ks[i] >= 0) ? (int)pa.slot_of[(int64_t)s_bk[a] + ks[i]] : -2;
            double uu[RSP_CH];
#pragma unroll
            for (int i = 0; i < RSP_CH; ++i) {
                bad |= sl[i] == -1;
                uu[i] = sl[i] >= 0 ? pa.Ut[(int64_t)s_ulp[a] + (int64_t)sl[i] * 256] : 0.0;
            }
            const int en = s_en[a];
#pragma unroll
            for (int i = 0; i < RSP_CH; ++i) {
                float av = 0.f;
#pragma unroll
                for (int j = 0; j < RSP_CAP; ++j) if (j < en && s_ec[a][j] == ks[i]) av = s_ev[a][j];
                if (en > RSP_CAP && ks[i] >= 0) {
                    const int e0 = s_e0[a];
                    for (int j = RSP_CAP; j < en; ++j) if (pa.acol[e0 + j] == ks[i]) av = pa.aval[e0 + j];
                }
                s_u[slot0 + i][a] = uu[i]; s_a[slot0 + i][a] = av;
            }
        }
    };
    auto apply = [&](double4_t (&T)[NTILE]) {
        __syncthreads();
        for (int i = 0; i < nst; ++i) {
            rsp_rank2<NT>(T, s_u[i], s_a[i], c, rq);
            const double uN = s_u[i][N], aN = (double)s_a[i][N];
            for (int a = lane; a < N; a += 64) s_vec[1][a] -= fma(s_u[i][a], aN, (double)s_a[i][a] * uN);
        }
        __syncthreads();
    };
    if (mask) stage(0);
    if (mask) stage(RSP_CH);
#ifdef RSP_DEBUG
    if (probe & 4096) {                                     // (diagnostic: the staged U~ summed over the neurons, per ring pixel)
        __syncthreads();
        for (int a = lane; a < p; a += 64) { double v = 0.0; for (int i = 0; i < nst; ++i) v += s_u[i][a]; W[(int64_t)a * g.d + m] = (float)v; }
        return;
    }
#endif
    // ---- the system: 2 NTILE coalesced 16-byte loads, all in flight at once ----
    double4_t T[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const double2 v0 = reinterpret_cast<const double2 *>(sp)[(t * 2) * 64 + lane], v1 = reinterpret_cast<const double2 *>(sp)[(t * 2 + 1) * 64 + lane];
        T[t] = (double4_t){v0.x, v0.y, v1.x, v1.y};
    }
    if (nst) apply(T);
    while (mask) {                                          // more than RSP_NS neurons around one pixel: further rounds under the live tiles (rare; this loop is where the spills are)
        nst = 0;
        stage(0);
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
    if (bad) atomicOr(errflag, 1);
    if (lam_out && lane == 0) lam_out[m] = lam;                         // (the ridge of this fit: the lam0 of an inverse built later, ring_solve_inv.hpp)
    __syncthreads();
    double wc[NT];
    rs_solve_core<NT>(T, s_vec, s_blk, s_part, sc, lam, (double)g.Tp, lane, probe, wc);
    if (rq == 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int a = 16 * k + c;
            if (a < p) W[(int64_t)a * g.d + m] = s_q[a] >= 0 ? (float)wc[k] : 0.f;
        }
    }
}

}  // namespace cnmfe
