// B2b, round 6: the per-pixel ridge solve WITHOUT a factorisation per fit -- out of a cached explicit inverse of the VIDEO's system.
//
// What bounds k_ring_solve6 (ring_solve_packed.hpp) is not its 11.5 GB but the serial chains of a 96 x 96 Cholesky per pixel and fit (six 16 x 16 diagonal
// steps of 240 dependent DPP FMAs, twelve substitution round trips): rounds 5 and 6 rescheduled them, shared them, moved them to fp32 -- 5.8-6.1 ms every time.
// The chains go away when the algebra changes.  A pixel's system (fit_ring_model.m:101-106) is
//         M = G0 + lam D - U~ A~' - A~ U~',        lam = 1e-5 (tr + Tp)                      (D: the rows that exist; border = ones row, see below)
// with G0 the Gram of the VIDEO on the pixel's ring -- the same in every fit -- and the footprints' part of rank 2 per neuron with a pixel on the ring
// (U~ = Yc Cc' - A (Cc Cc') / 2, ring_solve_packed.hpp).  So:
//   * ONCE per recording (k_ring_inverse; again only for pixels whose ridge has drifted):  K = inv(G0 + lam0 D) EXPLICITLY, by the block Cholesky of
//     ring_solve_core.hpp followed by an in-place triangular inversion and W' W product on the matrix pipe (424 MFMAs), written in register-tile order
//     (the bytes of the packed system), with k_g = K g0, k_u = K u0, lam0 and tr(G0) behind it;
//   * every fit (k_ring_apply): V = [A~ | U~] (8 + 8 columns: up to 8 neurons around a pixel), Z_I = sum_J K_IJ V_J and H = V' Z on the matrix pipe
//     (168 MFMAs, no dependency chain longer than one accumulator), Woodbury:
//         inv(G0 + lam0 D + V S V') = K - K V inv(S + H) V' K,      S = [[0, -I], [-I, 0]] = inv(S)
//     with the 16 x 16 `cap = S + H` inverted by Gauss-Jordan steps in registers (lane = row, DPP row broadcasts; A~ block first -- a Gram matrix, positive
//     pivots --, then its negative definite Schur complement: no pivoting needed, and only the steps of neurons that are there), the ones row (:101) by its
//     Schur complement as before, and the ridge's drift  delta = lam - lam0  by a Neumann series  x <- x0 - delta C0 x  whose terms shrink by ~0.4 delta / lam0
//     each (one term at 1e-3, two at 1e-2, four at 1e-1: scripts/probes/solve_inv/apply_emulation.py -- float64 emulation of exactly this arithmetic, 5e-10 of
//     max |w| on 200 pixels incl. image borders and frame stride 2).  K times a vector runs on the vector pipe out of the register tiles (both orientations
//     of a tile: one contracts over lanes, one over registers; partial sums through 2.8 KB of LDS).
//   * what the fast path does not take -- more than 8 neurons around a pixel, a series that has not converged after `maxit` terms, a non-finite result -- puts the
//     pixel on a list; k_ring_solve6 solves the list behind it (and the pixel's inverse is rebuilt at its current ridge).
// A neuron under the centre only (no pixel on the ring: the usual case inside a footprint) corrects g alone: its pair of columns is masked out of `cap`, its U~
// column stays in V for the bookkeeping of  V' K g = V' k_g - H c_g.
#pragma once
#include "ring_solve_packed.hpp"

namespace cnmfe {

constexpr int RI_EXTRA = 8;                          // doubles behind k_g, k_u: [0] lam0, [1] tr(G0) over the rows that exist
__host__ __device__ constexpr int64_t ri_stride(int NT) { return (int64_t)((NT * (NT + 1)) / 2) * 256 + 2 * 16 * NT + RI_EXTRA; }

struct InvArgs {
    const double *kp;                                // the inverses, ri_stride(NT) doubles per pixel
    const double *sys;                               // the packed systems (only their border vector g0 is read)
    const double *csum;                              // sum_t Cc(k, t) over the frames used (k_trace_subsum): u = u0 - A~ csum
    double *lam_out;                                 // the ridge of this fit per pixel (what a rebuild takes as lam0)
    int *flist, *rlist, *fcnt;                       // flist: pixels left to k_ring_solve6 (fcnt[0] of them); rlist: those among them whose inverse is to be rebuilt (fcnt[3]);
                                                     // fcnt[1] Neumann terms taken, fcnt[2] pixels with >= 1 term (statistics, probe bit 512)
    int maxit;
};

__device__ __forceinline__ double ri_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}
// 16 x 16 tile through LDS: nat[rho][gam] -> nat[gam][rho]
__device__ __forceinline__ double4_t ri_transpose(const double4_t &X, double *s_blk, int c, int rq) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s_blk[(rq + 4 * r) * RS_DS + c] = X[r];
    __syncthreads();
    double4_t Y;
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[r] = s_blk[c * RS_DS + rq + 4 * r];
    __syncthreads();
    return Y;
}
__device__ __forceinline__ int ri_perm(int a) { return (a & ~15) + ((a & 3) << 2) + ((a >> 2) & 3); }      // element rq + 4 r of a block -> 4 rq + r

// y = K v on the vector pipe.  Tiles: T[rs_tix(I, J)] (I >= J), lane (c, rq), register r = K[16 I + rq + 4 r][16 J + c].  v in LDS twice: sv[a] and
// svp[ri_perm(a)] (a lane's four rows of a block are 32 consecutive bytes).  A tile serves two products: (K_IJ v_J)[rho] contracts over the 16 lanes of a DPP
// row -- summed by a DPP butterfly, no LDS round trip --, (K_IJ' v_I)[gam] over registers and the four row groups -- the four partial sums of every block row go
// to their own slice of `part` (6 x 64 doubles), so that ONE barrier separates all products from all sums.  The lane-contraction sums land in `so` (row group rq
// writes rows rq + 4 r), then every lane adds the row-group partials of the elements it owns (a0 = lane, a1 = lane + 64) and returns them; `sop` (optional)
// receives the result in the permuted order.
template <int NT>
__device__ __forceinline__ void ri_matvec(const double4_t (&T)[(NT * (NT + 1)) / 2], const double *sv, const double *svp, double *so, double *sop,
                                          double *part, int c, int rq, double &y0, double &y1) {
    constexpr int N = 16 * NT;
    const int lane = rq * 16 + c;
#pragma unroll
    for (int I = 0; I < NT; ++I) {
        double4_t an = {0.0, 0.0, 0.0, 0.0};
        double at = 0.0;
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            const double vc = sv[16 * J + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) an[r] = fma(T[rs_tix(I, J)][r], vc, an[r]);
        }
#pragma unroll
        for (int J = I + 1; J < NT; ++J) {
            const double2 v01 = *reinterpret_cast<const double2 *>(svp + 16 * J + 4 * rq), v23 = *reinterpret_cast<const double2 *>(svp + 16 * J + 4 * rq + 2);
            at = fma(T[rs_tix(J, I)][0], v01.x, at); at = fma(T[rs_tix(J, I)][1], v01.y, at);
            at = fma(T[rs_tix(J, I)][2], v23.x, at); at = fma(T[rs_tix(J, I)][3], v23.y, at);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) an[r] = ri_row_sum(an[r]);
        const double mine = c == 0 ? an[0] : (c == 1 ? an[1] : (c == 2 ? an[2] : an[3]));
        if (c < 4) so[16 * I + rq + 4 * c] = mine;
        part[I * 64 + c * 4 + rq] = at;
        asm volatile("" ::: "memory");                     // (v is re-read per block row: kept in registers across the rows it costs 60 beside the tiles)
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    const int a0 = lane, a1 = lane + 64;
    y0 = 0.0; y1 = 0.0;
    if (a0 < N) {
        const double2 r01 = *reinterpret_cast<const double2 *>(part + (a0 >> 4) * 64 + (a0 & 15) * 4), r23 = *reinterpret_cast<const double2 *>(part + (a0 >> 4) * 64 + (a0 & 15) * 4 + 2);
        y0 = so[a0] + ((r01.x + r01.y) + (r23.x + r23.y));
    }
    if (a1 < N) {
        const double2 r01 = *reinterpret_cast<const double2 *>(part + (a1 >> 4) * 64 + (a1 & 15) * 4), r23 = *reinterpret_cast<const double2 *>(part + (a1 >> 4) * 64 + (a1 & 15) * 4 + 2);
        y1 = so[a1] + ((r01.x + r01.y) + (r23.x + r23.y));
    }
    if (sop) {
        if (a0 < N) sop[ri_perm(a0)] = y0;
        if (a1 < N) sop[ri_perm(a1)] = y1;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// K = inv(G0 + lam0 D) per pixel, explicitly.  After rs_factor the registers hold inv(L_kk)' on the diagonal and L_ik' below it (ring_solve_core.hpp); with
// products of the form X1' X2 only (same register layout in and out):
//   triangular inversion, columns left to right, in place:  W_jj = inv(L_jj) (the diagonal tile transposed through LDS),
//        W_ij = -inv(L_ii) (L_ij W_jj + sum_{j<k<i} L_ik W_kj)         -- L_ik = (tile ik)', inv(L_ii) = (tile ii)', the W_kj above in the column are final
//   K = W' W, rows top to bottom, in place:  K_ab = sum_{k >= a} W_ka' W_kb  (b <= a; the row's own tiles are overwritten behind their last use)
// lam0 = lam_in[m] where a fit has left one (> 0), else 1e-5 (tr(G0) + Tp) -- the ridge of a fit without footprints.
template <int NT>
__device__ __forceinline__ void ri_inverse_pixel(const int64_t m, const double *__restrict__ sys, const BgGeom &g, const int *__restrict__ dr, const int *__restrict__ dc,
                                                 const double *__restrict__ rowsum_base, const double *__restrict__ lam_in, double *__restrict__ kp) {
    constexpr int N = 16 * NT, NTILE = (NT * (NT + 1)) / 2;
    __shared__ int s_rs[N];                                             // block * 256 + local pixel of ring neighbour a, -1: none
    __shared__ __attribute__((aligned(16))) double s_vec[5][N];         // g0, g0 permuted, u0, u0 permuted, a product
    __shared__ __attribute__((aligned(16))) double s_blk[16 * RS_DS];
    __shared__ __attribute__((aligned(16))) double s_part[NT * 64];
    const int lane = threadIdx.x, c = lane & 15, rq = lane >> 4;
    const int p = g.p, mi = (int)m;
    const int rbm = mi % g.nr + g.roff, cbm = mi / g.nr + g.coff;
    const double *sp = sys + m * (int64_t)(NTILE * 256 + N);
    for (int a = lane; a < N; a += 64) {
        int rs = -1;
        if (a < p) {
            const int rb = rbm + dr[a], cb = cbm + dc[a];
            const int ra = g.r0_abs + rb, ca = g.c0_abs + cb;
            if (ra >= 1 && ra <= g.d1 && ca >= 1 && ca <= g.d2) rs = ((cb >> 4) * g.nbr + (rb >> 4)) * 256 + lp_of(rb & 15, cb & 15);
        }
        s_rs[a] = rs;
        const double g0 = rs >= 0 ? sp[NTILE * 256 + a] : 0.0, u0 = rs >= 0 ? rowsum_base[rs] : 0.0;
        s_vec[0][a] = g0; s_vec[1][ri_perm(a)] = g0;
        s_vec[2][a] = u0; s_vec[3][ri_perm(a)] = u0;
    }
    double4_t T[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const double2 v0 = reinterpret_cast<const double2 *>(sp)[(t * 2) * 64 + lane], v1 = reinterpret_cast<const double2 *>(sp)[(t * 2 + 1) * 64 + lane];
        T[t] = (double4_t){v0.x, v0.y, v1.x, v1.y};
    }
    __syncthreads();
    double tr = 0.0;
    bool rowex[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) rowex[I] = s_rs[16 * I + c] >= 0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c == rq + 4 * r && rowex[I]) tr += T[rs_tix(I, I)][r];
    tr = rs_wave_sum(tr);
    const double lin = lam_in ? lam_in[m] : 0.0;
    const double lam0 = lin > 0.0 ? lin : (tr + (double)g.Tp) * 1e-5;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c == rq + 4 * r && rowex[I]) T[rs_tix(I, I)][r] += lam0;
    rs_factor<NT>(T, s_blk, lane, c, rq);
    __syncthreads();
    // ---- W = inv(L), in place ----
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        T[rs_tix(j, j)] = ri_transpose(T[rs_tix(j, j)], s_blk, c, rq);              // inv(L_jj)' -> inv(L_jj); (inv(L_jj)' has served the columns to the left)
#pragma unroll
        for (int i = j + 1; i < NT; ++i) {
            double4_t acc = rs_mfma4(T[rs_tix(i, j)], T[rs_tix(j, j)], (double4_t){0.0, 0.0, 0.0, 0.0});
#pragma unroll
            for (int k = j + 1; k < i; ++k) acc = rs_mfma4(T[rs_tix(i, k)], T[rs_tix(k, j)], acc);
            T[rs_tix(i, j)] = -rs_mfma4(T[rs_tix(i, i)], acc, (double4_t){0.0, 0.0, 0.0, 0.0});
        }
    }
    // ---- K = W' W, in place ----
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int b = 0; b < a; ++b) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = a; k < NT; ++k) acc = rs_mfma4(T[rs_tix(k, a)], T[rs_tix(k, b)], acc);
            T[rs_tix(a, b)] = acc;
        }
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = a; k < NT; ++k) acc = rs_mfma4(T[rs_tix(k, a)], T[rs_tix(k, a)], acc);
        T[rs_tix(a, a)] = acc;
    }
    double *kq = kp + m * ri_stride(NT);
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        reinterpret_cast<double2 *>(kq)[(t * 2) * 64 + lane] = make_double2(T[t][0], T[t][1]);
        reinterpret_cast<double2 *>(kq)[(t * 2 + 1) * 64 + lane] = make_double2(T[t][2], T[t][3]);
    }
    // ---- k_g = K g0, k_u = K u0 ----
    double y0, y1;
    ri_matvec<NT>(T, s_vec[0], s_vec[1], s_vec[4], nullptr, s_part, c, rq, y0, y1);
    if (lane < N) kq[NTILE * 256 + lane] = y0;
    if (lane + 64 < N) kq[NTILE * 256 + lane + 64] = y1;
    ri_matvec<NT>(T, s_vec[2], s_vec[3], s_vec[4], nullptr, s_part, c, rq, y0, y1);
    if (lane < N) kq[NTILE * 256 + N + lane] = y0;
    if (lane + 64 < N) kq[NTILE * 256 + N + lane + 64] = y1;
    if (lane == 0) { kq[NTILE * 256 + 2 * N] = lam0; kq[NTILE * 256 + 2 * N + 1] = tr; }
}
template <int NT>
__global__ void __launch_bounds__(64, (NT <= 2 ? 4 : (NT <= 3 ? 3 : (NT <= 6 ? 2 : 1))))
k_ring_inverse(const double *__restrict__ sys, BgGeom g, const int *__restrict__ dr, const int *__restrict__ dc, const double *__restrict__ rowsum_base,
               const double *__restrict__ lam_in, double *__restrict__ kp) {
    ri_inverse_pixel<NT>((int)blockIdx.x, sys, g, dr, dc, rowsum_base, lam_in, kp);
}
template <int NT>
__global__ void __launch_bounds__(64, (NT <= 2 ? 4 : (NT <= 3 ? 3 : (NT <= 6 ? 2 : 1))))
k_ring_inverse_list(const double *__restrict__ sys, BgGeom g, const int *__restrict__ dr, const int *__restrict__ dc, const double *__restrict__ rowsum_base,
                    const double *__restrict__ lam_in, double *__restrict__ kp, const int *__restrict__ pix, const int *__restrict__ npix) {
    const int n = *npix;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        ri_inverse_pixel<NT>(pix[i], sys, g, dr, dc, rowsum_base, lam_in, kp);
        __syncthreads();
    }
}

// Gauss-Jordan inversion of the 16 x 16 `cap` in place, one matrix over the whole wave: lane (c, rq) holds a[jj] = entry (c, 4 rq + jj).  Step P: the pivot by
// readlane, the pivot row by a DPP row broadcast inside each 16-lane row (lane P of row group rq holds the row's columns 4 rq ..), a row's multiplier -- its entry in
// column P, held by the lane of the same c in row group P / 4 -- by a bpermute.  No pivoting (the order makes the pivots those of two Cholesky factorisations); a
// pivot that has lost seven digits against the diagonal entry it started from (two footprints that meet the ring in the same single pixel: linearly dependent
// columns of A~) sets `bad` -- the pixel is left to the factorising kernel.
template <int P> __device__ __forceinline__ void ri_gj4_step(double (&a)[4], int c, int rq, double dref, int &bad) {
    constexpr int jP = P & 3, gP = P >> 2;
    const double piv = ri_readlane<gP * 16 + P>(a[jP]);
    if (!(fabs(piv) > 1e-7 * fabs(dref))) bad = 1;
    const double inv = ri_rcp(piv);
    double f = __shfl(a[jP], gP * 16 + c);
    f = c == P ? 0.0 : f;
    const double sc = c == P ? inv : 1.0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) a[jj] *= sc;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) rs_fmac_bc_self<P, true>(a[jj], f);          // a[jj] -= f * (row P)[4 rq + jj]
    const double dn = c == P ? inv : -f * inv;
    a[jP] = rq == gP ? dn : a[jP];
    asm volatile("s_nop 0" : "+v"(a[jP]));
}
template <int P, int PEND> __device__ __forceinline__ void ri_gj4(double (&a)[4], int c, int rq, unsigned live, const double *dref, int &bad) {
    if constexpr (P < PEND) {
        if (live & (1u << P)) ri_gj4_step<P>(a, c, rq, dref[P], bad);
        ri_gj4<P + 1, PEND>(a, c, rq, live, dref, bad);
    }
}

template <int NT>
__global__ void __launch_bounds__(64, (NT <= 2 ? 4 : (NT <= 3 ? 3 : (NT <= 6 ? 2 : 1))))
k_ring_apply(InvArgs ia, PackArgs pa, BgGeom g, const int *__restrict__ dr, const int *__restrict__ dc, const double *__restrict__ rowsum,
             const unsigned char *__restrict__ active, float *__restrict__ W, int *__restrict__ errflag, int probe, const int *__restrict__ pix) {
    constexpr int N = 16 * NT, NTILE = (NT * (NT + 1)) / 2;
    __shared__ int s_q[N + 1];                                          // block-region pixel of ring neighbour a ([N]: the centre), -1: outside the field of view
    __shared__ __attribute__((aligned(16))) double s_u[RSP_NS][N + 2];  // U~ of the staged neurons ([N]: at the centre)
    __shared__ float s_a[RSP_NS][N + 2];                                // A of the staged neurons
    __shared__ __attribute__((aligned(16))) double s_gu[2][N];          // u (row sums of Bf), g (corrected below)
    __shared__ double s_cs[RSP_NS];                                     // csum of the staged neurons
    // the set-up's index arrays share their memory with everything behind the staging
    constexpr int X_SETUP = (5 * (N + 1) + 2 * RSP_CAP * (N + 1) + 2) * 4, X_POST = (16 * RS_DS + 128 + 256 + 4 * N) * 8;
    __shared__ __attribute__((aligned(16))) char s_x[X_SETUP > X_POST ? X_SETUP : X_POST];
    int *s_rs = reinterpret_cast<int *>(s_x), *s_bk = s_rs + (N + 1), *s_ulp = s_bk + (N + 1), *s_en = s_ulp + (N + 1), *s_e0 = s_en + (N + 1);
    int (*s_ec)[RSP_CAP] = reinterpret_cast<int (*)[RSP_CAP]>(s_e0 + (N + 1));
    float (*s_ev)[RSP_CAP] = reinterpret_cast<float (*)[RSP_CAP]>(s_e0 + (N + 1) + RSP_CAP * (N + 1));
    unsigned *s_mask = reinterpret_cast<unsigned *>(s_e0 + (N + 1) + 2 * RSP_CAP * (N + 1));
    double *s_blk = reinterpret_cast<double *>(s_x), *s_red = s_blk + 16 * RS_DS, *s_H = s_red + 128;
    static_assert(16 * RS_DS + 128 >= NT * 64, "ri_matvec's partial sums take s_blk and s_red");
    double (*s_vec)[N] = reinterpret_cast<double (*)[N]>(s_H + 256);    // [0] v, [1] v permuted, [2] t, [3] t permuted
    const int64_t m = pix ? pix[blockIdx.x] : (int)blockIdx.x;
    if (active && !active[m]) return;
    const int lane = threadIdx.x, c = lane & 15, rq = lane >> 4;
    const int p = g.p;
    const int mi = (int)m;
    const int rbm = mi % g.nr + g.roff, cbm = mi / g.nr + g.coff;
    const int blkm = (cbm >> 4) * g.nbr + (rbm >> 4);
    const bool corr = pa.arow != nullptr && !(probe & 8);
    const double *kq = ia.kp + m * ri_stride(NT);
    const double *sp = ia.sys + m * (int64_t)(NTILE * 256 + N);
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
                    for (int j = RSP_CAP; j < en; ++j) {
                        const int sl = pa.slot_of[(int64_t)blkm * pa.K + pa.acol[e0 + j]];
                        if (sl < 0) bad = 1; else atomicOr(&s_mask[sl >> 5], 1u << (sl & 31));
                    }
                }
            }
        }
        s_q[a] = q; s_rs[a] = rs; s_bk[a] = bk; s_ulp[a] = ulp; s_en[a] = en; s_e0[a] = e0s;
    }
    __syncthreads();
    for (int a = lane; a < N; a += 64) {
        const bool ex = s_q[a] >= 0;
        s_gu[0][a] = ex ? rowsum[s_rs[a]] : 0.0;
        s_gu[1][a] = ex ? sp[NTILE * 256 + a] : 0.0;
    }
    const double sc = rowsum[s_rs[N]];
    unsigned long long mask = 0;
    int lbm = 0, nst = 0;
    if (corr) {
        mask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[0]) |
               ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[1]) << 32);
        lbm = pa.lst_ptr[blkm];
    }
    bool leave = bad != 0 || __builtin_popcountll(mask) > RSP_NS;        // (wave-uniform below: `bad` is reduced first)
    leave = __builtin_amdgcn_readfirstlane((int)(__ballot(leave) != 0ull)) != 0;
    if (bad) atomicOr(errflag, 1);
    if (leave) {                                                        // more neurons around the pixel than the fast path takes: the factorising kernel's
        if (lane == 0) { const int i = atomicAdd(ia.fcnt, 1); ia.flist[i] = mi; }
        return;
    }
    auto stage = [&](int slot0) {
        int ks[RSP_CH];
#pragma unroll
        for (int i = 0; i < RSP_CH; ++i) {
            ks[i] = -1;
            if (mask) { const int s = __builtin_ctzll(mask); mask &= mask - 1; ks[i] = pa.lst_k[lbm + s]; ++nst; }
        }
#pragma unroll
        for (int i = 0; i < RSP_CH; ++i) if (lane == i) s_cs[slot0 + i] = ks[i] >= 0 ? ia.csum[ks[i]] : 0.0;
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
    // (every slot is staged: an empty one as zeros -- the products below run over all 16 columns)
    stage(0);
    stage(RSP_CH);
    static_assert(RSP_NS == 2 * RSP_CH, "two staging rounds fill the slots");
    if (bad) atomicOr(errflag, 1);
    // ---- the system's inverse: 2 NTILE coalesced 16-byte loads, all in flight at once; k_g, k_u; lam0, tr(G0) ----
    double4_t T[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const double2 v0 = reinterpret_cast<const double2 *>(kq)[(t * 2) * 64 + lane], v1 = reinterpret_cast<const double2 *>(kq)[(t * 2 + 1) * 64 + lane];
        T[t] = (double4_t){v0.x, v0.y, v1.x, v1.y};
    }
    const int a0 = lane, a1 = lane + 64;                                // the vector elements a lane owns
    const bool h0 = a0 < N, h1 = a1 < N;
    __syncthreads();                                                    // the staging is in LDS; the set-up's index arrays are dead
    double ukg, uku, lam0, delta, tau;
    unsigned livem = 0;
    {
        const double kg0 = h0 ? kq[NTILE * 256 + a0] : 0.0, kg1 = h1 ? kq[NTILE * 256 + a1] : 0.0;
        const double ku0 = h0 ? kq[NTILE * 256 + N + a0] : 0.0, ku1 = h1 ? kq[NTILE * 256 + N + a1] : 0.0;
        lam0 = ri_uni(kq[NTILE * 256 + 2 * N]);
        const double tr0 = kq[NTILE * 256 + 2 * N + 1];
        // g -= sum_q u~_q aN_q + a~_q uN_q;  trace of the corrections;  which neurons have a pixel on the ring
        double trc = 0.0, g0v = h0 ? s_gu[1][a0] : 0.0, g1v = h1 ? s_gu[1][a1] : 0.0;
        for (int i = 0; i < nst; ++i) {
            const double uN = s_u[i][N], aN = (double)s_a[i][N];
            const double u0 = h0 ? s_u[i][a0] : 0.0, al0 = h0 ? (double)s_a[i][a0] : 0.0, u1 = h1 ? s_u[i][a1] : 0.0, al1 = h1 ? (double)s_a[i][a1] : 0.0;
            g0v -= fma(u0, aN, al0 * uN); g1v -= fma(u1, aN, al1 * uN);
            trc = fma(u0, al0, fma(u1, al1, trc));
            if (__ballot(al0 != 0.0 || al1 != 0.0) != 0ull) livem |= (1u << i) | (1u << (8 + i));
        }
        if (h0) s_gu[1][a0] = g0v;
        if (h1) s_gu[1][a1] = g1v;
        trc = ri_wave_sum(trc);
        const double lam = ri_uni((tr0 - 2.0 * trc + (double)g.Tp) * 1e-5);
        delta = lam - lam0; tau = (double)g.Tp + lam;
        if (ia.lam_out && lane == 0) ia.lam_out[m] = lam;
        const double uu0 = h0 ? s_gu[0][a0] : 0.0, uu1 = h1 ? s_gu[0][a1] : 0.0;
        // k_g, k_u permuted into LDS (the column products V' k below read a lane's four rows of a block at once)
        if (h0) { s_vec[2][ri_perm(a0)] = kg0; s_vec[3][ri_perm(a0)] = ku0; }
        if (h1) { s_vec[2][ri_perm(a1)] = kg1; s_vec[3][ri_perm(a1)] = ku1; }
        ukg = ri_wave_sum(fma(uu0, kg0, uu1 * kg1)); uku = ri_wave_sum(fma(uu0, ku0, uu1 * ku1));
    }
    livem = (unsigned)__builtin_amdgcn_readfirstlane((int)livem);
    __syncthreads();
    auto leave_to_solve6 = [&](bool rebuild) {
        if (lane == 0) {
            const int i = atomicAdd(ia.fcnt, 1); ia.flist[i] = mi;
            if (rebuild) { const int j = atomicAdd(ia.fcnt + 3, 1); ia.rlist[j] = mi; }
        }
    };
#ifndef RI_CUT
#define RI_CUT 0
#endif
    if ((probe & 32) || RI_CUT == 1) {                                  // (phase probe: set-up, staging and the loads)
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) acc += (T[t][0] + T[t][1]) + (T[t][2] + T[t][3]);
        if (h0 && a0 < p) W[(int64_t)a0 * g.d + m] = (float)(acc + ukg + uku + delta);
        return;
    }
    // ---- Z_I = sum_J K_IJ V_J, H = V' Z on the matrix pipe; h_g = V' k_g, h_u = V' k_u beside them ----
    auto vtile = [&](int J) -> double4_t {                              // V_J: lane (c, rq), register r = V[16 J + rq + 4 r][c]; columns 0-7 A~, 8-15 U~
        double4_t R;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                    // (both reads unconditional, then a select: as a conditional read this compiled into a branch per element)
            const int a = 16 * J + rq + 4 * r;
            const float fa = s_a[c & 7][a]; const double du = s_u[c & 7][a];
            R[r] = c < 8 ? (double)fa : du;
        }
        return R;
    };
    double hg, hu;
    {
        double4_t H = {0.0, 0.0, 0.0, 0.0};
        double hgp = 0.0, hup = 0.0;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            double4_t Z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                const double4_t R = vtile(J);
                if (I == 0) {
                    const double2 k01 = *reinterpret_cast<const double2 *>(&s_vec[2][16 * J + 4 * rq]), k23 = *reinterpret_cast<const double2 *>(&s_vec[2][16 * J + 4 * rq + 2]);
                    const double2 q01 = *reinterpret_cast<const double2 *>(&s_vec[3][16 * J + 4 * rq]), q23 = *reinterpret_cast<const double2 *>(&s_vec[3][16 * J + 4 * rq + 2]);
                    hgp = fma(R[0], k01.x, fma(R[1], k01.y, fma(R[2], k23.x, fma(R[3], k23.y, hgp))));
                    hup = fma(R[0], q01.x, fma(R[1], q01.y, fma(R[2], q23.x, fma(R[3], q23.y, hup))));
                }
                if (J >= I) Z = rs_mfma4(T[rs_tix(J, I)], R, Z);                                  // K_IJ = (tile JI)'
                else Z = rs_mfma4(ri_transpose(T[rs_tix(I, J)], s_blk, c, rq), R, Z);             // K_IJ = tile IJ: its transpose as the first operand
                // (one product at a time: with the LDS reads of a whole row of products hoisted to its top the phase took 118 registers beside the tiles)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            H = rs_mfma4(vtile(I), Z, H);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // H to LDS (row-major 16 x 16), h_g / h_u summed over the row groups
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) s_H[(rq + 4 * r) * 16 + c] = H[r];
        s_red[c * 4 + rq] = hgp; s_red[64 + c * 4 + rq] = hup;
        __syncthreads();
        const double2 r01 = *reinterpret_cast<const double2 *>(s_red + c * 4), r23 = *reinterpret_cast<const double2 *>(s_red + c * 4 + 2);
        const double2 q01 = *reinterpret_cast<const double2 *>(s_red + 64 + c * 4), q23 = *reinterpret_cast<const double2 *>(s_red + 64 + c * 4 + 2);
        hg = (r01.x + r01.y) + (r23.x + r23.y); hu = (q01.x + q01.y) + (q23.x + q23.y);
        __syncthreads();
        if ((probe & 64) || RI_CUT == 2) {                              // (phase probe: ... and the products on the matrix pipe)
            if (h0 && a0 < p) W[(int64_t)a0 * g.d + m] = (float)(hg + hu + H[0]);
            return;
        }
    }
    // ---- the 16-vectors: lane c holds entry c (replicated over rq).  g = g0 - V cg, u = u0 - V cu ----
    // s_red: [0, 16) cg -> coefficients, [16, 32) cu, [32, 48) reference diagonals of the inversion, [48, 64) V'Kg -> s, [64, 80) V'Ku
    const double cgu = s_u[c & 7][N], cga = (double)s_a[c & 7][N], ccs = s_cs[c & 7];
    const double cg = c < 8 ? cgu : cga;
    const double cu = c < 8 ? ccs : 0.0;
    if (rq == 0) { s_red[c] = cg; s_red[16 + c] = cu; s_red[32 + c] = s_H[c * 17]; }
    __syncthreads();
    double VKg, VKu, uKg, uKu;
    {
        double Hcg = 0.0, Hcu = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const double2 h = *reinterpret_cast<const double2 *>(s_H + c * 16 + j);
            Hcg = fma(h.x, s_red[j], fma(h.y, s_red[j + 1], Hcg));
            Hcu = fma(h.x, s_red[16 + j], fma(h.y, s_red[16 + j + 1], Hcu));
        }
        VKg = hg - Hcg; VKu = hu - Hcu;
        uKg = ukg - ri_uni(ri_row_sum(hu * cg - cu * Hcg));          // (sums over the 16 entries: every DPP row holds them all)
        uKu = uku - ri_uni(ri_row_sum(cu * hu - cu * Hcu));
    }
    // ---- cap = S + H, masked, inverted in place ----
    {
        double a4[4];
        const bool mylive = (livem >> c) & 1u;
        const double2 h01 = *reinterpret_cast<const double2 *>(s_H + c * 16 + 4 * rq), h23 = *reinterpret_cast<const double2 *>(s_H + c * 16 + 4 * rq + 2);
        a4[0] = h01.x; a4[1] = h01.y; a4[2] = h23.x; a4[3] = h23.y;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * rq + jj;
            const double v = a4[jj] + (j == (c ^ 8) ? -1.0 : 0.0);
            a4[jj] = (mylive && ((livem >> j) & 1u)) ? v : (j == c ? 1.0 : 0.0);
        }
        int gjbad = 0;
        ri_gj4<0, 8>(a4, c, rq, livem, s_red + 32, gjbad);
        {   // the diagonal the second block starts from
            const int jd = c & 3;
            const double dv = jd == 0 ? a4[0] : (jd == 1 ? a4[1] : (jd == 2 ? a4[2] : a4[3]));
            if (rq == (c >> 2)) s_red[32 + c] = dv;
        }
        __syncthreads();
        ri_gj4<8, 16>(a4, c, rq, livem, s_red + 32, gjbad);
        if (gjbad) { leave_to_solve6(false); return; }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) s_H[c * 16 + 4 * rq + jj] = (mylive && ((livem >> (4 * rq + jj)) & 1u)) ? a4[jj] : 0.0;
        if (rq == 0) { s_red[48 + c] = VKg; s_red[64 + c] = VKu; }
        __syncthreads();
    }
    double yu, uCu, w0;
    {
        double yg = 0.0; yu = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const double2 h = *reinterpret_cast<const double2 *>(s_H + c * 16 + j);
            yg = fma(h.x, s_red[48 + j], fma(h.y, s_red[48 + j + 1], yg));
            yu = fma(h.x, s_red[64 + j], fma(h.y, s_red[64 + j + 1], yu));
        }
        const double uCg = uKg - ri_uni(ri_row_sum(VKu * yg));
        uCu = uKu - ri_uni(ri_row_sum(VKu * yu));
        w0 = (sc - uCg) / (tau - uCu);
        __syncthreads();
        if (rq == 0) s_red[c] = yg - w0 * yu;
        __syncthreads();
    }
    if ((probe & 128) || RI_CUT == 3) {                                 // (phase probe: ... and the small inversion)
        if (h0 && a0 < p) W[(int64_t)a0 * g.d + m] = (float)(w0 + yu + T[0][0] + T[NTILE - 1][3]);
        return;
    }
    // x[a] - sum_j V[a][j] coef[j] for the lane's elements (coef at s_red[0 .. 16))
    auto minus_V = [&](double x0_, double x1_, double &o0, double &o1) {
        for (int i = 0; i < nst; ++i) {
            const double ca = s_red[i], cu_ = s_red[8 + i];
            if (h0) x0_ -= fma((double)s_a[i][a0], ca, s_u[i][a0] * cu_);
            if (h1) x1_ -= fma((double)s_a[i][a1], ca, s_u[i][a1] * cu_);
        }
        o0 = x0_; o1 = x1_;
    };
    auto put_v = [&](double v0, double v1) {
        if (h0) { s_vec[0][a0] = v0; s_vec[1][ri_perm(a0)] = v0; }
        if (h1) { s_vec[0][a1] = v1; s_vec[1][ri_perm(a1)] = v1; }
        __syncthreads();
    };
    // ---- x0 = K (g - w0 u - V (yg - w0 yu)) ----
    double b0, b1;
    minus_V((h0 ? s_gu[1][a0] : 0.0) - w0 * (h0 ? s_gu[0][a0] : 0.0), (h1 ? s_gu[1][a1] : 0.0) - w0 * (h1 ? s_gu[0][a1] : 0.0), b0, b1);
    put_v(b0, b1);
    double xa, xb;
    ri_matvec<NT>(T, s_vec[0], s_vec[1], s_vec[2], nullptr, s_blk, c, rq, xa, xb);
    if (h0) s_gu[1][a0] = xa;                                           // x0 (the slot of g, dead behind the right-hand side)
    if (h1) s_gu[1][a1] = xb;
    int terms = 0;
    bool fail_ = false;
    if (fabs(delta) > 1e-9 * lam0 && !(probe & 16) && RI_CUT != 4) {
        const bool border = uKu > 1e-12 * tau;
        double cu0 = 0.0, cu1 = 0.0;
        if (border) {                                                   // C u = K (u - V yu)
            __syncthreads();
            if (rq == 0) s_red[c] = yu;
            __syncthreads();
            minus_V(h0 ? s_gu[0][a0] : 0.0, h1 ? s_gu[0][a1] : 0.0, b0, b1);
            put_v(b0, b1);
            ri_matvec<NT>(T, s_vec[0], s_vec[1], s_vec[2], nullptr, s_blk, c, rq, cu0, cu1);
        }
        double rprev = 1.0;
        fail_ = true;
#pragma unroll 1
        for (int it = 0; it < ia.maxit; ++it) {
            __syncthreads();
            put_v(xa, xb);
            double t0_, t1_;
            ri_matvec<NT>(T, s_vec[0], s_vec[1], s_vec[2], s_vec[3], s_blk, c, rq, t0_, t1_);             // t = K x (wanted in the permuted order: s_vec[3])
            double sp_ = 0.0;                                                                              // s = V' t
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                const double4_t R = vtile(J);
                const double2 t01 = *reinterpret_cast<const double2 *>(&s_vec[3][16 * J + 4 * rq]), t23 = *reinterpret_cast<const double2 *>(&s_vec[3][16 * J + 4 * rq + 2]);
                sp_ = fma(R[0], t01.x, fma(R[1], t01.y, fma(R[2], t23.x, fma(R[3], t23.y, sp_))));
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            s_red[c * 4 + rq] = sp_;
            __syncthreads();
            double s_;
            { const double2 r01 = *reinterpret_cast<const double2 *>(s_red + c * 4), r23 = *reinterpret_cast<const double2 *>(s_red + c * 4 + 2); s_ = (r01.x + r01.y) + (r23.x + r23.y); }
            __syncthreads();
            if (rq == 0) s_red[64 + c] = s_;
            __syncthreads();
            double y = 0.0;
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const double2 h = *reinterpret_cast<const double2 *>(s_H + c * 16 + j);
                y = fma(h.x, s_red[64 + j], fma(h.y, s_red[64 + j + 1], y));
            }
            if (rq == 0) s_red[c] = y;
            __syncthreads();
            minus_V(xa, xb, b0, b1);                                                                       // x - V capinv V' K x
            put_v(b0, b1);
            double cv0, cv1;
            ri_matvec<NT>(T, s_vec[0], s_vec[1], s_vec[2], nullptr, s_blk, c, rq, cv0, cv1);              // C x
            double w0p = 0.0;
            if (border) w0p = -ri_wave_sum(fma(h0 ? s_gu[0][a0] : 0.0, cv0, (h1 ? s_gu[0][a1] : 0.0) * cv1)) / (tau - uCu);
            const double n0 = (h0 ? s_gu[1][a0] : 0.0) - delta * (cv0 - w0p * cu0), n1 = (h1 ? s_gu[1][a1] : 0.0) - delta * (cv1 - w0p * cu1);
            const double dn = ri_wave_max(fmax(fabs(n0 - xa), fabs(n1 - xb))), xm = ri_wave_max(fmax(fabs(n0), fabs(n1)));
            xa = n0; xb = n1; ++terms;
            const double rr = dn / xm;
            if (!(rr < 1e30)) break;                                                                       // NaN / Inf: left to the factorising kernel
            if (rr * (rr / rprev) < 1e-9 || rr < 1e-9) { fail_ = false; break; }
            rprev = rr;
        }
    }
    {
        const double xm = ri_wave_max(fmax(fabs(xa), fabs(xb)));
        if (!(xm < 1e300)) fail_ = true;
    }
    if (probe & 512) { if (lane == 0) { atomicAdd(ia.fcnt + 1, terms); if (terms) atomicAdd(ia.fcnt + 2, 1); } }
    if (fail_) { leave_to_solve6(true); return; }
    if (h0 && a0 < p) W[(int64_t)a0 * g.d + m] = s_q[a0] >= 0 ? (float)xa : 0.f;
    if (h1 && a1 < p) W[(int64_t)a1 * g.d + m] = s_q[a1] >= 0 ? (float)xb : 0.f;
}

}  // namespace cnmfe
