// B2b: the per-pixel ridge solve of fit_ring_model.m:92-108 -- the kernel: set-up of one pixel's window, the gather of its normal equations
// from the block-pair covariance table into MFMA accumulator tiles, rs_solve_core (ring_solve_core.hpp), the store of the p weights.
#pragma once

namespace cnmfe {

__device__ const double rs_fill[2] = {0.0, 1.0};   // what a load reads for an entry with a missing neighbour: identity rows, no select (and no live mask) behind the load

template <int NT>
__global__ void __launch_bounds__(64, (NT <= 2 ? 4 : (NT <= 3 ? 3 : (NT <= 6 ? 2 : 1))))
k_ring_solve5(CovTab tab, BgGeom g, const int *__restrict__ dr, const int *__restrict__ dc, const double *__restrict__ rowsum,
              const unsigned char *__restrict__ active, float *__restrict__ W, int *__restrict__ errflag, int probe) {
    constexpr int N = 16 * NT, NTILE = (NT * (NT + 1)) / 2;
    __shared__ int s_node[N + 1];                                       // (window block << 8 | local pixel) of ring neighbour a, [N] = the centre; -1: none
    __shared__ int s_pt[256];                                           // block-pair codes of the window
    __shared__ __attribute__((aligned(16))) double s_vec[3][N];         // u -> z_u, g -> z_g, y
    __shared__ __attribute__((aligned(16))) double s_blk[16 * RS_DS];
    __shared__ __attribute__((aligned(16))) double s_part[4][64];
    const int64_t m = blockIdx.x;
    if (active && !active[m]) return;
    const int lane = threadIdx.x, c = lane & 15, rq = lane >> 4;
    const int p = g.p;
    const int rbm = (int)(m % g.nr) + g.roff, cbm = (int)(m / g.nr) + g.coff;
    const int br0 = (rbm - g.p_radius) >> 4, bc0 = (cbm - g.p_radius) >> 4;      // arithmetic shift: floor
    // the window is at most 4 x 4 blocks (host check): window block = wr + 4 wc whatever its real width, so every index below is a shift
    for (int a = lane; a <= N; a += 64) {
        int code = -1;
        if (a < p || a == N) {
            const int rb = a < p ? rbm + dr[a] : rbm, cb = a < p ? cbm + dc[a] : cbm;
            const int ra = g.r0_abs + rb, ca = g.c0_abs + cb;
            if (ra >= 1 && ra <= g.d1 && ca >= 1 && ca <= g.d2) code = ((((rb >> 4) - br0) + 4 * ((cb >> 4) - bc0)) << 8) | lp_of(rb & 15, cb & 15);
        }
        s_node[a] = code;
    }
    for (int q = lane; q < 256; q += 64) {
        const int a = q >> 4, b = q & 15;
        int ia = br0 + (a & 3), ja = bc0 + (a >> 2), ib = br0 + (b & 3), jb = bc0 + (b >> 2);
        int code = -1;
        if (ia >= 0 && ja >= 0 && ib >= 0 && jb >= 0 && ia < g.nbr && ib < g.nbr && ja < g.nbc && jb < g.nbc) {
            int dR = ib - ia, dC = jb - ja, sw = 0;
            if (dC < 0 || (dC == 0 && dR < 0)) { sw = 1; ia = ib; ja = jb; dR = -dR; dC = -dC; }
            if (dC <= tab.maxd && dR <= tab.maxd && dR >= -tab.maxd) {
                const int pidx = tab.pair_of[(ja * tab.nbr + ia) * tab.nrel + rel_index(dR, dC, tab.maxd)];
                code = pidx < 0 ? -1 : ((pidx << 2) | (sw << 1) | ((dR == 0 && dC == 0) ? 1 : 0));
            }
        }
        s_pt[q] = code;
    }
    __syncthreads();
    int bad = 0;
    // address of Cov(node na, node nb) in the table (both nodes exist)
    auto cov_ptr = [&](int na, int nb) -> const double * {
        // (self pairs: upper patch triangle only.  Mirroring them in the table to drop this case distinction was measured: 0.7 ms SLOWER at
        //  512 x 512 -- both orientations of a pair then miss the cache separately)
        const int code = s_pt[((na >> 8) << 4) + (nb >> 8)];
        bad |= code < 0;
        const int la = na & 255, lb = nb & 255;
        const bool sw = (code & 2) != 0;
        const bool flip = sw != (((code & 1) != 0) && ((sw ? lb : la) >> 4) > ((sw ? la : lb) >> 4));
        const int x = flip ? lb : la, y = flip ? la : lb;
        return tab.cov + ((int64_t)(code < 0 ? 0 : code >> 2) * BLKPX + x) * BLKPX + y;
    };
    // ---- border vectors u, g and the scalar s ----
    for (int a = lane; a < N; a += 64) {
        const int na = s_node[a], nc = s_node[N];
        double uv = 0.0, gv = 0.0;
        if (na >= 0 && !(probe & 1)) {
            const int ab = (((na >> 8) & 3) + br0) + ((na >> 10) + bc0) * g.nbr;
            uv = rowsum[(int64_t)ab * BLKPX + (na & 255)];
            gv = *cov_ptr(na, nc);
        }
        s_vec[0][a] = uv; s_vec[1][a] = gv;
    }
    const double sc = rowsum[(int64_t)((cbm >> 4) * g.nbr + (rbm >> 4)) * BLKPX + lp_of(rbm & 15, cbm & 15)];
    // ---- gather: tile (I, J), lane (c, rq), register r  =  G[16 I + c][16 J + rq + 4 r] ----
    double4_t T[NTILE];
    double tr = 0.0;
    {
        int nodeA[NT];
#pragma unroll
        for (int I = 0; I < NT; ++I) nodeA[I] = (probe & 1) ? -1 : s_node[16 * I + c];
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int na = nodeA[I], nb = s_node[16 * J + rq + 4 * r];
                    const double *ptr = (na >= 0 && nb >= 0) ? cov_ptr(na, nb) : &rs_fill[(I == J && c == rq + 4 * r) ? 1 : 0];
                    T[rs_tix(I, J)][r] = *ptr;
                    if (r == 3) __builtin_amdgcn_sched_barrier(0);      // addresses are formed tile by tile: hoisting all of them ahead of the loads costs 2 VGPRs per entry
                }
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (c == rq + 4 * r && nodeA[I] >= 0) tr += T[rs_tix(I, I)][r];
        tr = rs_wave_sum(tr);
        // ridge: lam = 1e-5 * trace over the real rows, ones row included (fit_ring_model.m:106)
        const double lam0 = (tr + (double)g.Tp) * 1e-5;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (c == rq + 4 * r && nodeA[I] >= 0) T[rs_tix(I, I)][r] += lam0;
    }
    const double lam = (tr + (double)g.Tp) * 1e-5;
    if (bad) atomicOr(errflag, 1);
    __syncthreads();
    // ---- factorisation + substitutions (ring_solve_core.hpp) ----
    double wc[NT];
    rs_solve_core<NT>(T, s_vec, s_blk, s_part, sc, lam, (double)g.Tp, lane, probe, wc);
    // the intercept w0 is discarded (fit_ring_model.m:107); neighbours outside the FOV keep weight 0
    if (rq == 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int a = 16 * k + c;
            if (a < p) W[(int64_t)a * g.d + m] = s_node[a] >= 0 ? (float)wc[k] : 0.f;
        }
    }
}

}  // namespace cnmfe
