// B2b on the LAG TABLE: the per-pixel ridge solve of fit_ring_model.m:92-108, PXW neighbouring pixels per workgroup (one wave each).
//
// The block-pair table cov[pair][256][256] serves the SYRK that builds it, but a pixel's 97 x 97 system is a one-pixel-wide curve through it: every
// 8-byte entry sits in a different 128-byte line (22 GB fetched for 3.8 GB of table, 64 lines per wave load -- the gather was the longest phase of
// k_ring_solve5).  k_cov_to_lag (bg.hip) therefore re-lays the table while it applies the footprint corrections:
//     lag[di][q] = Cov(q, q + delta_di),   delta in the canonical half of D = (O - O) u O u -O  (O = ring offsets),  q = block pixel (column-major)
// For a FIXED entry (a, b) of the system, G_m(a, b) = Cov(m + o_a, m + o_b) = lag[di(a, b)][q(m) + anchor(a, b)]: pixels m, m + 1, ... of a column
// read CONSECUTIVE doubles.  A workgroup takes PXW consecutive rows of one patch column; its threads load tile after tile as (entry, pixel) with the
// pixel fastest -- PXW x 8 contiguous bytes per entry -- into an LDS stage in the accumulator-tile order, and wave w picks pixel w's 256 entries up
// as four conflict-free ds_read_b64.  The address table (entry -> lag row, anchor) is the same for every pixel of the patch (host-built, L1-resident).
// Behind the gather: rs_solve_core (ring_solve_core.hpp) unchanged, every wave on its own LDS slices.
#pragma once

namespace cnmfe {

constexpr int64_t RSL_NONE = INT64_MIN;   // address-table entry of a pad position (offsets themselves may be negative: lag row 0 with an anchor left of the centre)
constexpr int RSL_TSTR = 258;      // doubles per (pixel, tile) stage row: 256 + 2, so the 16 lanes of a ds_write_b64 group (2 entries x 8 pixels) hit distinct banks

template <int NT, int PXW>
__global__ void __launch_bounds__(64 * PXW, (NT <= 6 ? 2 : 1))
k_ring_solve_lag(const double *__restrict__ lag, const int64_t *__restrict__ etab, BgGeom g, const int *__restrict__ dr, const int *__restrict__ dc,
                 const double *__restrict__ rowsum, const unsigned char *__restrict__ active, float *__restrict__ W, int *__restrict__ errflag, int probe,
                 int ngr, int ngroups) {
    constexpr int N = 16 * NT, NTILE = (NT * (NT + 1)) / 2;
    __shared__ __attribute__((aligned(16))) double s_stage[2][PXW][RSL_TSTR];
    __shared__ __attribute__((aligned(16))) double s_vec[PXW][3][N];        // u -> z_u, g -> z_g, y
    __shared__ __attribute__((aligned(16))) double s_blk[PXW][16 * RS_DS];
    __shared__ __attribute__((aligned(16))) double s_part[PXW][4][64];
    __shared__ unsigned s_vmask[PXW][4];                                    // bit a: ring node a of pixel w exists (inside the FOV)
    // groups of one XCD are neighbours in the patch (workgroup b runs on XCD b % 8): the two halves of a 128-byte line meet in one L2
    const int b = (int)blockIdx.x;
    const int gidx = (ngroups % 8 == 0) ? (b % 8) * (ngroups / 8) + b / 8 : b;
    const int col = gidx / ngr, row0 = (gidx % ngr) * PXW;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, rq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = g.p;
    // ---- which pixels of the group are fitted (uniform over the workgroup: every wave looks at all PXW) ----
    bool any = false, mine = false;
#pragma unroll
    for (int w = 0; w < PXW; ++w) {
        const int row = row0 + w;
        const bool on = row < g.nr && (!active || active[(int64_t)col * g.nr + row]);
        any |= on; if (w == wave) mine = on;
    }
    if (!any) return;
    const int row = row0 + wave;
    const int64_t m = (int64_t)col * g.nr + (row < g.nr ? row : g.nr - 1);
    const int rbm = (row < g.nr ? row : g.nr - 1) + g.roff, cbm = col + g.coff;      // block coordinates of this wave's centre
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int a = 64 * h + lane;
        bool ok = false;
        if (mine && a < p) {
            const int ra = g.r0_abs + rbm + dr[a], ca = g.c0_abs + cbm + dc[a];
            ok = ra >= 1 && ra <= g.d1 && ca >= 1 && ca <= g.d2;
        }
        const unsigned long long bm = __ballot(ok);
        if (lane == 0) { s_vmask[wave][2 * h] = (unsigned)bm; s_vmask[wave][2 * h + 1] = (unsigned)(bm >> 32); }
    }
    __syncthreads();
    // ---- gather ----
    // loader thread: pixel lp (fastest: PXW consecutive doubles per entry), entry slot es = the accumulator lane (c', rq') the value is for
    const int lp = tid % PXW, es = tid / PXW, ec = es & 15, erq = es >> 4;
    const int64_t q0 = (int64_t)(col + g.coff) * g.nr_b + (row0 + g.roff) + lp;     // block pixel of the loader's centre
    unsigned vl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vl[i] = s_vmask[lp][i];
    int bad = 0;
    // two-stage pipeline: the address-table entries of tile k + 3 and the table values of tile k + 2 are in flight while tile k is handed over
    auto addr = [&](int k, int64_t (&eo)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) eo[r] = etab[(k * 4 + r) * 64 + es];
    };
    auto fetch = [&](int k, const int64_t (&eo)[4], double (&v)[4]) {
        // tile k = (I, J), I >= J, in rs_tix order
        int I = 0; while ((I + 1) * (I + 2) / 2 <= k) ++I;
        const int J = k - I * (I + 1) / 2;
        const int a = 16 * I + ec;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bb = 16 * J + erq + 4 * r;
            const bool ok = ((vl[a >> 5] >> (a & 31)) & (vl[bb >> 5] >> (bb & 31)) & 1u) != 0;
            bad |= (ok && eo[r] == RSL_NONE);
            v[r] = (ok && eo[r] != RSL_NONE && !(probe & 1)) ? lag[eo[r] + q0] : ((I == J && ec == erq + 4 * r) ? 1.0 : 0.0);   // a missing neighbour: identity row (weight 0)
        }
    };
    double4_t T[NTILE];
    {
        double nx[2][4];
        int64_t eo[4];
        addr(0, eo); fetch(0, eo, nx[0]);
        if (NTILE > 1) { addr(1, eo); fetch(1, eo, nx[1]); }
        if (NTILE > 2) addr(2, eo);
#pragma unroll
        for (int k = 0; k < NTILE; ++k) {
            double *st = &s_stage[k & 1][lp][0];
#pragma unroll
            for (int r = 0; r < 4; ++r) st[r * 64 + es] = nx[k & 1][r];
            if (k + 2 < NTILE) fetch(k + 2, eo, nx[k & 1]);
            if (k + 3 < NTILE) addr(k + 3, eo);
            __syncthreads();                                // stage[k & 1] complete; the readers of stage[(k + 1) & 1] (tile k - 1) are past their reads
            const double *sr = &s_stage[k & 1][wave][0];
#pragma unroll
            for (int r = 0; r < 4; ++r) T[k][r] = sr[r * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (bad) atomicOr(errflag, 1);
    if (!mine) return;                                       // (no workgroup barrier below this line)
    // ---- border vectors u, g and the scalar s; trace and ridge ----
    unsigned vmw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vmw[i] = s_vmask[wave][i];
    auto vm = [&](int a) -> bool { return ((vmw[a >> 5] >> (a & 31)) & 1u) != 0; };
    const int64_t qm = (int64_t)cbm * g.nr_b + rbm;
    for (int a = lane; a < N; a += 64) {
        double uv = 0.0, gv = 0.0;
        if (vm(a) && !(probe & 1)) {
            const int rb = rbm + dr[a], cb = cbm + dc[a];
            uv = rowsum[(int64_t)((cb >> 4) * g.nbr + (rb >> 4)) * BLKPX + lp_of(rb & 15, cb & 15)];
            gv = lag[etab[NTILE * 256 + a] + qm];
        }
        s_vec[wave][0][a] = uv; s_vec[wave][1][a] = gv;
    }
    const double sc = rowsum[(int64_t)((cbm >> 4) * g.nbr + (rbm >> 4)) * BLKPX + lp_of(rbm & 15, cbm & 15)];
    double tr = 0.0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c == rq + 4 * r && vm(16 * I + c)) tr += T[rs_tix(I, I)][r];
    tr = rs_wave_sum(tr);
    // ridge: lam = 1e-5 * trace over the real rows, ones row included (fit_ring_model.m:106)
    const double lam = (tr + (double)g.Tp) * 1e-5;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c == rq + 4 * r && vm(16 * I + c)) T[rs_tix(I, I)][r] += lam;
    rs_sync();
    // ---- factorisation + substitutions (ring_solve_core.hpp) ----
    double wc[NT];
    rs_solve_core<NT, false>(T, s_vec[wave], s_blk[wave], s_part[wave], sc, lam, (double)g.Tp, lane, probe, wc);
    // the intercept w0 is discarded (fit_ring_model.m:107); neighbours outside the FOV keep weight 0
    if (rq == 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int a = 16 * k + c;
            if (a < p) W[(int64_t)a * g.d + m] = vm(a) ? (float)wc[k] : 0.f;
        }
    }
}

}  // namespace cnmfe
