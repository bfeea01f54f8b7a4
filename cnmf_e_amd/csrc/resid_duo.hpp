// R1 "duo roles" (r1_variant 14): the ring product of resid_arc.hpp on a 32 x 32 tile with TWO roles.   (included by resid.hip; gfx950 only)
//
// k_residual_arc_dma1 (16 x 32 tile, four roles of 24 offsets) moves 5.6 halo pixels through L2 -> LDS per centre pixel and exchanges three
// partial sums per pixel and chunk.  Here a 512-thread workgroup owns 1024 centres: the VERTICAL role (waves 0-3: four consecutive rows of one
// column per thread) carries the left and right arcs, the HORIZONTAL role (waves 4-7: four consecutive columns of one row) the top and bottom
// arcs -- 48 offsets x 4 centres = 192 weight registers per thread.  Per centre pixel and chunk: 3.75 halo pixels staged (62 x 62 for 32 x 32),
// ONE partial sum exchanged (the vertical role's, which also carries the centre value, so the keeper holds nothing but its own sums), one
// barrier per 1024 pixels instead of per 512.  Same LDS reads per product (0.5) and the same packed FMAs as the four-role kernel.
// LDS: halo[2][62 pieces x 1 KB] + part[2][33 x 32 x 16 B] = 160768 B of 163840.
#pragma once

namespace cnmfe {

constexpr int DUO_T = 32;

template <int R, int ARC_D = 3>
__global__ void __launch_bounds__(512, 2) k_residual_duo(R1Args a) {
    constexpr int P = 4;
    constexpr int TR = DUO_T, TC = DUO_T, NT = 512, NWV = NT / 64;
    constexpr int HR = TR + 2 * R, HC = TC + 2 * R;
    constexpr int HRp = HR + 1;                                        // 63 for R = 15: == -1 (mod 16), see the lane maps below
    static_assert(HRp % 16 == 15 || HRp % 16 == 1, "halo column stride must be +-1 (mod 16)");
    constexpr int NHp = HRp * HC;
    constexpr int NPC = (NHp + 63) / 64, NHs = NPC * 64;                // DMA pieces (one wave instruction = 64 slots) per buffer
    constexpr int NIT = (NPC + NWV - 1) / NWV;
    constexpr int NA = ArcConst<R>::tab.n[0];
    static_assert(ArcConst<R>::tab.n[1] == NA && ArcConst<R>::tab.n[2] == NA && ArcConst<R>::tab.n[3] == NA && NA % 2 == 0, "arcs must be balanced");
    constexpr int NW = NA / 2;
    constexpr int TRp = TR + 1, PARTN = TRp * TC;                      // partial sums [col][row], column stride 33: lane -> column stores hit distinct slots
    extern __shared__ __attribute__((aligned(16))) float4 lds[];      // halo[2][NHs] | part[2][PARTN]
    float4 *halo = lds, *part = lds + 2 * NHs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tmap = a.tile_map[blockIdx.x];
    const int tile_r = tmap & 0xffff, tile_c = tmap >> 16;
    const int hr0 = tile_r * TR + a.roff - R, hc0 = tile_c * TC + a.coff - R;
    const bool horiz = wave >= NWV / 2;                                // wave-uniform role; waves w and w + 4 share a SIMD (dispatch order 0, 2, 1, 3, 0, ...)
    const int rt = tid & 255;
    // vertical:   lane -> column (32 per half wave), half wave -> row group.  A ds_read_b128 lane group holds 16 distinct columns at one row:
    //             slots c * HRp + const == -+c + const (mod 16), distinct.
    // horizontal: lane -> row (32 per half wave), half wave -> column group: 16 distinct consecutive rows of one column per lane group.
    int cr0, cc0, hbase;
    if (!horiz) { const int c = rt & 31, g = rt >> 5; cr0 = g * P; cc0 = c; hbase = c * HRp + g * P; }
    else        { const int r = rt & 31, q = rt >> 5; cr0 = r; cc0 = q * P; hbase = q * P * HRp + r; }
    f2 wa[P][NW], wb[P][NW];                                           // vertical: left / right arc; horizontal: top / bottom arc
    float dl[P]; bool fv[P];                                          // dl: used by the vertical role only (it folds Ymean - b0 into its partial sums)
    const uint32_t fmb0 = (uint32_t)((tile_c * TC + cc0) * a.nr + tile_r * TR + cr0) * 16u;   // horizontal role: byte offset of centre 0 in a Ysig chunk; centre j is j columns on
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int pr = tile_r * TR + cr0 + (horiz ? 0 : j), pc = tile_c * TC + cc0 + (horiz ? j : 0);
        fv[j] = pr < a.nr && pc < a.nc;
        const int64_t m = fv[j] ? (int64_t)pc * a.nr + pr : 0;         // off-patch centres read pixel 0; never stored
        const uint32_t mb = (uint32_t)m * 4u;
        dl[j] = ld_off(a.dlt, mb);
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            int i0, i1, i2, i3;
            if (!horiz) { i0 = ArcConst<R>::tab.ring[0][2 * k]; i1 = ArcConst<R>::tab.ring[0][2 * k + 1]; i2 = ArcConst<R>::tab.ring[1][2 * k]; i3 = ArcConst<R>::tab.ring[1][2 * k + 1]; }
            else        { i0 = ArcConst<R>::tab.ring[2][2 * k]; i1 = ArcConst<R>::tab.ring[2][2 * k + 1]; i2 = ArcConst<R>::tab.ring[3][2 * k]; i3 = ArcConst<R>::tab.ring[3][2 * k + 1]; }
            wa[j][k].x = ld_off(a.W + (int64_t)i0 * a.d, mb);
            wa[j][k].y = ld_off(a.W + (int64_t)i1 * a.d, mb);
            wb[j][k].x = ld_off(a.W + (int64_t)i2 * a.d, mb);
            wb[j][k].y = ld_off(a.W + (int64_t)i3 * a.d, mb);
        }
    }
    // staging plan: piece (j * NWV + wave) of a buffer = 64 consecutive slots of the lane-linear image of [HC][HRp]; slots outside the block fetch
    // a clamped address (their weights are exactly 0)
    uint32_t qoff[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int idx = (j * NWV + wave) * 64 + lane;
        const int hr = idx % HRp, hc = idx / HRp;
        int rb = hr0 + hr, cb = hc0 + (hc < HC ? hc : HC - 1);
        rb = rb < 0 ? 0 : (rb >= a.nr_b ? a.nr_b - 1 : rb);
        cb = cb < 0 ? 0 : (cb >= a.nc_b ? a.nc_b - 1 : cb);
        qoff[j] = (uint32_t)(cb * a.nr_b + rb) * 16u;
    }
    const unsigned ldsA = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float4 *)lds;
    const unsigned lds0 = ldsA + (unsigned)wave * 1024u;
    const unsigned partA = ldsA + (unsigned)(2 * NHs + cc0 * TRp + cr0) * 16u;           // this thread's first centre in part[0]
    const bool full = (tile_r + 1) * TR <= a.nr && (tile_c + 1) * TC <= a.nc;             // workgroup-uniform: no centre of the tile lies outside the patch
    const int64_t cbeg = ((int64_t)blockIdx.y * a.tseg) >> 2;
    const int64_t tend = (int64_t)blockIdx.y * a.tseg + a.tseg < a.T ? (int64_t)blockIdx.y * a.tseg + a.tseg : a.T;
    const int64_t cend = (tend + 3) >> 2;
    const int probe = __builtin_amdgcn_readfirstlane(a.probe);
    auto issue = [&](int64_t c) {
        if ((probe & 1) && c > cbeg + 1) return;
        const int64_t cx = c < cend ? c : cend - 1;
        const float4 *y4 = a.Y4 + cx * a.d_b;
        const int b = (int)((c - cbeg) & 1);
        const unsigned dst = lds0 + (unsigned)b * (unsigned)(NHs * 16);
#pragma unroll
        for (int j = 0; j < NIT; ++j)
            if ((j + 1) * NWV <= NPC || j * NWV + wave < NPC) glds16(y4, qoff[j], dst + (unsigned)(j * NWV) * 1024u);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue(cbeg);
    f2 acc[P][2];                                                     // horizontal role: carried across the barrier (its sums of the previous chunk)
#pragma unroll
    for (int j = 0; j < P; ++j) { acc[j][0] = (f2){0.f, 0.f}; acc[j][1] = (f2){0.f, 0.f}; }
    for (int64_t c = cbeg; c <= cend; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of halo(c) (and the keeper's stores of chunk c-2)
        if (!(probe & 64)) __builtin_amdgcn_s_barrier();              // halo(c) complete; the vertical partial sums of chunk c-1 complete (probe 64: timing without it)
        asm volatile("" ::: "memory");
        if (c < cend) issue(c + 1);                                   // into the buffer of chunk c-1: everybody is done with it
        const int cb_ = (int)((c - cbeg) & 1);
        if (horiz && c > cbeg) {                                      // the keeper finishes chunk c-1
            if (full) {
                // interior tile: four reads in flight, one wait, four unmasked stores (uniform base + the lane's offset; centre j is j columns on)
                const unsigned pa = partA + (unsigned)(cb_ ^ 1) * (unsigned)(PARTN * 16);
                f4v_t p0, p1, p2, p3;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\tds_read_b128 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(pa), "n"(TRp * 16), "n"(2 * TRp * 16), "n"(3 * TRp * 16) : "memory");
                const f4v_t y0 = {-(p0.x + acc[0][0].x), -(p0.y + acc[0][0].y), -(p0.z + acc[0][1].x), -(p0.w + acc[0][1].y)};
                const f4v_t y1 = {-(p1.x + acc[1][0].x), -(p1.y + acc[1][0].y), -(p1.z + acc[1][1].x), -(p1.w + acc[1][1].y)};
                const f4v_t y2 = {-(p2.x + acc[2][0].x), -(p2.y + acc[2][0].y), -(p2.z + acc[2][1].x), -(p2.w + acc[2][1].y)};
                const f4v_t y3 = {-(p3.x + acc[3][0].x), -(p3.y + acc[3][0].y), -(p3.z + acc[3][1].x), -(p3.w + acc[3][1].y)};
                if (!(probe & 8)) {
                    const float4 *ob = a.Ysig4 + (c - 1) * a.d;
                    asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(fmb0), "v"(y0), "s"(ob) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(fmb0), "v"(y1), "s"(ob + a.nr) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(fmb0), "v"(y2), "s"(ob + 2 * a.nr) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(fmb0), "v"(y3), "s"(ob + 3 * a.nr) : "memory");
                }
            } else {
                const float4 *pb = part + (cb_ ^ 1) * PARTN + cc0 * TRp + cr0;
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const float4 p0 = pb[j * TRp];                     // vertical sums - centre value - (Ymean - b0)
                    const float4 yo = make_float4(-(p0.x + acc[j][0].x), -(p0.y + acc[j][0].y), -(p0.z + acc[j][1].x), -(p0.w + acc[j][1].y));
                    if (fv[j] && !(probe & 8)) st4_off(a.Ysig4 + (c - 1) * a.d, fmb0 + (uint32_t)(j * a.nr) * 16u, yo);
                }
            }
        }
        if (c == cend) break;
        const float4 *hb = halo + cb_ * NHs + hbase;
#pragma unroll
        for (int j = 0; j < P; ++j) { acc[j][0] = (f2){0.f, 0.f}; acc[j][1] = (f2){0.f, 0.f}; }
        if (probe & 2) { }
        else if (!horiz) { arc_product<R, 0, P, HRp, NW, ARC_D>(hb, wa, acc); arc_product<R, 1, P, HRp, NW, ARC_D>(hb, wb, acc); }
        else             { arc_product<R, 2, P, HRp, NW, ARC_D>(hb, wa, acc); arc_product<R, 3, P, HRp, NW, ARC_D>(hb, wb, acc); }
        if (!horiz && !(probe & 4)) {
            const unsigned pa = partA + (unsigned)cb_ * (unsigned)(PARTN * 16);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const float4 cv = hb[R * HRp + R + j];                 // (Y - Ymean) at the centre
                const f4v_t o = {acc[j][0].x - (cv.x + dl[j]), acc[j][0].y - (cv.y + dl[j]), acc[j][1].x - (cv.z + dl[j]), acc[j][1].y - (cv.w + dl[j])};
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(pa), "v"(o), "n"(j * 16) : "memory");    // (hipcc splits the float4 store into two ds_write_b64: 2-way bank conflicts at column stride 33)
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // partial sums written before the next barrier
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int R>
static int launch_r1_duo(cnmfe_ctx *ctx, const R1Args &a, int ntile_c, int64_t nseg) {
    constexpr int HR = DUO_T + 2 * R, HC = DUO_T + 2 * R, HRp = HR + 1;
    constexpr int NPC = (HRp * HC + 63) / 64;
    constexpr size_t shmem = (2 * (size_t)NPC * 64 + 2 * (size_t)(DUO_T + 1) * DUO_T) * sizeof(float4);
    static_assert(shmem <= 160 * 1024, "duo kernel exceeds LDS");
    static_assert(((2 * R) * HRp + 2 * R + 4) * 16 < 65536, "ds_read immediate offset overflow");
    dim3 grid((unsigned)((int64_t)a.ntile_r * ntile_c), (unsigned)nseg);
    // measured and not kept (profiles/r03/README.md): an interleaved order of an arc's reads (many-centre reads alternating with one-centre reads: +2 %),
    // s_setprio 1 for either role (keeper: +5 %, vertical: -1 %), two LDS reads in flight instead of three (+5 %; four do not fit 256 VGPRs)
    CK(hipFuncSetAttribute((const void *)k_residual_duo<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    LAUNCH(ctx, "residual_r1", (k_residual_duo<R>), grid, dim3(512), shmem, a);
    return 0;
}

}  // namespace cnmfe
