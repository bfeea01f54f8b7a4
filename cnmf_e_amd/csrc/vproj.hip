// The projections of a VIRTUAL residual (round 4): what the spatial and the temporal update need of
//     Ysig = Y(patch) - b0 - W (R - mean R),  R = Y_block - A_prev C_prev          (update_spatial_parallel.m:162-166 == update_temporal_parallel.m:149-152)
// without ever forming Ysig.  With the centred video Yc = Y - Ymean (resident) and dlt = Ymean(patch) - b0:
//     Ysig = Yc(patch) + dlt - W Yc_block + (W A_prev)(C_prev - mean)             (the last term: the "footprint term", kept pending beside Ysig, resid.hip)
// * spatial update (HALS_spatial.m:31, U = Ysig C' - T mean(Ysig) mean(C)' = Ysig Cc', Cc the centred traces: constants drop out):
//       U(m,k) = P(m,k) - sum_i W(m,i) P(m + o_i, k),        P = Yc Cc'  (block pixels x neurons, fp64)
//   P is what the ring fit's window projection computes anyway (bg.hip, k_win_proj4 / k_win_fix) in the same iteration with the same traces: the fit leaves
//   it with the patch and the update costs nnz(IND) (p + 1) table reads.  When no valid table is there (traces changed since the fit, a frame stride != 1 in the
//   fit, masks that reach beyond the fit's lists) the table is built here by the same kernel -- one read of the video.
// * temporal update (HALS_temporal.m:48, U = A' Ysig):
//       U(k,t) = sum_j B(j,k) Yc(j,t) + sum_m A(m,k) dlt(m),   B = E A - W' A   (E: patch rows -> block rows; support: footprint (+) ring)
//   a block-tiled projection of the centred video on the fp64 matrix pipe: per 16x16 block of pixels the neurons whose B meets it (<= 64), contraction over
//   the block's 256 pixels, partial sums per (block, neuron) added in a fixed order (bit-reproducible).  One read of the video.
// Everything is accumulated in fp64 from exact fp32 products, so the cancellation between the A and W'A parts of B (the background they remove) costs nothing.
// Round 5: with the video's digit planes resident the same contraction runs on the int8 matrix pipe (vproj_i8.hpp: exact int32 sums of 32-bit fixed-point operands,
// the partial sums still fp64); the fp64 kernel below serves the patches without planes.
// The pending footprint term enters both updates through their projections as before (residual_term_fold_spatial / residual_term_project, resid.hip).
#include "common.hpp"
#include "win_proj.hpp"
#include "vproj_i8.hpp"
#include <climits>

namespace cnmfe {

static BgGeom vp_geom(const Patch *P) {
    BgGeom g{};
    g.nr = P->nr; g.nc = P->nc; g.nr_b = P->nr_b; g.nc_b = P->nc_b; g.roff = P->roff; g.coff = P->coff;
    g.r0_abs = P->brect[0]; g.c0_abs = P->brect[2]; g.d1 = P->d1; g.d2 = P->d2;
    g.nbr = (P->nr_b + BLK - 1) / BLK; g.nbc = (P->nc_b + BLK - 1) / BLK;
    g.d = P->d; g.d_b = P->d_b; g.T = P->T; g.kstride = 1; g.Tp = P->T; g.Tpad = (P->T + 15) / 16 * 16;
    g.p = P->p; g.p_radius = 0;
    for (int i = 0; i < P->p; ++i) g.p_radius = std::max(g.p_radius, std::max(std::abs(P->dr[i]), std::abs(P->dc[i])));
    g.nbw = ((2 * g.p_radius) >> 4) + 2; g.bf4 = 1;
    return g;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// spatial:  U(e) = P(m_e, k_e) - sum_i W(m_e, i) P(m_e + o_i, k_e)
// ------------------------------------------------------------------------------------------------------------------------------------------
// sum over the frame segments of the window projection's partial tables (k_win_proj4 writes one per segment)
__global__ void __launch_bounds__(256) k_vp_segsum(const double *__restrict__ Ut, int64_t ut_stride, int nseg, int64_t n, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = Ut[i];
    for (int sg = 1; sg < nseg; ++sg) v += Ut[sg * ut_stride + i];
    out[i] = v;
}

// One thread per entry of the search mask (entries of a column are consecutive pixels: for a fixed ring offset the lanes of a wave read consecutive weights and,
// mostly, consecutive table entries).  Eight offsets at a time: their loads are independent; the sum keeps the ring order (bit-reproducible).
__global__ void __launch_bounds__(256) k_vp_spatial(const int *__restrict__ erow, const int *__restrict__ ecol, int64_t nnz, BgGeom g, const int *__restrict__ dr,
                                                    const int *__restrict__ dc, const float *__restrict__ W, const double *__restrict__ tab,
                                                    const int *__restrict__ lp, const short *__restrict__ slot_of, int Kt, const int *__restrict__ kmap,
                                                    float *__restrict__ U, double *__restrict__ Q, int *__restrict__ err) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const int m = erow[e], kt = kmap[ecol[e]];
    const int rbm = m % g.nr + g.roff, cbm = m / g.nr + g.coff;
    bool bad = false;
    auto at = [&](int rb, int cb) -> int64_t {              // index of P(pixel, kt) in the table, -1 if the table does not hold it
        const int b = (cb >> 4) * g.nbr + (rb >> 4);
        const int sl = slot_of[(int64_t)b * Kt + kt];
        return sl < 0 ? -1 : ((int64_t)lp[b] + sl) * BLKPX + lp_of(rb & 15, cb & 15);
    };
    double acc = 0.0;
    for (int i0 = 0; i0 < g.p; i0 += 8) {
        float w8[8]; int64_t ix[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u < g.p ? i0 + u : g.p - 1;
            const int rb = rbm + dr[i], cb = cbm + dc[i];
            const bool in = i0 + u < g.p && rb >= 0 && rb < g.nr_b && cb >= 0 && cb < g.nc_b;
            w8[u] = in ? W[(int64_t)i * g.d + m] : 0.f;     // (a weight is 0 where the neighbour is outside the field of view)
            ix[u] = (in && w8[u] != 0.f) ? at(rb, cb) : 0;
            if (ix[u] < 0) { bad = true; ix[u] = 0; w8[u] = 0.f; }
        }
        double p8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p8[u] = tab[ix[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (w8[u] != 0.f) acc += (double)w8[u] * p8[u];
    }
    if (Q) Q[e] = acc;                                      // (bg_ssub: the ring sum alone, fp64 -- upsampled by k_vp_ssub_combine)
    else {
        const int64_t i0 = at(rbm, cbm);
        if (i0 < 0) bad = true;
        U[e] = (float)((i0 < 0 ? 0.0 : tab[i0]) - acc);
    }
    if (bad) atomicOr(err, 1);                             // (the host checked the coverage: an inconsistent table, reported by the next wait)
}

// which blocks of the block region the entries of a CSC column (patch rows) reach when every pixel is grown by R: per block column bj the block rows lo..hi
// (exact for masks whose image columns are runs; a superset otherwise).  Appends (block) to `out`.
static void reach_blocks(const Patch *P, const BgGeom &g, int R, const int32_t *rowidx, int64_t e0, int64_t e1, std::vector<int> &lo, std::vector<int> &hi, std::vector<int> &out) {
    int bjmin = INT_MAX, bjmax = -1;
    // the entries of a CSC column ascend with the pixel index, i.e. image column by image column: one update per IMAGE COLUMN (its first and last stored row)
    // instead of one per entry -- this list building sat in front of the temporal projection's first launch with the GPU idle (0.45 ms at 500 neurons)
    auto flush = [&](int cb, int rfirst, int rlast) {
        const int bi0 = std::max(0, rfirst - R) >> 4, bi1 = std::min(P->nr_b - 1, rlast + R) >> 4;
        const int bj0 = std::max(0, cb - R) >> 4, bj1 = std::min(P->nc_b - 1, cb + R) >> 4;
        for (int bj = bj0; bj <= bj1; ++bj) {
            if (hi[bj] < 0) { lo[bj] = bi0; hi[bj] = bi1; } else { lo[bj] = std::min(lo[bj], bi0); hi[bj] = std::max(hi[bj], bi1); }
        }
        bjmin = std::min(bjmin, bj0); bjmax = std::max(bjmax, bj1);
    };
    int64_t e = e0;
    while (e < e1) {
        const int m0 = rowidx[e];
        const int c = m0 / P->nr;                                            // image column of this run (patch coordinates)
        const int col_end = (c + 1) * P->nr;
        int mlast = m0;
        ++e;
        while (e < e1 && rowidx[e] < col_end && rowidx[e] >= mlast) { mlast = rowidx[e]; ++e; }   // (an unsorted column just starts a new run: still a superset)
        flush(c + P->coff, m0 - c * P->nr + P->roff, mlast - c * P->nr + P->roff);
    }
    for (int bj = bjmin; bj <= bjmax; ++bj) {
        if (hi[bj] < 0) continue;
        for (int bi = lo[bj]; bi <= hi[bj]; ++bi) out.push_back(bj * g.nbr + bi);
        hi[bj] = -1;
    }
}

int vproj_spatial(cnmfe_ctx *ctx, Patch *P, int32_t K, const float *C, int c_order, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                  const int *dErow, const int *dEcol, const float *dCc, int64_t ldc, float *dU, double *dQ) {
    HostTrace ht(ctx, "vproj_spatial");
    const BgGeom g = vp_geom(P);
    const int nblk = g.nbr * g.nbc, R = g.p_radius;
    const int64_t nnz = IND_colptr[K];
    if ((int64_t)nblk * K >= (int64_t(1) << 31)) return 1;
    // the (block, neuron) pairs the update reads: the mask of k grown by the ring
    std::vector<int> lo(g.nbc, 0), hi(g.nbc, -1), need_ptr((size_t)K + 1, 0), need;
    need.reserve((size_t)K * 16);
    for (int32_t k = 0; k < K; ++k) {
        reach_blocks(P, g, R, IND_rowidx, IND_colptr[k], IND_colptr[k + 1], lo, hi, need);
        need_ptr[k + 1] = (int)need.size();
    }
    // does the patch's table hold them?  (the fit's window projection left it: same video, traces = rows of the bound matrix of this generation)
    std::vector<int32_t> rows;
    const bool have_rows = bound_rows_of(ctx, C, c_order, K, rows);
    std::vector<int> kmap((size_t)K, -1);
    bool reuse = P->pt_valid && have_rows && P->pt_gen == ctx->bound_gen && P->pt_K > 0 && (int)P->pt_lp_h.size() == nblk + 1;
    if (reuse) {
        std::vector<int> colof((size_t)ctx->bound_K, -1);
        for (int32_t c = 0; c < P->pt_K; ++c) colof[P->pt_rows[c]] = c;
        for (int32_t k = 0; k < K && reuse; ++k) {
            if (need_ptr[k + 1] == need_ptr[k]) { kmap[k] = 0; continue; }        // (an empty mask reads nothing)
            const int c = colof[rows[k]];
            if (c < 0) { reuse = false; break; }
            kmap[k] = c;
            for (int i = need_ptr[k]; i < need_ptr[k + 1]; ++i)
                if (P->pt_slot_h[(size_t)need[i] * P->pt_K + c] < 0) { reuse = false; break; }
        }
    }
    ht.mark(reuse ? "need lists (table of the fit)" : "need lists (table to build)");
    if (!reuse) {
        // the table of exactly these pairs, by the fit's own kernel: lists per block, longest first
        std::vector<int> cnt((size_t)nblk + 1, 0);
        for (int b : need) ++cnt[b + 1];
        std::vector<int> lst_ptr((size_t)nblk + 1, 0);
        for (int b = 0; b < nblk; ++b) {
            if (cnt[b + 1] > WIN_NLB) return 1;            // denser than the window kernel is built for: the sweep serves this update
            lst_ptr[b + 1] = lst_ptr[b] + cnt[b + 1];
        }
        std::vector<int> lst_k(need.size()), fill((size_t)nblk, 0), blk_nt[4], blall;
        std::vector<short> slot_of((size_t)nblk * K, (short)-1);
        for (int32_t k = 0; k < K; ++k)
            for (int i = need_ptr[k]; i < need_ptr[k + 1]; ++i) {
                const int b = need[i], s_ = fill[b]++;
                lst_k[lst_ptr[b] + s_] = k; slot_of[(size_t)b * K + k] = (short)s_;
            }
        for (int b = 0; b < nblk; ++b) { const int n = lst_ptr[b + 1] - lst_ptr[b]; if (n) blk_nt[(n - 1) >> 4].push_back(b); }
        for (int t = 3; t >= 0; --t) blall.insert(blall.end(), blk_nt[t].begin(), blk_nt[t].end());
        DevBuf &dLk = ctx->inc[1], &dBl = ctx->inc[3], &dUt = ctx->inc[4];
        P->pt_valid = false;
        RET(to_dev(ctx, P->pt_lp, lst_ptr.data(), lst_ptr.size()));
        RET(to_dev(ctx, P->pt_slot, slot_of.data(), slot_of.size()));
        RET(to_dev(ctx, dLk, lst_k.data(), lst_k.size()));
        RET(to_dev(ctx, dBl, blall.data(), blall.size()));
        const int nb_ = (int)blall.size();
        const int64_t nent = (int64_t)std::max<size_t>(1, lst_k.size()) * BLKPX;
        RET(P->pt_tab.ensure((size_t)nent * sizeof(double)));
        if (nb_ > 0) {
            int nsg = nb_ >= 512 ? std::max(1, std::min(8, (2048 + nb_ - 1) / nb_)) : std::max(1, std::min(16, (4096 + nb_ - 1) / std::max(1, nb_)));
            // the int8 kernel sums a frame segment in int32 (4 digit pairs x 2^14 per frame): at most I8_SEG_FRAMES frames per segment -- a longer recording gets more segments
            const bool i8_tab = !P->derived && P->dig_valid && P->T <= (int64_t)I8_SEG_FRAMES * 16 && ctx->opt("win_i8", 1) != 0;
            if (i8_tab) nsg = std::max(nsg, (int)((P->T + I8_SEG_FRAMES - 1) / I8_SEG_FRAMES));
            RET(dUt.ensure((size_t)nsg * nent * sizeof(double)));
            const int nbig = (int)blk_nt[3].size();
            if (i8_tab) {
                // round 6: the table on the int8 pipe out of the resident digit planes, by the fit's own kernel (win_proj_i8.hpp) -- what a recording whose fit runs on
                // every other frame needs every iteration (the fit's table covers ITS frames only: BASELINE configs[4]), and any patch whose fit left no table
                RET(win_i8_table(ctx, P, "spatial_trace_dig", "spatial_ptab_proj", K, dCc, ldc, lst_ptr, blall, P->pt_lp.as<int>(), dLk.as<int>(), nsg, dUt.as<double>(), nent));
            } else {
            if (nbig)
                LAUNCH(ctx, "spatial_ptab_proj", k_win_proj4<true>, dim3((unsigned)(nbig * nsg)), dim3(256), 0, P->Yc4.as<float4>(), g, dCc, ldc, P->pt_lp.as<int>(), dLk.as<int>(),
                       dBl.as<int>(), nsg, dUt.as<double>(), nent, (double *)nullptr, (int64_t)0);
            if (nb_ > nbig)
                LAUNCH(ctx, "spatial_ptab_proj", k_win_proj4<false>, dim3((unsigned)((nb_ - nbig) * nsg)), dim3(256), 0, P->Yc4.as<float4>(), g, dCc, ldc, P->pt_lp.as<int>(),
                       dLk.as<int>(), dBl.as<int>() + nbig, nsg, dUt.as<double>(), nent, (double *)nullptr, (int64_t)0);
            }
            LAUNCH(ctx, "spatial_ptab_sum", k_vp_segsum, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, dUt.as<double>(), nent, nsg, nent, P->pt_tab.as<double>());
        }
        P->pt_K = K; P->pt_lp_h.swap(lst_ptr); P->pt_slot_h.swap(slot_of);
        P->pt_rows = rows; P->pt_gen = have_rows ? ctx->bound_gen : -1;
        P->pt_valid = true;
        for (int32_t k = 0; k < K; ++k) kmap[k] = k;
        ht.mark("table built");
    }
    if (nnz == 0) return 0;
    DevBuf &dKmap = ctx->vp[0];
    RET(to_dev(ctx, dKmap, kmap.data(), kmap.size()));
    int *dErr = nullptr;
    RET(ctx_errflag(ctx, &dErr));
    LAUNCH(ctx, "spatial_from_ptab", k_vp_spatial, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, dErow, dEcol, nnz, g, P->ring_dr.as<int>(), P->ring_dc.as<int>(),
           P->W.as<float>(), P->pt_tab.as<double>(), P->pt_lp.as<int>(), P->pt_slot.as<short>(), (int)P->pt_K, dKmap.as<int>(), dU, dQ, dErr);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// temporal:  U(k,t) = sum_j B(j,k) Yc(j,t) + sum_m A(m,k) dlt(m),   B = E A - W' A
// ------------------------------------------------------------------------------------------------------------------------------------------
// B of one (block, neuron) entry: 256 threads = the block's pixels in memory order (px = r + 16 c).  The neuron's footprint inside the block grown by the ring
// radius sits in LDS as a dense (16 + 2R)^2 window; B(j) = A(j) - sum_i W(j - o_i, i) A(j - o_i) then costs p LDS reads and a weight load per non-zero hit.
// Table layout: Bt[((g16[b] + slot / 16) * 256 + px) * 16 + slot % 16] -- per block and group of 16 list slots a [pixel][16] panel, what the projection's
// A operand reads with one conflict-free LDS access per lane.
#ifndef VP_AH
#define VP_AH 8
#endif
#ifndef VP_WG_TARGET
#define VP_WG_TARGET 2048
#endif
constexpr int VP_WS = 64;                                   // window side: 16 + 2 * 24
__global__ void __launch_bounds__(256) k_vp_build_b(const int *__restrict__ ent_blk, const int *__restrict__ ent_k, const int *__restrict__ ent_slot, const int *__restrict__ g16,
                                                    BgGeom g, int R, const int64_t *__restrict__ colptr, const int *__restrict__ erow, const float *__restrict__ aval,
                                                    const int *__restrict__ dr, const int *__restrict__ dc, const float *__restrict__ W, double *__restrict__ Bt) {
    __shared__ float win[VP_WS * VP_WS];
    const int b = ent_blk[blockIdx.x], k = ent_k[blockIdx.x], slot = ent_slot[blockIdx.x];
    const int bi = b % g.nbr, bj = b / g.nbr;
    const int wr0 = bi * 16 - R, wc0 = bj * 16 - R, ws = 16 + 2 * R;
    for (int i = threadIdx.x; i < ws * ws; i += 256) win[i] = 0.f;
    __syncthreads();
    for (int64_t e = colptr[k] + threadIdx.x; e < colptr[k + 1]; e += 256) {
        const int m = erow[e];
        const int rw = m % g.nr + g.roff - wr0, cw = m / g.nr + g.coff - wc0;
        if (rw >= 0 && rw < ws && cw >= 0 && cw < ws) win[cw * ws + rw] = aval[e];
    }
    __syncthreads();
    const int px = threadIdx.x, rb = bi * 16 + (px & 15), cb = bj * 16 + (px >> 4);
    double v = 0.0;
    if (rb < g.nr_b && cb < g.nc_b) {
        double acc = 0.0;
        for (int i = 0; i < g.p; ++i) {
            const int rm = rb - dr[i], cm = cb - dc[i];                     // the centre pixel whose i-th ring neighbour is this pixel
            const float a = win[(cm - wc0) * ws + (rm - wr0)];
            if (a != 0.f) acc += (double)W[(int64_t)i * g.d + (int64_t)(cm - g.coff) * g.nr + (rm - g.roff)] * (double)a;      // (a != 0: a patch pixel)
        }
        v = (double)win[(cb - wc0) * ws + (rb - wr0)] - acc;
    }
    Bt[(((int64_t)g16[b] + (slot >> 4)) * BLKPX + px) * 16 + (slot & 15)] = v;
}

// cst[k] = sum_m A(m,k) (Ymean(m) - b0(m)) in fp64, one workgroup per neuron, fixed association
__global__ void __launch_bounds__(256) k_vp_const(const int64_t *__restrict__ colptr, const int *__restrict__ erow, const float *__restrict__ aval, BgGeom g,
                                                  const double *__restrict__ ymean, const double *__restrict__ b0, double *__restrict__ cst) {
    __shared__ double red[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int64_t e = colptr[k] + threadIdx.x; e < colptr[k + 1]; e += 256) {
        const int m = erow[e];
        const int64_t q = (int64_t)(m / g.nr + g.coff) * g.nr_b + (m % g.nr + g.roff);
        s += (double)aval[e] * (ymean[q] - b0[m]);
    }
    red[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) cst[k] = red[0];
}

// The projection.  Workgroup = (block, frame segment), 4 waves; a wave takes chunk GROUPS of 16 chunks (64 frames) and contracts over the block's 256 pixels:
//   v_mfma_f64_16x16x4:  A operand [M = list slot (lane & 15)][K = pixel (lane >> 4)]  <- the B panel in LDS
//                        B operand [K = pixel (lane >> 4)][N = chunk s + (lane & 15)]   <- lane loads the float4 of ITS pixel and chunk: component j feeds MFMA j,
//                                                                                         whose column n is frame 4 (s + n) + j
//   64 steps of 4 pixels; a wave load moves 16 chunks x 4 consecutive pixels x 16 B (64-byte runs, every byte used).
// D: lane holds rows (lane >> 4) + 4 r = list slots, column lane & 15 = its chunk: the four components are four consecutive frames -> one 32-byte store per
// (slot, chunk) into the partial buffer part[(l0 + slot) * ldp + 4 chunk ..].
// the centred video once more, in the ORDER the projection below reads it: Yt4[((blk * ncg + cg) * 64 + st) * 64 + lane] = chunk 16 cg + (lane & 15) of pixel
// 4 st + (lane >> 4) of the block (row + 16 column inside the block; 0 outside the block region or behind the last chunk).  A wave instruction of the projection
// wants 4 consecutive pixels of 16 chunks: 16 segments of 64 bytes, 4 MB apart in the frame-major video -- here ONE kilobyte of consecutive addresses
// (round 5; + one video's worth of HBM, kept only when that leaves 8 GB free)
__global__ void __launch_bounds__(256) k_tile_video(const float4 *__restrict__ Y4, BgGeom g, int64_t Tc, int64_t ncg, float4 *__restrict__ Yt4) {
    const int blk = blockIdx.x, bi = blk % g.nbr, bj = blk / g.nbr;
    const int64_t cg = blockIdx.y;
    const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    const int64_t cl = cg * 16 + n;
    float4 *o = Yt4 + ((int64_t)blk * ncg + cg) * 64 * 64 + lane;
    for (int st = threadIdx.x >> 6; st < 64; st += 4) {
        const int px = 4 * st + kq, rb = bi * 16 + (px & 15), cb = bj * 16 + (px >> 4);
        const bool in = rb < g.nr_b && cb < g.nc_b && cl < Tc;
        o[st * 64] = in ? Y4[cl * g.d_b + (int64_t)cb * g.nr_b + rb] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int NT, bool TILED>
__global__ void __launch_bounds__(256) k_vp_proj_b(const float4 *__restrict__ Y4, BgGeom g, int64_t Tc, const int *__restrict__ blk_list, const int *__restrict__ lst_ptr,
                                                   const int *__restrict__ g16, const double *__restrict__ Bt, int nseg, double *__restrict__ part, int64_t ldp) {
    extern __shared__ __attribute__((aligned(16))) double Bl[];        // [NT][256][16]
    const int blk = blk_list[blockIdx.x / nseg], seg = blockIdx.x % nseg;
    {
        const double2 *src = reinterpret_cast<const double2 *>(Bt + (int64_t)g16[blk] * BLKPX * 16);
        double2 *dst = reinterpret_cast<double2 *>(Bl);
        for (int i = threadIdx.x; i < NT * BLKPX * 8; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int l0 = lst_ptr[blk], nl = lst_ptr[blk + 1] - l0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
    const int bi = blk % g.nbr, bj = blk / g.nbr;
    // pixel of step st for this lane: px = 4 st + kq -> row (px & 15), column (px >> 4) of the block; clamped at the region's edge (B is 0 there)
    const int rb0 = bi * 16, cb0 = bj * 16;
    const int64_t ncg = (Tc + 15) >> 4;
    const int64_t cgs = (ncg + nseg - 1) / nseg, cg0 = seg * cgs, cg1 = cg0 + cgs < ncg ? cg0 + cgs : ncg;
    for (int64_t cg = cg0 + wave; cg < cg1; cg += 4) {
        const int64_t cl = cg * 16 + n, clc = cl < Tc ? cl : Tc - 1;
        const float4 *yb = TILED ? Y4 + (((int64_t)blk * ncg + cg) * 64) * 64 + lane : Y4 + clc * g.d_b;
        double4_t acc[4][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[j][t] = (double4_t){0.0, 0.0, 0.0, 0.0};
        auto qof = [&](int st) -> int64_t {
            const int px = 4 * st + kq;
            if constexpr (TILED) return (int64_t)st * 64;
            int rb = rb0 + (px & 15), cb = cb0 + (px >> 4);
            rb = rb < g.nr_b ? rb : g.nr_b - 1; cb = cb < g.nc_b ? cb : g.nc_b - 1;
            return (int64_t)cb * g.nr_b + rb;
        };
        constexpr int AH = VP_AH;                          // loads in flight per lane
        float4 y[AH];
#pragma unroll
        for (int u = 0; u < AH; ++u) y[u] = yb[qof(u)];
#pragma unroll 1
        for (int st0 = 0; st0 < 64; st0 += AH) {
            // the NEXT eight steps' loads go out before this iteration's MFMAs (the compiler sinks them behind the MFMAs otherwise: no prefetch distance at all);
            // the last iteration repeats step 63 -- no branch in the loop
            float4 yn[AH];
#pragma unroll
            for (int u = 0; u < AH; ++u) yn[u] = yb[qof(st0 + AH + u < 64 ? st0 + AH + u : 63)];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < AH; ++u) {
                const int px = 4 * (st0 + u) + kq;
                const double b0 = (double)y[u].x, b1 = (double)y[u].y, b2 = (double)y[u].z, b3 = (double)y[u].w;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double a = Bl[(t * BLKPX + px) * 16 + n];
#ifdef CNMFE_PROBE_NOMFMA                                       // timing probe (scripts/build_variant.py): the loads stay, the matrix work goes -- results are garbage
                    acc[0][t][0] = fma(a, b0, acc[0][t][0]); acc[1][t][0] = fma(a, b1, acc[1][t][0]); acc[2][t][0] = fma(a, b2, acc[2][t][0]); acc[3][t][0] = fma(a, b3, acc[3][t][0]);
#else
                    acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[1][t], 0, 0, 0);
                    acc[2][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[2][t], 0, 0, 0);
                    acc[3][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[3][t], 0, 0, 0);
#endif
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < AH; ++u) y[u] = yn[u];
        }
        if (cl < Tc) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = t * 16 + kq + 4 * r;
                    if (slot < nl) {
                        double *o = part + (int64_t)(l0 + slot) * ldp + 4 * cl;
                        *reinterpret_cast<double2 *>(o) = make_double2(acc[0][t][r], acc[1][t][r]);
                        *reinterpret_cast<double2 *>(o + 2) = make_double2(acc[2][t][r], acc[3][t][r]);
                    }
                }
        }
    }
}

// U(k, t) = float(cst[k] + sum over the (block, k) entries of k, in ascending block order, of their partial sums); 0 past T
__global__ void __launch_bounds__(256) k_vp_reduce(const double *__restrict__ part, int64_t ldp, const int *__restrict__ nptr, const int *__restrict__ nent,
                                                   const double *__restrict__ cst, int64_t T, float *__restrict__ U, int64_t ldu) {
    const int k = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ldu) return;
    double s = cst[k];
    for (int i = nptr[k]; i < nptr[k + 1]; ++i) s += part[(int64_t)nent[i] * ldp + t];
    U[(int64_t)k * ldu + t] = t < T ? (float)s : 0.f;
}

// ---- the host side of a block-tiled projection, shared by the full-resolution form and bg_ssub's low-resolution one ----
struct VpLists {
    int64_t nent = 0;
    std::vector<int> lst_ptr, g16, ent_blk, ent_k, ent_slot, kent, kptr, blk_nt[4], blall;
};
// need[need_ptr[k] .. need_ptr[k + 1]): the blocks neuron k's B column meets.  Entries in (block, slot) order; per neuron its entries in ascending block order
// (the order of the final sum), followed by `extra_row0 + k` when extra_row0 >= 0 (a partial row of its own per neuron, bg_ssub).  1: a block meets more than
// WIN_NLB neurons
static int vp_make_lists(const std::vector<int> &need_ptr, const std::vector<int> &need, int nblk, int32_t K, bool extra, VpLists &L) {
    L.nent = (int64_t)need.size();
    std::vector<int> cnt((size_t)nblk + 1, 0);
    for (int b : need) ++cnt[b + 1];
    L.lst_ptr.assign((size_t)nblk + 1, 0); L.g16.assign((size_t)nblk + 1, 0);
    for (int b = 0; b < nblk; ++b) {
        if (cnt[b + 1] > WIN_NLB) return 1;                // more than 64 neurons over one block: the sweep serves this update
        L.lst_ptr[b + 1] = L.lst_ptr[b] + cnt[b + 1];
        L.g16[b + 1] = L.g16[b] + ((cnt[b + 1] + 15) >> 4);
    }
    L.ent_blk.resize((size_t)L.nent); L.ent_k.resize((size_t)L.nent); L.ent_slot.resize((size_t)L.nent);
    std::vector<int> fill((size_t)nblk, 0), kent((size_t)L.nent);
    for (int32_t k = 0; k < K; ++k)
        for (int i = need_ptr[k]; i < need_ptr[k + 1]; ++i) {
            const int b = need[i], s_ = fill[b]++, e = L.lst_ptr[b] + s_;
            L.ent_blk[e] = b; L.ent_k[e] = k; L.ent_slot[e] = s_; kent[i] = e;
        }
    for (int32_t k = 0; k < K; ++k) std::sort(kent.begin() + need_ptr[k], kent.begin() + need_ptr[k + 1]);     // (entry index ascends with the block)
    if (!extra) { L.kent.swap(kent); L.kptr = need_ptr; }
    else {
        L.kent.clear(); L.kent.reserve(kent.size() + K); L.kptr.assign((size_t)K + 1, 0);
        for (int32_t k = 0; k < K; ++k) {
            L.kent.insert(L.kent.end(), kent.begin() + need_ptr[k], kent.begin() + need_ptr[k + 1]);
            L.kent.push_back((int)L.nent + k);
            L.kptr[k + 1] = (int)L.kent.size();
        }
    }
    for (int t = 0; t < 4; ++t) L.blk_nt[t].clear();
    for (int b = 0; b < nblk; ++b) { const int n = L.lst_ptr[b + 1] - L.lst_ptr[b]; if (n) L.blk_nt[(n - 1) >> 4].push_back(b); }
    L.blall.clear();
    for (int t = 3; t >= 0; --t) L.blall.insert(L.blall.end(), L.blk_nt[t].begin(), L.blk_nt[t].end());
    return 0;
}
static int vp_upload_lists(cnmfe_ctx *ctx, const VpLists &L) {
    DevBuf *V = ctx->vp;
    RET(to_dev(ctx, V[1], L.ent_blk.data(), L.ent_blk.size())); RET(to_dev(ctx, V[2], L.ent_k.data(), L.ent_k.size())); RET(to_dev(ctx, V[3], L.ent_slot.data(), L.ent_slot.size()));
    RET(to_dev(ctx, V[4], L.g16.data(), L.g16.size())); RET(to_dev(ctx, V[5], L.lst_ptr.data(), L.lst_ptr.size()));
    RET(to_dev(ctx, V[9], L.kptr.data(), L.kptr.size())); RET(to_dev(ctx, V[10], L.kent.data(), L.kent.size()));
    RET(to_dev(ctx, V[11], L.blall.data(), L.blall.size()));
    return 0;
}
// the projection launches: the video of patch V (its centred frames, or its read-order copy when `tiled`) against the B panels in ctx->vp[6], partial rows into ctx->vp[8]
static int vp_launch_proj(cnmfe_ctx *ctx, Patch *V, const BgGeom &g, const VpLists &L, bool tiled, int64_t ldp) {
    DevBuf *B = ctx->vp;
    DevBuf &dG16 = B[4], &dLp = B[5], &dBt = B[6], &dPart = B[8], &dBl = B[11];
    int off = 0;
    for (int t = 3; t >= 0; --t) {
        const int nb_ = (int)L.blk_nt[t].size();
        if (!nb_) continue;
        const int total = (int)L.blall.size();
        // frame segments: enough workgroups to fill the chip a few times over, each at least a few chunk groups per wave
        const int64_t ncg = ((V->Tc + 15) >> 4);
        int nsg = (int)std::max<int64_t>(1, std::min<int64_t>((ncg + 7) / 8, (VP_WG_TARGET + total - 1) / std::max(1, total)));
        const size_t shmem = (size_t)(t + 1) * BLKPX * 16 * sizeof(double);
#define VP_GO(NT_) do { if (shmem > 64 * 1024) { CK(hipFuncSetAttribute((const void *)k_vp_proj_b<NT_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
                                                     CK(hipFuncSetAttribute((const void *)k_vp_proj_b<NT_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); } \
            if (tiled) LAUNCH(ctx, "temporal_proj_B", (k_vp_proj_b<NT_, true>), dim3((unsigned)(nb_ * nsg)), dim3(256), shmem, V->yt4.as<float4>(), g, V->Tc, dBl.as<int>() + off, dLp.as<int>(), \
                   dG16.as<int>(), dBt.as<double>(), nsg, dPart.as<double>(), ldp); \
            else LAUNCH(ctx, "temporal_proj_B", (k_vp_proj_b<NT_, false>), dim3((unsigned)(nb_ * nsg)), dim3(256), shmem, V->Yc4.as<float4>(), g, V->Tc, dBl.as<int>() + off, dLp.as<int>(), \
                   dG16.as<int>(), dBt.as<double>(), nsg, dPart.as<double>(), ldp); } while (0)
        if (t == 0) VP_GO(1); else if (t == 1) VP_GO(2); else if (t == 2) VP_GO(3); else VP_GO(4);
#undef VP_GO
        off += nb_;
    }
    return 0;
}

int vproj_temporal(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                   const int64_t *dColptr, const int *dErow, const float *dAval, float *dU, int64_t ldu) {
    HostTrace ht(ctx, "vproj_temporal");
    const BgGeom g = vp_geom(P);
    const int nblk = g.nbr * g.nbc, R = g.p_radius;
    if (16 + 2 * R > VP_WS) return 1;
    (void)A_val;
    // the blocks B(:, k) meets: the footprint grown by the ring
    std::vector<int> lo(g.nbc, 0), hi(g.nbc, -1), need_ptr((size_t)K + 1, 0), need;
    need.reserve((size_t)K * 16);
    for (int32_t k = 0; k < K; ++k) {
        reach_blocks(P, g, R, A_rowidx, A_colptr[k], A_colptr[k + 1], lo, hi, need);
        need_ptr[k + 1] = (int)need.size();
    }
    VpLists L;
    if (vp_make_lists(need_ptr, need, nblk, K, false, L)) return 1;
    const int64_t nent = L.nent;
    const int64_t ldp = ldu;                               // partial sums: one row of ldu doubles per entry
    if (nent * ldp * 8 > (int64_t(24) << 30)) return 1;    // (a partial buffer beyond 24 GB: not what this path is for)
    ht.mark("lists");
    DevBuf *V = ctx->vp;
    DevBuf &dEb = V[1], &dEk = V[2], &dEs = V[3], &dG16 = V[4], &dBt = V[6], &dCst = V[7], &dPart = V[8], &dNptr = V[9], &dNent = V[10];
    RET(vp_upload_lists(ctx, L));
    const size_t bt_bytes = (size_t)std::max(1, L.g16[nblk]) * BLKPX * 16 * sizeof(double);
    RET(dBt.ensure_hw(bt_bytes, ctx->hw_vp[6]));
    RET(dCst.ensure_hw((size_t)K * sizeof(double), ctx->hw_vp[7]));
    RET(dPart.ensure_hw((size_t)std::max<int64_t>(1, nent) * ldp * sizeof(double), ctx->hw_vp[8]));
    CK(hipMemsetAsync(dBt.p, 0, bt_bytes, ctx->st()));      // (the padding slots of a group must be 0)
    if (nent > 0)
        LAUNCH(ctx, "temporal_build_B", k_vp_build_b, dim3((unsigned)nent), dim3(256), 0, dEb.as<int>(), dEk.as<int>(), dEs.as<int>(), dG16.as<int>(), g, R, dColptr, dErow, dAval,
               P->ring_dr.as<int>(), P->ring_dc.as<int>(), P->W.as<float>(), dBt.as<double>());
    // round 5 (late): on the int8 pipe out of the video's digit planes when the fit left them resident (vproj_i8.hpp) -- their pixel-major copy is made once per upload
    bool i8 = !P->derived && P->dig_valid && ctx->opt("proj_i8", 1) != 0 && nent > 0 && P->dig_T16 * 16 >= ldu;
    if (i8 && !P->digp_valid) {
        const size_t dbytes = (size_t)nblk * P->dig_T16 * 16 * 64 * sizeof(uint4);
        if (P->digp.cap < dbytes) { size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); i8 = fr >= dbytes + ((size_t)8 << 30); }
        if (i8) {
            RET(P->digp.ensure(dbytes));
            LAUNCH(ctx, "temporal_dig_pixmajor", k_dig_pixmajor, dim3((unsigned)nblk, (unsigned)P->dig_T16), dim3(256), 0, P->dig.as<uint4>(), P->dig_T16, P->digp.as<uint4>());
            P->digp_valid = true;
        }
    }
    LAUNCH(ctx, "temporal_const", k_vp_const, dim3((unsigned)K), dim3(256), 0, dColptr, dErow, dAval, g, P->ymean_d.as<double>(), P->b0.as<double>(), dCst.as<double>());
    if (i8) {
        const int ngrp = L.g16[nblk];
        std::vector<int> grp_blk((size_t)std::max(1, ngrp), 0);
        for (int b = 0; b < nblk; ++b) for (int q = L.g16[b]; q < L.g16[b + 1]; ++q) grp_blk[q] = b;
        DevBuf &dGb = V[24], &dBd = V[25], &dBs = V[26];
        RET(to_dev(ctx, dGb, grp_blk.data(), grp_blk.size()));
        RET(dBd.ensure_hw((size_t)std::max(1, ngrp) * 1024 * sizeof(uint4), ctx->hw_vp[25]));
        RET(dBs.ensure_hw((size_t)std::max(1, ngrp) * 16 * sizeof(double), ctx->hw_vp[26]));
        LAUNCH(ctx, "temporal_panel_dig", k_vp_bdig, dim3((unsigned)ngrp), dim3(256), 0, dBt.as<double>(), dGb.as<int>(), P->dig_sc.as<double>(), dBd.as<uint4>(), dBs.as<double>());
        const int64_t T16 = P->dig_T16;
        int off = 0;
        for (int t = 3; t >= 0; --t) {
            const int nb_ = (int)L.blk_nt[t].size();
            if (!nb_) continue;
            const int total = (int)L.blall.size();
            const int nsg = (int)std::max<int64_t>(1, std::min<int64_t>((T16 + 7) / 8, (VP_WG_TARGET + total - 1) / std::max(1, total)));
            const size_t shmem = (size_t)(t + 1) * 1024 * sizeof(uint4);
#define VP_GO8(NT_, V0_) LAUNCH(ctx, "temporal_proj_B", (k_vp_proj_i8<NT_, V0_>), dim3((unsigned)(nb_ * nsg)), dim3(256), shmem, P->digp.as<uint4>(), T16, ctx->vp[11].as<int>() + off, \
                           ctx->vp[5].as<int>(), ctx->vp[4].as<int>(), dBd.as<uint4>(), dBs.as<double>(), nsg, dPart.as<double>(), ldp)
            if (ctx->opt("proj_i8_planes", 3) >= 4) { if (t == 0) VP_GO8(1, 0); else if (t == 1) VP_GO8(2, 0); else if (t == 2) VP_GO8(3, 0); else VP_GO8(4, 0); }
            else { if (t == 0) VP_GO8(1, 1); else if (t == 1) VP_GO8(2, 1); else if (t == 2) VP_GO8(3, 1); else VP_GO8(4, 1); }
#undef VP_GO8
            off += nb_;
        }
    } else {
    // the block-tiled copy of the centred video (k_tile_video), built at the first projection of a patch when the memory allows
    bool tiled = false;
    if (!P->derived && ctx->opt("proj_tiled", 1) != 0) {
        const int64_t ncg_t = (P->Tc + 15) >> 4;
        const size_t ybytes = (size_t)nblk * ncg_t * 64 * 64 * sizeof(float4);
        if (!P->yt4_valid) {
            bool ok = true;
            if (P->yt4.cap < ybytes) { size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); ok = fr >= ybytes + ((size_t)8 << 30); }
            if (ok) {
                RET(P->yt4.ensure(ybytes));
                LAUNCH(ctx, "temporal_tile_video", k_tile_video, dim3((unsigned)nblk, (unsigned)ncg_t), dim3(256), 0, P->Yc4.as<float4>(), g, P->Tc, ncg_t, P->yt4.as<float4>());
                P->yt4_valid = true;
            }
        }
        tiled = P->yt4_valid;
    }
    RET(vp_launch_proj(ctx, P, g, L, tiled, ldp));
    }
    LAUNCH(ctx, "temporal_reduce_B", k_vp_reduce, dim3((unsigned)((ldu + 255) / 256), (unsigned)K), dim3(256), 0, dPart.as<double>(), ldp, dNptr.as<int>(), dNent.as<int>(),
           dCst.as<double>(), P->T, dU, ldu);
    ht.mark("launches");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// bg_ssub > 1 (round 5): the same two projections THROUGH the resampling maps.  With D = imresize(., 1/s) (bicubic, antialiased: the low-resolution residual
// patch R holds D Yc as its resident video), Up = imresize(., [nr_b nc_b]) (bicubic) and W_L the ring weights on the low-resolution grid,
//     Ysig = Yc(patch) + dlt - Up W_L D Yc + [footprint term: full-resolution ELL rows of up(W_L down(A_prev)), pending as before]
//     (update_spatial_parallel.m:167-178 == update_temporal_parallel.m:153-165)
// * spatial:   U(m,k) = P_F(m,k) - sum_{l in taps(m)} up(m,l) Q_L(l,k),   P_F = Yc Cc' on the mask's entries (one read of the video rows under the masks, fp64 sums),
//                                                                        Q_L(l,k) = sum_i W_L(l,i) P_L(l + o_i, k),  P_L = (D Yc) Cc': vproj_spatial's table of R
//   Q_L on the low-resolution bounding box of every mask's taps (dense per neuron: the upsampling looks its 4 x 4 taps up by position)
// * temporal:  A' Ysig = A' Yc + A' dlt 1' - B_L' (D Yc),   B_L = W_L' (Up' A): the block-tiled projection of R's video with panels built from up' A, plus the
//   footprint rows' own fp64 projection of the full-resolution video as one more partial row per neuron
// Both replace the low-resolution sweep, its upsample (a 10.5 GB read + a 10.5 GB write at H) and the projections of the realised Ysig.
// ------------------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_vp_rows_block(const int *__restrict__ erow, int64_t nnz, int nr, int nr_b, int roff, int coff, int *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const int m = erow[e];
    out[e] = (m / nr + coff) * nr_b + (m % nr + roff);
}
// P_F(e) = sum_t Yc(q_e, t) Cc(k_e, t): k_proj_spatial (factor.hip) on the centred VIDEO with fp64 sums -- the background these sums still contain is removed by the
// upsampled ring term afterwards, so the cancellation must not cost digits
__global__ void __launch_bounds__(256) k_vp_rows_spatial(const float4 *__restrict__ Y4, int64_t d_b, int64_t T, const int *__restrict__ erq, const int *__restrict__ ecol, int64_t nnz,
                                                         const float *__restrict__ Cc, int64_t ldc, int64_t tchunk, double *__restrict__ part) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const int64_t t0 = (int64_t)blockIdx.y * tchunk, t1 = t0 + tchunk < T ? t0 + tchunk : T;      // tchunk is a multiple of 4
    const float4 *y = Y4 + erq[e];
    const float *c = Cc + (int64_t)ecol[e] * ldc;            // centred traces are 0 beyond T
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int64_t t = t0; t < t1; t += 4) {
        const float4 yv = y[(t >> 2) * d_b];
        const float4 cv = *reinterpret_cast<const float4 *>(c + t);
        a0 = fma((double)yv.x, (double)cv.x, a0); a1 = fma((double)yv.y, (double)cv.y, a1); a2 = fma((double)yv.z, (double)cv.z, a2); a3 = fma((double)yv.w, (double)cv.w, a3);
    }
    part[(int64_t)blockIdx.y * nnz + e] = (a0 + a1) + (a2 + a3);
}
// U(e) = sum of the frame parts of P_F(e), in order, - sum over the upsampling taps of pixel m_e of w_c w_r Q_L(tap, k_e);  bb[k] = (first low row, first low column,
// rows, offset) of neuron k's dense low-resolution box in Q
__global__ void __launch_bounds__(256) k_vp_ssub_combine(const int *__restrict__ erow, const int *__restrict__ ecol, int64_t nnz, int nr, int roff, int coff,
                                                         const double *__restrict__ part, int nparts, const int *__restrict__ ir, const float *__restrict__ wr, int Pr,
                                                         const int *__restrict__ ic, const float *__restrict__ wc, int Pc, const int4 *__restrict__ bb,
                                                         const double *__restrict__ Q, float *__restrict__ U) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const int m = erow[e], k = ecol[e];
    const int rb = m % nr + roff, cb = m / nr + coff;
    double pf = 0.0;
    for (int j = 0; j < nparts; ++j) pf += part[(int64_t)j * nnz + e];
    const int4 b = bb[k];
    double acc = 0.0;
    for (int u = 0; u < Pc; ++u) {
        const float wcu = wc[cb * Pc + u];
        if (wcu == 0.f) continue;
        const int64_t cofs = (int64_t)b.w + (int64_t)(ic[cb * Pc + u] - b.y) * b.z - b.x;
        for (int q = 0; q < Pr; ++q) {
            const float w = wr[rb * Pr + q];
            if (w == 0.f) continue;
            acc = fma((double)wcu * (double)w, Q[cofs + ir[rb * Pr + q]], acc);
        }
    }
    U[e] = (float)(pf - acc);
}

// every image-column run of a CSC column over patch rows (ascending): f(block-region column, first block-region row, last one)
template <typename F>
static void for_column_runs(const Patch *P, const int32_t *rowidx, int64_t e0, int64_t e1, F f) {
    int64_t e = e0;
    while (e < e1) {
        const int m0 = rowidx[e];
        const int c = m0 / P->nr, col_end = (c + 1) * P->nr;
        int mlast = m0;
        ++e;
        while (e < e1 && rowidx[e] < col_end && rowidx[e] >= mlast) { mlast = rowidx[e]; ++e; }
        f(c + P->coff, m0 - c * P->nr + P->roff, mlast - c * P->nr + P->roff);
    }
}
// low-resolution bounding box [r0, r1] x [c0, c1] of the upsampling taps of a column's pixels (r1 < r0: the column is empty)
static void low_box(const Patch *M, const int32_t *rowidx, int64_t e0, int64_t e1, int &r0, int &r1, int &c0, int &c1) {
    r0 = c0 = INT_MAX; r1 = c1 = -1;
    for_column_runs(M, rowidx, e0, e1, [&](int cb, int rf, int rl) {
        r0 = std::min(r0, M->ss_rlo[rf]); r1 = std::max(r1, M->ss_rhi[rl]);
        c0 = std::min(c0, M->ss_clo[cb]); c1 = std::max(c1, M->ss_chi[cb]);
    });
}

int vproj_spatial_ssub(cnmfe_ctx *ctx, Patch *M, int32_t K, const float *C, int c_order, const int64_t *IND_colptr, const int32_t *IND_rowidx,
                       const int *dErow, const int *dEcol, const float *dCc, int64_t ldc, float *dU) {
    HostTrace ht(ctx, "vproj_spatial_ssub");
    Patch *R = get_patch(ctx, M->ss_res);
    if (!R || !R->ring_ready) return 1;
    RET(ssub_taps(ctx, M, R));
    const int64_t nnz = IND_colptr[K];
    if (nnz == 0) return 0;
    const int d1s = M->ss_d1s;
    // the low-resolution mask: per neuron the dense box of its mask's taps
    std::vector<int64_t> lcp((size_t)K + 1, 0);
    std::vector<int> bb((size_t)K * 4, 0);
    for (int32_t k = 0; k < K; ++k) {
        int r0, r1, c0, c1;
        low_box(M, IND_rowidx, IND_colptr[k], IND_colptr[k + 1], r0, r1, c0, c1);
        const int h = r1 >= r0 ? r1 - r0 + 1 : 0, w = r1 >= r0 ? c1 - c0 + 1 : 0;
        bb[(size_t)k * 4] = h ? r0 : 0; bb[(size_t)k * 4 + 1] = h ? c0 : 0; bb[(size_t)k * 4 + 2] = h; bb[(size_t)k * 4 + 3] = (int)lcp[k];
        lcp[k + 1] = lcp[k] + (int64_t)h * w;
    }
    const int64_t nl = lcp[K];
    if (nl >= (int64_t(1) << 31)) return 1;
    std::vector<int32_t> lrow((size_t)nl), lcol((size_t)nl);
    for (int32_t k = 0; k < K; ++k) {
        const int r0 = bb[(size_t)k * 4], c0 = bb[(size_t)k * 4 + 1], h = bb[(size_t)k * 4 + 2];
        int64_t o = lcp[k];
        const int w = h ? (int)((lcp[k + 1] - lcp[k]) / h) : 0;
        for (int c = 0; c < w; ++c)
            for (int r = 0; r < h; ++r, ++o) { lrow[o] = (c0 + c) * d1s + (r0 + r); lcol[o] = k; }
    }
    ht.mark("low-resolution masks");
    DevBuf *V = ctx->vp;
    DevBuf &dLr = V[16], &dLc = V[17], &dQ = V[18], &dBB = V[19], &dPart = V[20], &dErq = V[21];
    RET(to_dev(ctx, dLr, lrow.data(), lrow.size())); RET(to_dev(ctx, dLc, lcol.data(), lcol.size())); RET(to_dev(ctx, dBB, bb.data(), bb.size()));
    RET(dQ.ensure_hw((size_t)nl * sizeof(double), ctx->hw_vp[18]));
    // Q_L: the ring sums of R's table on the boxes (the table P_L = (D Yc) Cc' is built by vproj_spatial: one read of the low-resolution video)
    const int rc = vproj_spatial(ctx, R, K, C, c_order, lcp.data(), lrow.data(), dLr.as<int>(), dLc.as<int>(), dCc, ldc, nullptr, dQ.as<double>());
    if (rc) return rc;
    // P_F: the rows of the centred video under the masks
    const int64_t T = M->T;
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(32, T / 256));
    const int64_t tchunk = ((T + nparts - 1) / nparts + 3) & ~int64_t(3);
    RET(dPart.ensure_hw((size_t)nparts * nnz * sizeof(double), ctx->hw_vp[20]));
    const int *erq = dErow;
    if (M->nr != M->nr_b || M->roff || M->coff) {
        RET(dErq.ensure_hw((size_t)nnz * sizeof(int), ctx->hw_vp[21]));
        LAUNCH(ctx, "ssub_rows_block", k_vp_rows_block, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, dErow, nnz, M->nr, M->nr_b, M->roff, M->coff, dErq.as<int>());
        erq = dErq.as<int>();
    }
    LAUNCH(ctx, "spatial_proj_rows", k_vp_rows_spatial, dim3((unsigned)((nnz + 255) / 256), (unsigned)nparts), dim3(256), 0, M->Yc4.as<float4>(), M->d_b, T, erq, dEcol, nnz, dCc, ldc,
           tchunk, dPart.as<double>());
    LAUNCH(ctx, "spatial_ssub_combine", k_vp_ssub_combine, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, dErow, dEcol, nnz, M->nr, M->roff, M->coff, dPart.as<double>(), nparts,
           M->ss_ir.as<int>(), M->ss_wr.as<float>(), M->ss_Pr, M->ss_ic.as<int>(), M->ss_wc.as<float>(), M->ss_Pc, reinterpret_cast<const int4 *>(dBB.p), dQ.as<double>(), dU);
    ht.mark("launches");
    return 0;
}

// B_L of one (low-resolution block, neuron) entry: B_L = -W_L' (Up' A).  The footprint's entries around the block go into a dense full-resolution window, up' A on the
// block's ring-grown low-resolution window is gathered from it through the TRANSPOSED taps -- rows first, then columns (separable; fixed order: bit-reproducible) --,
// the transposed ring product as in k_vp_build_b.  span_r[2 bi], span_r[2 bi + 1] / span_c: the full-resolution rows / columns the window of block row bi / block column
// bj reaches (host).  Dynamic LDS: winL[ws * ws] + mid[ws * fmax] doubles, winF[fmax * fmax] floats.
__global__ void __launch_bounds__(256) k_vp_build_b_ssub(const int *__restrict__ ent_blk, const int *__restrict__ ent_k, const int *__restrict__ ent_slot, const int *__restrict__ g16,
                                                         BgGeom g, int R, int fmax, const int *__restrict__ span_r, const int *__restrict__ span_c,
                                                         const int64_t *__restrict__ colptr, const int *__restrict__ erq, const float *__restrict__ aval, int nr_bF,
                                                         const int *__restrict__ trp, const int *__restrict__ tri, const float *__restrict__ trw,
                                                         const int *__restrict__ tcp, const int *__restrict__ tci, const float *__restrict__ tcw,
                                                         const int *__restrict__ dr, const int *__restrict__ dc, const float *__restrict__ W, double *__restrict__ Bt) {
    extern __shared__ __attribute__((aligned(16))) double vb_lds[];
    const int ws = 16 + 2 * R;
    double *winL = vb_lds, *mid = vb_lds + ws * ws;
    float *winF = reinterpret_cast<float *>(mid + ws * fmax);
    const int b = ent_blk[blockIdx.x], k = ent_k[blockIdx.x], slot = ent_slot[blockIdx.x];
    const int bi = b % g.nbr, bj = b / g.nbr;
    const int wr0 = bi * 16 - R, wc0 = bj * 16 - R;
    const int fr0 = span_r[2 * bi], fh = span_r[2 * bi + 1] - fr0 + 1, fc0 = span_c[2 * bj], fw = span_c[2 * bj + 1] - fc0 + 1;      // (<= fmax)
    for (int i = threadIdx.x; i < fh * fw; i += 256) winF[i] = 0.f;
    __syncthreads();
    for (int64_t e = colptr[k] + threadIdx.x; e < colptr[k + 1]; e += 256) {
        const int q = erq[e];
        const int rw = q % nr_bF - fr0, cw = q / nr_bF - fc0;
        if (rw >= 0 && rw < fh && cw >= 0 && cw < fw) winF[cw * fh + rw] = aval[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ws * fw; i += 256) {       // rows: mid[cF][jr] = sum over the rows low row jr feeds
        const int jr = wr0 + i % ws, cF = i / ws;
        double s = 0.0;
        if (jr >= 0 && jr < g.nr_b) {
            const float *col = winF + cF * fh - fr0;
            for (int t = trp[jr]; t < trp[jr + 1]; ++t) s = fma((double)trw[t], (double)col[tri[t]], s);
        }
        mid[cF * ws + i % ws] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ws * ws; i += 256) {       // columns
        const int jc = wc0 + i / ws;
        double a = 0.0;
        if (jc >= 0 && jc < g.nc_b)
            for (int t = tcp[jc]; t < tcp[jc + 1]; ++t) a = fma((double)tcw[t], mid[(tci[t] - fc0) * ws + i % ws], a);
        winL[i] = a;
    }
    __syncthreads();
    const int px = threadIdx.x, rb = bi * 16 + (px & 15), cb = bj * 16 + (px >> 4);
    double v = 0.0;
    if (rb < g.nr_b && cb < g.nc_b) {
        double acc = 0.0;
        for (int i = 0; i < g.p; ++i) {
            const int rm = rb - dr[i], cm = cb - dc[i];                     // the centre pixel whose i-th ring neighbour is this pixel
            const double a = winL[(cm - wc0) * ws + (rm - wr0)];
            if (a != 0.0) acc += (double)W[(int64_t)i * g.d + (int64_t)(cm - g.coff) * g.nr + (rm - g.roff)] * a;      // (a != 0: a pixel of the low-resolution grid)
        }
        v = -acc;
    }
    Bt[(((int64_t)g16[b] + (slot >> 4)) * BLKPX + px) * 16 + (slot & 15)] = v;
}
// the footprint rows' own projection, part[row0 + k][t] = sum_e A(e) Yc(q_e, t) in fp64: k_proj_temporal (factor.hip) on the centred video
__global__ void __launch_bounds__(256) k_vp_rows_temporal(const float4 *__restrict__ Y4, int64_t d_b, int64_t T, const int64_t *__restrict__ colptr, const int *__restrict__ erq,
                                                          const float *__restrict__ aval, int64_t cchunk, double *__restrict__ part, int64_t ldp) {
    const int k = blockIdx.x;
    const int64_t e0 = colptr[k], e1 = colptr[k + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t Tc = (T + 3) >> 2;
    const int64_t c0 = (int64_t)blockIdx.y * cchunk, c1 = c0 + cchunk < Tc ? c0 + cchunk : Tc;
    constexpr int NE = 6;
    const int nit = (int)(e1 - e0 + 63 < (int64_t)64 * NE ? (e1 - e0 + 63) >> 6 : NE);
    int rw[NE]; float av[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int64_t e = e0 + lane + 64 * i;
        const bool in = i < nit && e < e1;
        rw[i] = in ? erq[e] : 0; av[i] = in ? aval[e] : 0.f;
    }
    const int64_t et = e0 + (int64_t)64 * NE;
    for (int64_t c = c0 + wave; c < c1; c += 4) {
        const float4 *y = Y4 + c * d_b;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (i < nit) {
                const float4 yv = y[rw[i]];
                const double a = (double)av[i];
                s0 = fma(a, (double)yv.x, s0); s1 = fma(a, (double)yv.y, s1); s2 = fma(a, (double)yv.z, s2); s3 = fma(a, (double)yv.w, s3);
            }
        for (int64_t e = et + lane; e < e1; e += 64) {
            const double a = (double)aval[e]; const float4 yv = y[erq[e]];
            s0 = fma(a, (double)yv.x, s0); s1 = fma(a, (double)yv.y, s1); s2 = fma(a, (double)yv.z, s2); s3 = fma(a, (double)yv.w, s3);
        }
        for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o); }
        if (lane == 0) {
            double *u = part + (int64_t)k * ldp + 4 * c;     // ldp is a multiple of 4 >= T
            u[0] = s0; u[1] = s1; u[2] = s2; u[3] = s3;
        }
    }
}

int vproj_temporal_ssub(cnmfe_ctx *ctx, Patch *M, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                        const int64_t *dColptr, const int *dErow, const float *dAval, float *dU, int64_t ldu) {
    HostTrace ht(ctx, "vproj_temporal_ssub");
    (void)A_val;
    Patch *R = get_patch(ctx, M->ss_res);
    if (!R || !R->ring_ready) return 1;
    RET(ssub_taps(ctx, M, R));
    const BgGeom g = vp_geom(R);
    const BgGeom gM = vp_geom(M);
    const int nblk = g.nbr * g.nbc, RL = g.p_radius, ws = 16 + 2 * RL;
    if (ws > VP_WS) return 1;
    const int d1s = M->ss_d1s, d2s = M->ss_d2s;
    // the full-resolution rows / columns a ring-grown low-resolution block window reaches through the transposed taps, per block row / column
    int fmax = 1;
    std::vector<int> span_r((size_t)2 * g.nbr, 0), span_c((size_t)2 * g.nbc, 0);
    auto span = [&](const std::vector<int> &ptr, const std::vector<int> &idx, int n_low, int nb, std::vector<int> &out) {
        for (int bi = 0; bi < nb; ++bi) {
            int a0 = INT_MAX, a1 = -1;
            for (int j = std::max(0, bi * 16 - RL); j < bi * 16 - RL + ws && j < n_low; ++j) for (int t = ptr[j]; t < ptr[j + 1]; ++t) { a0 = std::min(a0, idx[t]); a1 = std::max(a1, idx[t]); }
            if (a1 < a0) { a0 = 0; a1 = 0; }
            out[(size_t)2 * bi] = a0; out[(size_t)2 * bi + 1] = a1;
            fmax = std::max(fmax, a1 - a0 + 1);
        }
    };
    span(M->ss_trp_h, M->ss_tri_h, d1s, g.nbr, span_r); span(M->ss_tcp_h, M->ss_tci_h, d2s, g.nbc, span_c);
    const size_t shm_b = ((size_t)ws * ws + (size_t)ws * fmax) * sizeof(double) + (size_t)fmax * fmax * sizeof(float);
    if (shm_b > 150 * 1024) return 1;
    // the low-resolution blocks B_L(:, k) meets: the box of up' A's support grown by the ring
    std::vector<int> need_ptr((size_t)K + 1, 0), need;
    need.reserve((size_t)K * 8);
    for (int32_t k = 0; k < K; ++k) {
        int r0, r1, c0, c1;
        low_box(M, A_rowidx, A_colptr[k], A_colptr[k + 1], r0, r1, c0, c1);
        if (r1 >= r0) {
            const int bi0 = std::max(0, r0 - RL) >> 4, bi1 = std::min(d1s - 1, r1 + RL) >> 4, bj0 = std::max(0, c0 - RL) >> 4, bj1 = std::min(d2s - 1, c1 + RL) >> 4;
            for (int bj = bj0; bj <= bj1; ++bj) for (int bi = bi0; bi <= bi1; ++bi) need.push_back(bj * g.nbr + bi);
        }
        need_ptr[k + 1] = (int)need.size();
    }
    VpLists L;
    if (vp_make_lists(need_ptr, need, nblk, K, true, L)) return 1;
    const int64_t nent = L.nent, ldp = ldu;
    if ((nent + K) * ldp * 8 > (int64_t(24) << 30)) return 1;
    ht.mark("lists");
    DevBuf *V = ctx->vp;
    DevBuf &dEb = V[1], &dEk = V[2], &dEs = V[3], &dG16 = V[4], &dBt = V[6], &dCst = V[7], &dPart = V[8], &dNptr = V[9], &dNent = V[10], &dErq = V[21], &dSr = V[22], &dSc = V[23];
    RET(vp_upload_lists(ctx, L));
    RET(to_dev(ctx, dSr, span_r.data(), span_r.size())); RET(to_dev(ctx, dSc, span_c.data(), span_c.size()));
    const size_t bt_bytes = (size_t)std::max(1, L.g16[nblk]) * BLKPX * 16 * sizeof(double);
    RET(dBt.ensure_hw(bt_bytes, ctx->hw_vp[6]));
    RET(dCst.ensure_hw((size_t)K * sizeof(double), ctx->hw_vp[7]));
    RET(dPart.ensure_hw((size_t)(nent + K) * ldp * sizeof(double), ctx->hw_vp[8]));
    CK(hipMemsetAsync(dBt.p, 0, bt_bytes, ctx->st()));
    const int64_t nnz = A_colptr[K];
    const int *erq = dErow;
    if (M->nr != M->nr_b || M->roff || M->coff) {
        RET(dErq.ensure_hw((size_t)std::max<int64_t>(1, nnz) * sizeof(int), ctx->hw_vp[21]));
        if (nnz) LAUNCH(ctx, "ssub_rows_block", k_vp_rows_block, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, dErow, nnz, M->nr, M->nr_b, M->roff, M->coff, dErq.as<int>());
        erq = dErq.as<int>();
    }
    if (nent > 0) {
        if (shm_b > 48 * 1024) CK(hipFuncSetAttribute((const void *)k_vp_build_b_ssub, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_b));
        LAUNCH(ctx, "temporal_build_B", k_vp_build_b_ssub, dim3((unsigned)nent), dim3(256), shm_b, dEb.as<int>(), dEk.as<int>(), dEs.as<int>(), dG16.as<int>(), g, RL, fmax, dSr.as<int>(), dSc.as<int>(),
               dColptr, erq, dAval, M->nr_b, M->ss_trp.as<int>(), M->ss_tri.as<int>(), M->ss_trw.as<float>(), M->ss_tcp.as<int>(), M->ss_tci.as<int>(), M->ss_tcw.as<float>(),
               R->ring_dr.as<int>(), R->ring_dc.as<int>(), R->W.as<float>(), dBt.as<double>());
    }
    LAUNCH(ctx, "temporal_const", k_vp_const, dim3((unsigned)K), dim3(256), 0, dColptr, dErow, dAval, gM, M->ymean_d.as<double>(), M->b0.as<double>(), dCst.as<double>());
    {   // A' Yc of the footprint rows: partial rows nent .. nent + K - 1 (every frame group of a row is written: empty footprints write zeros through the loop bounds)
        const int64_t Tc = M->Tc;
        const int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(64, Tc / 32));
        const int64_t cchunk = (Tc + nchunk - 1) / nchunk;
        CK(hipMemsetAsync((char *)dPart.p + (size_t)nent * ldp * sizeof(double), 0, (size_t)K * ldp * sizeof(double), ctx->st()));
        LAUNCH(ctx, "temporal_proj_rows", k_vp_rows_temporal, dim3((unsigned)K, (unsigned)nchunk), dim3(256), 0, M->Yc4.as<float4>(), M->d_b, M->T, dColptr, erq, dAval, cchunk,
               dPart.as<double>() + nent * ldp, ldp);
    }
    RET(vp_launch_proj(ctx, R, g, L, false, ldp));
    LAUNCH(ctx, "temporal_reduce_B", k_vp_reduce, dim3((unsigned)((ldu + 255) / 256), (unsigned)K), dim3(256), 0, dPart.as<double>(), ldp, dNptr.as<int>(), dNent.as<int>(),
           dCst.as<double>(), M->T, dU, ldu);
    ht.mark("launches");
    return 0;
}

// loads this translation unit's code object now (HIP loads it at the first launch of one of its kernels -- milliseconds each that would otherwise fall into the first iteration): cnmfe_create
int tu_warm_vproj() { hipFuncAttributes at; return hipFuncGetAttributes(&at, (const void *)k_vp_segsum) == hipSuccess ? 0 : -1; }

}  // namespace cnmfe
