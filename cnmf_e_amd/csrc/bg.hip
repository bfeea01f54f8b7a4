// B1/B2: ring-model background fit  ==  endoscope/fit_ring_model.m:1-127  on the resident block.
//
// The reference regresses, per patch pixel m, the centred background residual Bf(m,:) on the
// <=p ring neighbours Bf(ring(m),:) plus a constant: a (p+1)x(p+1) Gram + solve per pixel, with the
// operands GATHERED rows of one shared matrix (naive: 2*d*(p+1)^2*T flop and d*p*T gathered floats).
// MI355X formulation ("local covariance"):
//   B1  Bf = (Y - Ymean) - A*(C - Cmean)  written once, fp32, tiled as [16x16-pixel block][frame][256]
//   B2a Cov(a,b) = sum_t Bf(a,t)*Bf(b,t) for every pair of pixels whose 16x16 blocks are within
//       +-2 blocks: a block-sparse SYRK, 256x256xT' GEMMs on the fp64 MFMA pipe
//       (v_mfma_f64_16x16x4_f64; fp32 inputs are exact in fp64, accumulation is fp64 like MATLAB's)
//   B2b per pixel: gather the (p+1)^2 Gram and the RHS from the covariance table, add the ridge
//       1e-5*trace (fit_ring_model.m:106), Cholesky-solve in fp64 in LDS, write the p weights.
// Every ring pair of every patch pixel lies within +-2 blocks, so B2a computes each needed
// covariance exactly once (symmetric pairs once) instead of once per centre pixel.
#include "common.hpp"
#include "ring_solve_core.hpp"
#include "win_proj.hpp"
#include <type_traits>

namespace cnmfe {
// option win_i8_planes: 0 (default) = three digit planes of the video in the window projection when the sums run over at least 2048 frames (the rounding of a sample to
// 2^-23 of its pixel's largest value averages out with the number of frames: W 5e-7 .. 9e-7 of the oracle's at T = 3000 .. 20000), four for shorter recordings (T = 96:
// A moved by 2.2e-6, above the tests' 2e-6 -- and a short recording's projection costs microseconds either way); 3 / 4 force the choice
static inline bool win_planes3(cnmfe_ctx *ctx, int64_t T16) {
    const int64_t o = ctx->opt("win_i8_planes", 0);
    return o == 3 || (o != 4 && T16 * 16 >= 2048);
}


typedef float float4_t __attribute__((ext_vector_type(4)));

// ---- B1 ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_bf(const float4 *__restrict__ Y4, int64_t Tc, BgGeom g,
                                                  const int *__restrict__ arow, const int *__restrict__ acol, const float *__restrict__ aval,
                                                  const float *__restrict__ Cc, int64_t ldc, float *__restrict__ bf, int tchunk, double *__restrict__ rs) {
    const int blk = blockIdx.x;                    // 16x16 block id, column-major over (nbr, nbc)
    const int bi = blk % g.nbr, bj = blk / g.nbr;
    const int lp = threadIdx.x;                    // local pixel in 4x4-patch order (lp_of)
    const int lr = ((lp >> 4) & 3) * 4 + (lp & 3), lc = (lp >> 6) * 4 + ((lp >> 2) & 3);
    const int rb = bi * BLK + lr, cb = bj * BLK + lc;
    const bool in = rb < g.nr_b && cb < g.nc_b;
    const int64_t q = in ? (int64_t)cb * g.nr_b + rb : 0;
    int e0 = 0, e1 = 0;
    if (in && arow) { e0 = arow[q]; e1 = arow[q + 1]; }
    const int64_t tp0 = (int64_t)blockIdx.y * tchunk;          // tchunk is a multiple of 4
    const int64_t tp1 = tp0 + tchunk < g.Tpad ? tp0 + tchunk : g.Tpad;
    float *out = bf + ((int64_t)blk * g.Tpad) * BLKPX + lp;
    if (g.kstride == 1) {                          // the video is resident centred: Bf = Yc - A*(C - Cmean), 4 frames per load
        for (int64_t tp = tp0; tp < tp1; tp += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t c = tp >> 2;
            if (in && c < Tc) {
                v = Y4[c * g.d_b + q];
                for (int e = e0; e < e1; ++e) {
                    const float av = aval[e];
                    const float4 c4 = *reinterpret_cast<const float4 *>(Cc + (int64_t)acol[e] * ldc + tp);
                    v.x -= av * c4.x; v.y -= av * c4.y; v.z -= av * c4.z; v.w -= av * c4.w;
                }
            }
            if (g.bf4) reinterpret_cast<float4 *>(bf)[((int64_t)blk * (g.Tpad >> 2) + c) * BLKPX + lp] = v;
            else { out[tp * BLKPX] = v.x; out[(tp + 1) * BLKPX] = v.y; out[(tp + 2) * BLKPX] = v.z; out[(tp + 3) * BLKPX] = v.w; }
        }
    } else {                                       // frame subsampling Bf(:, 1:k:end)  (fit_ring_model.m:87)
        const float *Ys = reinterpret_cast<const float *>(Y4);
        for (int64_t tp = tp0; tp < tp1; ++tp) {
            float v = 0.f;
            if (in && tp < g.Tp) {
                const int64_t t = tp * g.kstride;
                v = Ys[((t >> 2) * g.d_b + q) * 4 + (t & 3)];
                for (int e = e0; e < e1; ++e) v -= aval[e] * Cc[(int64_t)acol[e] * ldc + t];
            }
            if (g.bf4) bf[(((int64_t)blk * (g.Tpad >> 2) + (tp >> 2)) * BLKPX + lp) * 4 + (tp & 3)] = v;
            else out[tp * BLKPX] = v;
        }
    }
}

// row sums of Bf over the used frames (the "ones" row of X, fit_ring_model.m:101)
__global__ void __launch_bounds__(256) k_rowsum(const float *__restrict__ bf, int64_t Tpad, double *__restrict__ rs, int bf4) {
    const int64_t blk = blockIdx.x;
    if (bf4) {
        const float4 *src4 = reinterpret_cast<const float4 *>(bf) + blk * (Tpad >> 2) * BLKPX + threadIdx.x;
        double s0 = 0, s1 = 0;
        for (int64_t c = 0; c < (Tpad >> 2); ++c) { const float4 v = src4[c * BLKPX]; s0 += (double)v.x + (double)v.z; s1 += (double)v.y + (double)v.w; }
        rs[blk * BLKPX + threadIdx.x] = s0 + s1;
        return;
    }
    const float *src = bf + blk * Tpad * BLKPX + threadIdx.x;
    double s0 = 0, s1 = 0;
    int64_t t = 0;
    for (; t + 1 < Tpad; t += 2) { s0 += src[t * BLKPX]; s1 += src[(t + 1) * BLKPX]; }
    if (t < Tpad) s0 += src[t * BLKPX];
    rs[blk * BLKPX + threadIdx.x] = s0 + s1;
}

// ---- B2a: block-sparse SYRK on the matrix pipe ------------------------------------------------------
// One workgroup = one 128x128 quadrant of one 256x256 block-pair covariance (16x16x4 MFMA: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]);
// K (= frames) advances GK = 16 per LDS stage.
constexpr int GK = 16;

// ---- dense tile slots ----------------------------------------------------------------------------------
// A first version gave every wave a FIXED 16-slot pattern of the quadrant with a `need` test per slot: hipcc turned that into two
// scalar branches per MFMA -- the issue stream, not the matrix pipe, was saturated (77 TF/s), and the four waves of a workgroup were
// unevenly loaded (84 %).  Here the host lists the needed 16x16 sub-tiles of
// every (displacement class, quadrant); tile t of an item goes to wave t%4, slot t/4, so slots are dense, the
// waves are balanced to within one tile, and the stage body is straight-line code instantiated per slot count.
// Bf is stored [block][frame/4][256][4] (like the resident video): a lane's A (or B) fragment for ALL FOUR
// k-steps of a 16-frame stage is ONE conflict-free ds_read_b128 (lane (l&15, l>>4) reads quad-row l>>4: MFMA m
// contracts frames {4*(l>>4) + m}), at lane base + a per-slot scalar tile offset.
constexpr int G4_NBUF = 4, G4_STAGE_F = 2 * GK * 128;

struct G4Wave {                       // per-wave constants of a work item (all wave-uniform except lbase, vo0, vo1)
    const float *gA, *gB;
    unsigned vo0, vo1, dA0;
    int lbase, nst, probe, ns, lane;
    double *out;                      // cov + pair*256*256 + (ih*128)*256 + jh*128
};

// the whole stage loop for a wave that owns NS slots (the last one only if `ns == NS`): instantiated per NS and
// selected ONCE per workgroup, so the loop body is branch-free straight-line code with its own register allocation
// (a switch inside the loop made hipcc keep the accumulators in scratch: 456 spilled VGPRs).
// fp64 MFMAs: exact products of the fp32 operands, fp64 sums.  Rounds 1-4 also carried an fp32-MFMA mode with fp64 shadow accumulation (89 ms at H, W error 2e-4)
// and a split-bf16 mode (4 bf16 products per fp32 product, 50 ms, 9e-5) for the direct Gram: both retired in round 5, when the int8 digit Gram (gram_i8.hpp: 64 ms,
// EXACT up to a 32-bit quantisation, 1e-8 of W) took over every use that fits its int32 range -- this kernel remains for the outlier branch (a clipped, frame-selected
// Bf) and for recordings beyond 24576 used frames.
template <int NS>
__device__ __forceinline__ void gram4_run(const G4Wave &w, const float *smem, const int *__restrict__ tlw) {
    int ao[NS], bo[NS], ti[NS], tj[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int code = __builtin_amdgcn_readfirstlane(sl < w.ns ? tlw[sl * 4] : 0);
        ti[sl] = code & 7; tj[sl] = code >> 4;
        ao[sl] = ti[sl] * 64; bo[sl] = GK * 128 + tj[sl] * 64;
    }
    const bool last = w.ns == NS;
    auto issue = [&](int st) {
        int sc = st < w.nst ? st : w.nst - 1;                            // clamp: keeps the vmcnt arithmetic uniform
        if (w.probe & 1) sc = 0;                                         // A/B probe (gram_probe bit 0): every stage re-reads stage 0 -> no fabric traffic
        const float *sa = w.gA + (int64_t)sc * GK * BLKPX, *sb = w.gB + (int64_t)sc * GK * BLKPX;
        const unsigned d = w.dA0 + (unsigned)(st & (G4_NBUF - 1)) * (G4_STAGE_F * 4u);
        glds16(sa, w.vo0, d); glds16(sa, w.vo1, d + 4096u);
        glds16(sb, w.vo0, d + 8192u); glds16(sb, w.vo1, d + 12288u);
    };
    double4_t acc[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) acc[sl] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s0 = 0; s0 < G4_NBUF - 1; ++s0) issue(s0);
    for (int st = 0; st < w.nst; ++st) {
        if (!(w.probe & 2)) {                                                         // (gram_probe bit 1: timing experiment without the stage sync)
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(4 * (G4_NBUF - 2)) : "memory");    // this wave's part of stage st has landed
        __builtin_amdgcn_s_barrier();                                                // ... and everybody else's; buffer st-1 is free
        }
        asm volatile("" ::: "memory");
        issue(st + G4_NBUF - 1);
        int lb = w.lbase;
        asm volatile("" : "+v"(lb));            // opaque per stage: no hoisting of 2*NS per-slot address VGPRs out of the loop
        const float *lp = smem + (st & (G4_NBUF - 1)) * G4_STAGE_F + lb;
        // slots go in batches of two: the fragments of the next batch (4 ds_read_b128) are issued before the current
        // batch's MFMAs, and the two slots' MFMA chains are interleaved so that no MFMA waits on its predecessor's
        // result (the one-slot version stalled an LDS latency per slot: hipcc waited lgkmcnt(0) before every chain)
        float4 fa[2][2], fb[2][2];
        auto ldb = [&](int buf, int base) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (base + u < NS) {
                    fa[buf][u] = *reinterpret_cast<const float4 *>(lp + ao[base + u]);
                    fb[buf][u] = *reinterpret_cast<const float4 *>(lp + bo[base + u]);
                }
        };
        ldb(0, 0);
#pragma unroll
        for (int base = 0; base < NS; base += 2) {
            const int cur = (base >> 1) & 1;
            if (base + 2 < NS) ldb(cur ^ 1, base + 2);
            asm volatile("" ::: "memory");
            const bool on0 = true, on1 = (base + 1 < NS - 1) || (base + 1 == NS - 1 && last);
            const bool only0_last = (base == NS - 1);                   // odd NS: the last slot stands alone
            const bool do0 = only0_last ? last : on0;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (base + u >= NS) continue;
                    if (u == 0 ? !do0 : !on1) continue;
                    const float av = kq == 0 ? fa[cur][u].x : kq == 1 ? fa[cur][u].y : kq == 2 ? fa[cur][u].z : fa[cur][u].w;
                    const float bv = kq == 0 ? fb[cur][u].x : kq == 1 ? fb[cur][u].y : kq == 2 ? fb[cur][u].z : fb[cur][u].w;
                    acc[base + u] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av, (double)bv, acc[base + u], 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // drain the clamped tail loads before the LDS is released
    const int fl = w.lane & 15;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
        if (sl < NS - 1 || last) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = (w.lane >> 4) + 4 * r;                    // D layout (fp64 16x16): row = (lane >> 4) + 4 r
                w.out[(int64_t)(ti[sl] * 16 + rr) * BLKPX + tj[sl] * 16 + fl] = acc[sl][r];
            }
        }
}

__global__ void __launch_bounds__(256, 2) k_gram4(const float *__restrict__ bf, int64_t Tpad, const int4 *__restrict__ pairs,
                                                  const int *__restrict__ work, int nwork, const int *__restrict__ tl_cnt,
                                                  const int *__restrict__ tl, int probe, double *__restrict__ cov) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // the ONLY LDS object (a second one de-pipelines the DMA)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if (nwg % 8 == 0) bid = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    if (bid >= nwork) return;
    const int wk = work[bid];
    const int pair = wk >> 2, quad = wk & 3;
    const int ih = quad & 1, jh = quad >> 1;
    const int4 pr = pairs[pair];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lidx = pr.z * 4 + quad;
    const int cnt = __builtin_amdgcn_readfirstlane(tl_cnt[lidx]);
    G4Wave w;
    w.gA = bf + ((int64_t)pr.x * Tpad) * BLKPX + ih * 512;             // a quad-row is 256 px x 4 frames; half ih starts at px 128
    w.gB = bf + ((int64_t)pr.y * Tpad) * BLKPX + jh * 512;
    // DMA: one wave-instruction = 64 px x 4 frames (1 KB) of one quad-row.  Per half and stage: 4 quad-rows x 2 = 8
    // instructions; wave w moves instruction w (quad-row w>>1, pixels (w&1)*64..) and w+4 (quad-row 2 + (w>>1)).
    w.vo0 = (unsigned)((((wave >> 1) * BLKPX + (wave & 1) * 64 + lane) * 4) * 4);
    w.vo1 = w.vo0 + 2u * BLKPX * 4u * 4u;
    w.dA0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem + (unsigned)wave * 1024u;
    w.lbase = (lane >> 4) * 512 + (lane & 15) * 4;                     // quad-row l>>4, pixel l&15 of the tile, 4 frames
    w.nst = (int)(Tpad / GK); w.probe = probe; w.lane = lane;
    w.ns = cnt > wave ? (cnt - wave + 3) >> 2 : 0;                     // tiles wave, wave+4, ...
    w.out = cov + (int64_t)pair * BLKPX * BLKPX + (int64_t)(ih * 128) * BLKPX + jh * 128;
    const int *tlw = tl + lidx * 64 + wave;
    switch ((cnt + 3) >> 2) {                                          // slots of the busiest wave; the others skip the last one
#define G4_CASE(N) case N: gram4_run<N>(w, smem, tlw); break;
        G4_CASE(1) G4_CASE(2) G4_CASE(3) G4_CASE(4) G4_CASE(5) G4_CASE(6) G4_CASE(7) G4_CASE(8)
        G4_CASE(9) G4_CASE(10) G4_CASE(11) G4_CASE(12) G4_CASE(13) G4_CASE(14) G4_CASE(15) G4_CASE(16)
#undef G4_CASE
        default: break;
    }
}

// ---- B2a', incremental: Cov(Bf) from the covariance of the VIDEO ------------------------------------------------
// Bf = Yc - A Cc (fit_ring_model.m:45-47), so  sum_t Bf_i Bf_j = sum_t Yc_i Yc_j - sum_k A_jk U~_ik - sum_k A_ik U~_jk  with
// U~_ik = sum_t (Yc_i - 1/2 sum_l A_il Cc_l)(t) Cc_k(t)   (the 1/2 shares the A G A' term between the two sums; G = Cc Cc').
// The first term does not depend on A, C: it is computed ONCE per patch and frame stride on the fp64 matrix pipe (the block-sparse
// SYRK above on Yc alone) and kept; every later fit only needs U~ for pixels within two 16x16 blocks of a footprint -- per block a
// (256 px) x (footprints near it, <= 64) x T' GEMM on the fp64 pipe, 0.1-0.2 TFLOP instead of 9.6 -- and one sweep over the table.
// Everything is fp64 (exact fp32 products, fp64 sums): the difference of the two large terms keeps ~1e-13 relative accuracy.
// csum[k] = sum over the used frames of Cc_k  (the ones-row of X: rowsum(Bf) = rowsum(Yc) - A csum)
__global__ void __launch_bounds__(256) k_trace_subsum(const float *__restrict__ Cc, int64_t ldc, int64_t Tp, int kstride, double *__restrict__ csum) {
    const float *row = Cc + (int64_t)blockIdx.x * ldc;
    __shared__ double red[256];
    double s = 0;
    for (int64_t tp = threadIdx.x; tp < Tp; tp += 256) s += (double)row[tp * kstride];
    red[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) csum[blockIdx.x] = red[0];
}

// U(i,k) = sum_t Yc_i(t) Cc_k(t) for the 256 pixels of a block and the (<= 16 NT) traces on its list, plus G(k,l) = sum_t Cc_k Cc_l on the same
// list: plain GEMMs on the fp64 matrix pipe with both operands read straight from global memory -- lane (fi, kq) of a 16x16x4 fragment wants
// ONE float (pixel fi of a 4x4 patch / trace fi of a 16-trace group, frame kq of the chunk), and the 64 lanes of a pixel fragment cover four
// full 64-byte segments of the 4-frame-interleaved video.  No LDS, no barriers; loads run two chunks ahead of the MFMAs.  One launch covers
// all blocks (longest lists first) x frame segments; partial sums go to per-segment buffers that k_win_fix adds in a fixed order.
// (A first version staged Z = Yc - 1/2 A Cc through LDS with two barriers per chunk and one launch per list length: 7.5 ms.)
template <int NT>
__device__ __forceinline__ void win_body(const float4 *__restrict__ Y4, const BgGeom &g, const float *__restrict__ Cc, int64_t ldc, int blk, int l0, int nl,
                                         const int *__restrict__ lst_k, int64_t c0, int64_t c1, double *__restrict__ Ut, double *__restrict__ Gb) {
    const int bi = blk % g.nbr, bj = blk / g.nbr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, kq = lane >> 4;
    const float *Ys = reinterpret_cast<const float *>(Y4);
    const float *ya[4]; const float *tb[NT];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int lp = (wave * 4 + a) * 16 + fi;
        const int lr = ((lp >> 4) & 3) * 4 + (lp & 3), lc = (lp >> 6) * 4 + ((lp >> 2) & 3);
        const int rb = bi * BLK + lr, cb = bj * BLK + lc;
        ya[a] = (rb < g.nr_b && cb < g.nc_b) ? Ys + ((int64_t)cb * g.nr_b + rb) * 4 : nullptr;
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) { const int sl = b * 16 + fi; tb[b] = sl < nl ? Cc + (int64_t)lst_k[l0 + sl] * ldc : nullptr; }
    double4_t acc[4][NT], accg[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        accg[b] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    struct Frag { float y[4]; float t[NT]; };
    auto load = [&](int64_t c) {
        Frag f;
        const int64_t tp = 4 * c + kq, t = tp * g.kstride;
        const bool on = c < c1 && tp < g.Tp;
        const int64_t yo = ((t >> 2) * g.d_b) * 4 + (t & 3);
#pragma unroll
        for (int a = 0; a < 4; ++a) f.y[a] = (on && ya[a]) ? ya[a][yo] : 0.f;
#pragma unroll
        for (int b = 0; b < NT; ++b) f.t[b] = (on && tb[b]) ? tb[b][t] : 0.f;
        return f;
    };
    const bool gw = wave < NT;                              // wave w also owns row-group w of G
    auto mm = [&](const Frag &f) {
        double bv[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[b] = (double)f.t[b];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double av = (double)f.y[a];
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[b], acc[a][b], 0, 0, 0);
        }
        if (gw) {
            double gv = bv[0];
#pragma unroll
            for (int b = 1; b < NT; ++b) gv = wave == b ? bv[b] : gv;
#pragma unroll
            for (int b = 0; b < NT; ++b) accg[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(gv, bv[b], accg[b], 0, 0, 0);
        }
    };
    // loads run WIN_AHEAD chunks in front of the MFMAs; loads past the segment return zeros.  (Measured at H: 2 ahead 5.5 ms, 3 ahead 5.4 ms,
    // 6 ahead 6.0 ms -- each extra chunk costs 18 VGPRs and beyond 256 the kernel drops to one wave per SIMD; with lists of ~48 traces the
    // 13 fp64 MFMAs per chunk and wave are ~4 ms of matrix-pipe time at the clock this kernel runs at, so latency is not what is left.)
    Frag f[WIN_AHEAD];
#pragma unroll
    for (int d = 0; d < WIN_AHEAD; ++d) f[d] = load(c0 + d);
    for (int64_t c = c0; c < c1; c += WIN_AHEAD) {
#pragma unroll
        for (int d = 0; d < WIN_AHEAD; ++d) {
            const Frag nx = load(c + WIN_AHEAD + d);
            mm(f[d]);
            f[d] = nx;
        }
    }
    // D layout (fp64 16x16): row = (lane>>4) + 4r, col = lane&15
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int slot = b * 16 + fi;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (slot < nl)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ut[(int64_t)(l0 + slot) * BLKPX + (wave * 4 + a) * 16 + kq + 4 * r] = acc[a][b][r];
        if (gw)
#pragma unroll
            for (int r = 0; r < 4; ++r) Gb[(int64_t)blk * WIN_NLB * WIN_NLB + (wave * 16 + kq + 4 * r) * WIN_NLB + slot] = accg[b][r];
    }
}

__global__ void __launch_bounds__(256) k_win_proj(const float4 *__restrict__ Y4, BgGeom g, const float *__restrict__ Cc, int64_t ldc, const int *__restrict__ lst_ptr,
                                                  const int *__restrict__ lst_k, const int *__restrict__ blk_list, int nseg, double *__restrict__ Ut, int64_t ut_stride,
                                                  double *__restrict__ Gb, int64_t gb_stride) {
    const int blk = blk_list[blockIdx.x / nseg], seg = blockIdx.x % nseg;
    const int l0 = lst_ptr[blk], nl = lst_ptr[blk + 1] - l0;
    double *ut = Ut + seg * ut_stride, *gb = Gb + seg * gb_stride;
    const int64_t nchunk = (g.Tp + 3) >> 2;
    int64_t cseg = ((nchunk + nseg - 1) / nseg + 1) & ~int64_t(1);
    const int64_t c0 = seg * cseg, c1 = c0 + cseg < nchunk ? c0 + cseg : nchunk;
    switch ((nl + 15) >> 4) {
        case 1: win_body<1>(Y4, g, Cc, ldc, blk, l0, nl, lst_k, c0, c1, ut, gb); break;
        case 2: win_body<2>(Y4, g, Cc, ldc, blk, l0, nl, lst_k, c0, c1, ut, gb); break;
        case 3: win_body<3>(Y4, g, Cc, ldc, blk, l0, nl, lst_k, c0, c1, ut, gb); break;
        case 4: win_body<4>(Y4, g, Cc, ldc, blk, l0, nl, lst_k, c0, c1, ut, gb); break;
        default: break;
    }
}

// U~(i,k) = sum_seg U_seg(i,k) - 1/2 sum_{l at i} A_il sum_seg G_seg(l,k), written to segment 0.
// Workgroup = (block, WF_S traces of its list): a small patch has ~100 blocks but 16 frame segments, and one workgroup per block summed its
// 16 x (list length + 16) strided partials in one serial loop per thread -- 0.21 ms per launch against 0.06 ms for the whole 512 x 512 frame.
constexpr int WF_S = 4;
__global__ void __launch_bounds__(256) k_win_fix(BgGeom g, int K, const int *__restrict__ arow, const int *__restrict__ acol, const float *__restrict__ aval,
                                                 const int *__restrict__ lst_ptr, const short *__restrict__ slot_of, const int *__restrict__ blk_list, int nseg,
                                                 double *__restrict__ Ut, int64_t ut_stride, const double *__restrict__ Gb, int64_t gb_stride, double *__restrict__ Praw,
                                                 const double *__restrict__ GK, const int *__restrict__ lst_k) {
    __shared__ double G[WIN_NLB * WF_S];                     // G[r][j]: column s0 + j of the block's list Gram matrix
    const int blk = blk_list[blockIdx.x];
    const int l0 = lst_ptr[blk], nl = lst_ptr[blk + 1] - l0;
    const int s0 = (int)blockIdx.y * WF_S;
    if (s0 >= nl) return;
    const int lp = threadIdx.x;
    const int nlp = ((nl + 15) >> 4) << 4;
    {
        const int r = lp / WF_S, c = s0 + lp % WF_S;         // WIN_NLB * WF_S == 256: one entry per thread
        double v = 0.0;
        if (GK) {                                            // (win_proj_i8.hpp: one K x K Gram matrix of the centred traces per fit instead of per-block, per-segment copies)
            if (r < nl && c < nl) v = GK[(int64_t)lst_k[l0 + r] * K + lst_k[l0 + c]];
        } else if (r < nlp && c < nlp) {
            const double *gp = Gb + (int64_t)blk * WIN_NLB * WIN_NLB + r * WIN_NLB + c;
#pragma unroll 4
            for (int sg = 0; sg < nseg; ++sg) v += gp[sg * gb_stride];
        }
        G[lp] = v;
    }
    __syncthreads();
    const int lr = ((lp >> 4) & 3) * 4 + (lp & 3), lc = (lp >> 6) * 4 + ((lp >> 2) & 3);
    const int rb = (blk % g.nbr) * BLK + lr, cb = (blk / g.nbr) * BLK + lc;
    int e0 = 0, e1 = 0;
    if (rb < g.nr_b && cb < g.nc_b) { const int64_t q = (int64_t)cb * g.nr_b + rb; e0 = arow[q]; e1 = arow[q + 1]; }
    const int s1 = s0 + WF_S < nl ? s0 + WF_S : nl;
    for (int s = s0; s < s1; ++s) {
        double *u = Ut + (int64_t)(l0 + s) * BLKPX + lp;
        double v = u[0];
#pragma unroll 4
        for (int sg = 1; sg < nseg; ++sg) v += u[sg * ut_stride];
        double w = 0.0;
        for (int e = e0; e < e1; ++e) w += (double)aval[e] * G[(int)slot_of[(int64_t)blk * K + acol[e]] * WF_S + (s - s0)];
        u[0] = v - 0.5 * w;
        if (Praw) Praw[(int64_t)(l0 + s) * BLKPX + lp] = v;      // U itself = Yc Cc' on the block: the P table of the sweep-free spatial update (vproj.hip)
    }
}
static_assert(WIN_NLB * WF_S == 256, "k_win_fix: one Gram entry per thread");

// cov(pair)(i,j) = base(pair)(i,j) - sum_{k at j} A_jk U~_a(k, i) - sum_{k at i} A_ik U~_b(k, j)  over the needed 16x16 sub-tiles
// Thread = (half h, row ty of a sub-tile, column PAIR tx2): 16-byte loads and stores (one wave instruction moves 1 KB of the 7.7 GB sweep); the two
// 128-thread halves of the workgroup take the row patches pi, pi + 1 side by side (a half is two whole waves: their need masks may differ).
__global__ void __launch_bounds__(256) k_cov_correct(const double *__restrict__ base, double *__restrict__ cov, const int4 *__restrict__ pairs,
                                                     const unsigned short *__restrict__ needmask, BgGeom g, int K, const int *__restrict__ arow,
                                                     const int *__restrict__ acol, const float *__restrict__ aval, const int *__restrict__ lst_ptr,
                                                     const short *__restrict__ slot_of, const double *__restrict__ Ut) {
    const int pair = blockIdx.x;
    const int4 pr = pairs[pair];
    const int ba = pr.x, bb = pr.y, rel = pr.z;
    const int half = threadIdx.x >> 7, ty = (threadIdx.x >> 3) & 15, tx = (threadIdx.x & 7) * 2;
    const int la = lst_ptr[ba], lb = lst_ptr[bb];
    auto pix = [&](int blk, int lp, int &e0, int &e1) {
        const int lr = ((lp >> 4) & 3) * 4 + (lp & 3), lc = (lp >> 6) * 4 + ((lp >> 2) & 3);
        const int rb = (blk % g.nbr) * BLK + lr, cb = (blk / g.nbr) * BLK + lc;
        e0 = e1 = 0;
        if (rb < g.nr_b && cb < g.nc_b) { const int64_t q = (int64_t)cb * g.nr_b + rb; e0 = arow[q]; e1 = arow[q + 1]; }
    };
    const double *bp = base + (int64_t)pair * BLKPX * BLKPX;
    double *cp = cov + (int64_t)pair * BLKPX * BLKPX;
    // gridDim.y workgroups share a pair (its 16 row patches split evenly): a small patch has too few pairs to fill the chip with one each
    const int pi0 = (int)blockIdx.y * (16 / (int)gridDim.y), pi1 = pi0 + 16 / (int)gridDim.y;
    for (int pi = pi0 + half; pi < pi1; pi += 2) {
        const unsigned mask = needmask[rel * 16 + pi];
        if (!mask) continue;
        const int ilp = pi * 16 + ty;
        int ei0, ei1; pix(ba, ilp, ei0, ei1);
        for (int pj = 0; pj < 16; ++pj) {
            if (!((mask >> pj) & 1u)) continue;
            const int jlp = pj * 16 + tx;
            int ej0, ej1, ek0, ek1; pix(bb, jlp, ej0, ej1); pix(bb, jlp + 1, ek0, ek1);
            const int idx = ilp * BLKPX + jlp;
            double2 v = *reinterpret_cast<const double2 *>(bp + idx);
            for (int e = ej0; e < ej1; ++e) v.x -= (double)aval[e] * Ut[(int64_t)(la + slot_of[(int64_t)ba * K + acol[e]]) * BLKPX + ilp];
            for (int e = ek0; e < ek1; ++e) v.y -= (double)aval[e] * Ut[(int64_t)(la + slot_of[(int64_t)ba * K + acol[e]]) * BLKPX + ilp];
            for (int e = ei0; e < ei1; ++e) {
                const double2 u = *reinterpret_cast<const double2 *>(Ut + (int64_t)(lb + slot_of[(int64_t)bb * K + acol[e]]) * BLKPX + jlp);
                v.x -= (double)aval[e] * u.x; v.y -= (double)aval[e] * u.y;
            }
            *reinterpret_cast<double2 *>(cp + idx) = v;
        }
    }
}

__global__ void __launch_bounds__(256) k_rowsum_correct(const double *__restrict__ base, double *__restrict__ rs, BgGeom g, const int *__restrict__ arow,
                                                        const int *__restrict__ acol, const float *__restrict__ aval, const double *__restrict__ csum) {
    const int blk = blockIdx.x, lp = threadIdx.x;
    const int lr = ((lp >> 4) & 3) * 4 + (lp & 3), lc = (lp >> 6) * 4 + ((lp >> 2) & 3);
    const int rb = (blk % g.nbr) * BLK + lr, cb = (blk / g.nbr) * BLK + lc;
    double v = base[(int64_t)blk * BLKPX + lp];
    if (rb < g.nr_b && cb < g.nc_b) {
        const int64_t q = (int64_t)cb * g.nr_b + rb;
        for (int e = arow[q]; e < arow[q + 1]; ++e) v -= (double)aval[e] * csum[acol[e]];
    }
    rs[(int64_t)blk * BLKPX + lp] = v;
}

// ---- helpers on the covariance table ----------------------------------------------------------------
struct CovTab {
    const double *cov; const int *pair_of;   // pair_of[blk*nrel + rel] or -1
    const int *wcodes; int woff;             // [(bc0 + woff) * (nbr + 2 woff) + (br0 + woff)][256]: the block-pair codes of the 4 x 4-block window with origin (br0, bc0) >= -woff, k_win_codes
    int nbr, nbc;
    int maxd, nrel;                          // largest block displacement between two ring pixels of one centre (2: radius <= 16, 3: <= 24); nrel_of(maxd)
};
// canonical displacement index: dC in 0..maxd; dC == 0 -> dR in 0..maxd (0..maxd); dC >= 1 -> dR in -maxd..maxd.  maxd = 2: 13 classes (0..2, 3..7, 8..12)
__host__ __device__ __forceinline__ int rel_index(int dR, int dC, int maxd) { return dC == 0 ? dR : (maxd + 1) + (dC - 1) * (2 * maxd + 1) + dR + maxd; }
__host__ __device__ __forceinline__ int nrel_of(int maxd) { return (maxd + 1) + maxd * (2 * maxd + 1); }

// The block-pair codes of a pixel's 4 x 4-block window depend on the window's origin only, so they are tabulated once per fit instead of being
// re-derived (25 integer operations and a dependent load per entry, 256 entries) by every pixel's wave.
__global__ void __launch_bounds__(256) k_win_codes(const int *__restrict__ pair_of, int nbr, int nbc, int woff, int maxd, int nrel, int *__restrict__ wcodes) {
    const int br0 = (int)(blockIdx.x % (nbr + 2 * woff)) - woff, bc0 = (int)(blockIdx.x / (nbr + 2 * woff)) - woff;
    const int q = threadIdx.x, a = q >> 4, b = q & 15;
    int ia = br0 + (a & 3), ja = bc0 + (a >> 2), ib = br0 + (b & 3), jb = bc0 + (b >> 2);
    int code = -1;
    if (ia >= 0 && ja >= 0 && ib >= 0 && jb >= 0 && ia < nbr && ib < nbr && ja < nbc && jb < nbc) {
        int dR = ib - ia, dC = jb - ja, sw = 0;
        if (dC < 0 || (dC == 0 && dR < 0)) { sw = 1; ia = ib; ja = jb; dR = -dR; dC = -dC; }
        if (dC <= maxd && dR <= maxd && dR >= -maxd) {
            const int pidx = pair_of[(ja * nbr + ia) * nrel + rel_index(dR, dC, maxd)];
            code = pidx < 0 ? -1 : ((pidx << 2) | (sw << 1) | ((dR == 0 && dC == 0) ? 1 : 0));
        }
    }
    wcodes[(int64_t)blockIdx.x * 256 + q] = code;
}
}  // namespace cnmfe
#include "ring_solve.hpp"
#include "ring_solve_packed.hpp"
#include "ring_solve_inv.hpp"
#include "ring_solve_staged.hpp"
#include "gram_i8.hpp"
#include "win_proj_i8.hpp"
namespace cnmfe {

// ind_active = abs(W_old)*sum(A,2) > 0  (fit_ring_model.m:28)
__global__ void k_active(const float *__restrict__ W, BgGeom g, const int *__restrict__ dr, const int *__restrict__ dc,
                         const float *__restrict__ asum, unsigned char *__restrict__ active, int *__restrict__ nactive) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int on = 0;
    if (m < g.d) {
        const int rbm = (int)(m % g.nr) + g.roff, cbm = (int)(m / g.nr) + g.coff;
        float s = 0.f;
        // eight offsets at a time, every load unconditional (neighbours outside the block read pixel 0 and are masked): the loads of a group are
        // independent, one load behind a branch per trip made the kernel 2 x 96 memory latencies long.  The sum keeps the ring order.
        for (int i0 = 0; i0 < g.p; i0 += 8) {
            float w8[8], a8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u < g.p ? i0 + u : g.p - 1;
                const int rb = rbm + dr[i], cb = cbm + dc[i];
                const bool in = i0 + u < g.p && rb >= 0 && rb < g.nr_b && cb >= 0 && cb < g.nc_b;
                w8[u] = W[(int64_t)i * g.d + m];
                a8[u] = asum[in ? (int64_t)cb * g.nr_b + rb : 0];
                if (!in) w8[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (w8[u] != 0.f) s += fabsf(w8[u]) * a8[u];
        }
        on = s > 0.f;
        active[m] = (unsigned char)on;
    }
    unsigned long long b = __ballot(on);
    if ((threadIdx.x & 63) == 0) atomicAdd(nactive, (int)__popcll(b));
}

// b0 = Ymean(ind_patch) - A(ind_patch,:)*Cmean   (fit_ring_model.m:44), double
__global__ void k_b0(const double *__restrict__ ymean, BgGeom g, const int *__restrict__ arow, const int *__restrict__ acol,
                     const float *__restrict__ aval, const double *__restrict__ Cmean, double *__restrict__ b0) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= g.d) return;
    const int64_t q = (int64_t)((int)(m / g.nr) + g.coff) * g.nr_b + (int)(m % g.nr) + g.roff;
    double v = ymean[q];
    if (arow) for (int e = arow[q]; e < arow[q + 1]; ++e) v -= (double)aval[e] * Cmean[acol[e]];
    b0[m] = v;
}

// ---- outlier branch (fit_ring_model.m:50-56): Bf_old = W_old*Bf; entries of the patch rows above Bf_old + thresh*sn take the value of
// Bf_old.  Bf_old is formed from the unmodified Bf (the reference computes it before touching tmp_Bf), so the clipped rows go to a
// second buffer.  One thread per (patch pixel, 4 frames); cnt[t] = sum(ind_outlier(:, t)) feeds the frame selection of :62-67.
__device__ __forceinline__ int64_t bf4_index(const BgGeom &g, int rb, int cb, int64_t c) {
    return ((int64_t)((cb >> 4) * g.nbr + (rb >> 4)) * (g.Tpad >> 2) + c) * BLKPX + lp_of(rb & 15, cb & 15);
}
__global__ void __launch_bounds__(256) k_outlier_clip(const float4 *__restrict__ bf, float4 *__restrict__ bf2, BgGeom g, const int *__restrict__ dr,
                                                      const int *__restrict__ dc, const float *__restrict__ W, const float *__restrict__ sn_b,
                                                      double thresh, int *__restrict__ cnt) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (m >= g.d) return;
    const int rbm = (int)(m % g.nr) + g.roff, cbm = (int)(m / g.nr) + g.coff;
    double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
    for (int i = 0; i < g.p; ++i) {
        const float w = W[(int64_t)i * g.d + m];
        if (w == 0.f) continue;                                   // outside the FOV (or an exact zero: no term either way)
        const float4 v = bf[bf4_index(g, rbm + dr[i], cbm + dc[i], c)];
        o0 += (double)w * v.x; o1 += (double)w * v.y; o2 += (double)w * v.z; o3 += (double)w * v.w;
    }
    const int64_t at = bf4_index(g, rbm, cbm, c);
    float4 y = bf[at];
    const double lim = thresh * (double)sn_b[(int64_t)cbm * g.nr_b + rbm];
    const int64_t t = c * 4;
    if (t < g.T && (double)y.x > o0 + lim) { y.x = (float)o0; atomicAdd(&cnt[t], 1); }
    if (t + 1 < g.T && (double)y.y > o1 + lim) { y.y = (float)o1; atomicAdd(&cnt[t + 1], 1); }
    if (t + 2 < g.T && (double)y.z > o2 + lim) { y.z = (float)o2; atomicAdd(&cnt[t + 2], 1); }
    if (t + 3 < g.T && (double)y.w > o3 + lim) { y.w = (float)o3; atomicAdd(&cnt[t + 3], 1); }
    bf2[at] = y;
}
// Bf = Bf(:, ind_frames) (:66): frame j of the destination is frame sel[j] of the source; frames past nsel are zero padding
__global__ void __launch_bounds__(256) k_select_frames(const float *__restrict__ src, int64_t Tpad_src, float *__restrict__ dst, int64_t Tpad_dst,
                                                       const int *__restrict__ sel, int64_t nsel) {
    const int64_t blk = blockIdx.x, lp = threadIdx.x;
    for (int64_t c = blockIdx.y; c < (Tpad_dst >> 2); c += gridDim.y) {
        float v[4];
        for (int i = 0; i < 4; ++i) {
            const int64_t j = c * 4 + i;
            float x = 0.f;
            if (j < nsel) { const int64_t t = sel[j]; x = src[((blk * (Tpad_src >> 2) + (t >> 2)) * BLKPX + lp) * 4 + (t & 3)]; }
            v[i] = x;
        }
        reinterpret_cast<float4 *>(dst)[(blk * (Tpad_dst >> 2) + c) * BLKPX + lp] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
// quantile(x, q) of the Statistics Toolbox as documented: the sorted values are the (0.5/n), (1.5/n), ... quantiles, linear
// interpolation between them, the extremes outside
static double matlab_quantile(std::vector<int> x, double q) {
    std::sort(x.begin(), x.end());
    const int64_t n = (int64_t)x.size();
    const double r = q * (double)n + 0.5;                 // 1-based fractional rank
    if (r <= 1.0) return x[0];
    if (r >= (double)n) return x[n - 1];
    const int64_t lo = (int64_t)std::floor(r);
    return x[lo - 1] + (r - (double)lo) * (double)(x[lo] - x[lo - 1]);
}

static int solve_launch(cnmfe_ctx *ctx, Patch *P, const CovTab &tab, const BgGeom &g, const double *rowsum, const unsigned char *act, int nt, int probe, const double *fill) {
    const unsigned ngrid = (unsigned)P->d;
    const int *pix = nullptr;
    int *dErr = nullptr;
    RET(ctx_errflag(ctx, &dErr));
    // one wave per pixel, the matrix in MFMA accumulator tiles (ring_solve.hpp).  Measured and removed (profiles/r02/solve_ab_c3.txt): the
    // panel-blocked LDS solver (22.5 ms against 8.0) and the looped-block-column variant (10.6 ms: it spills ~200 tile registers)
#define RS5_CASE(NT_) case NT_: LAUNCH(ctx, "bg_ring_solve", (k_ring_solve5<NT_>), dim3(ngrid), dim3(64), 0, tab, g, P->ring_dr.as<int>(), \
                                        P->ring_dc.as<int>(), rowsum, act, P->W.as<float>(), dErr, probe, pix, fill); break;
    switch (nt) { RS5_CASE(1) RS5_CASE(2) RS5_CASE(3) RS5_CASE(4) RS5_CASE(5) RS5_CASE(6) RS5_CASE(7) RS5_CASE(8) default: break; }
#undef RS5_CASE
    return 0;
}

// The large buffers of the ring fit, allocated when the ring is set (cnmfe_ring_init: once per patch, after the upload) instead of inside the first fit: the
// video's covariance table, the context's working table, the tiled Bf and the window projection's partial sums -- 18 GB at the headline size.  A fit then
// queues its kernels without a hipMalloc in between (tens of milliseconds of the first iteration, and on some boxes the dispatch behind a fresh multi-GB
// allocation stalled for 0.5-0.8 s, profiles/r03/README.md).  Sizes follow the geometry only; a buffer that is already large enough is left alone.
int bg_reserve(cnmfe_ctx *ctx, Patch *P) {
    if (ctx->opt("gram_incremental", 1) == 0 || P->p <= 0) return 0;
    int p_radius = 0;
    for (int i = 0; i < P->p; ++i) p_radius = std::max(p_radius, std::max(std::abs(P->dr[i]), std::abs(P->dc[i])));
    const int maxd = (2 * p_radius + 15) >> 4;
    if (maxd > 3) return 0;
    const int nbr = (P->nr_b + BLK - 1) / BLK, nbc = (P->nc_b + BLK - 1) / BLK, nblk = nbr * nbc;
    const int pi0 = std::max(0, P->roff - p_radius) / BLK, pi1 = std::min(P->nr_b - 1, P->roff + P->nr - 1 + p_radius) / BLK;
    const int pj0 = std::max(0, P->coff - p_radius) / BLK, pj1 = std::min(P->nc_b - 1, P->coff + P->nc - 1 + p_radius) / BLK;
    int64_t npairs = 0;                                      // as the pair list of bg_fit_ring counts them: touched blocks, canonical displacements within maxd
    for (int j = pj0; j <= pj1; ++j)
        for (int i = pi0; i <= pi1; ++i)
            for (int dC = 0; dC <= maxd; ++dC)
                for (int dR = (dC == 0 ? 0 : -maxd); dR <= maxd; ++dR) {
                    const int i2 = i + dR, j2 = j + dC;
                    if (i2 >= pi0 && i2 <= pi1 && j2 >= pj0 && j2 <= pj1) ++npairs;
                }
    const bool i8 = ctx->opt("gram_i8", 1) != 0 && P->T <= 24576;                   // (as bg_fit_ring decides for a stride-1 build: steps of four 16-frame stages)
    const int64_t Tpad = i8 ? (P->T + 4 * GK - 1) / (4 * GK) * (4 * GK) : (P->T + GK - 1) / GK * GK;
    const size_t tab = (size_t)npairs * BLKPX * BLKPX * sizeof(double);
    RET(P->cov_base.ensure(tab));
    RET(P->rowsum_base.ensure((size_t)nblk * BLKPX * sizeof(double)));
    RET(ctx->cov.ensure(tab));
    RET(ctx->rowsum.ensure((size_t)nblk * BLKPX * sizeof(double)));
    RET(ctx->bf.ensure((size_t)nblk * Tpad * BLKPX * sizeof(float)));
    const int nsg = nblk >= 512 ? std::max(1, std::min(8, (2048 + nblk - 1) / nblk)) : std::max(1, std::min(16, (4096 + nblk - 1) / std::max(1, nblk)));
    RET(ctx->inc[6].ensure((size_t)nsg * nblk * WIN_NLB * WIN_NLB * sizeof(double)));
    // the packed systems of the ring solve (ring_solve_packed.hpp), under the same conditions as the fit applies
    const int nt = (P->p + 15) / 16;
    if (ctx->opt("solve_packed", 1) != 0 && nt >= 1 && nt <= 8) {
        const size_t sys_bytes = (size_t)P->d * ((size_t)(nt * (nt + 1) / 2) * 256 + 16 * nt) * sizeof(double);
        if (P->sys.cap < sys_bytes) {
            size_t fr = 0, tot = 0;
            CK(hipMemGetInfo(&fr, &tot));
            if (fr >= sys_bytes + ((size_t)8 << 30)) { P->sys_valid = false; P->kinv_valid = false; P->kinv_lam_valid = false; P->kinv_fits = 0; RET(P->sys.ensure(sys_bytes)); }
        }
    }
    // round 5: the two further copies of the video (digit planes of the window projection, the temporal projection's read-order copy) under the rule their builders
    // apply -- one more video's worth each, only if that leaves 8 GB free.  Allocated HERE, while the caller sets its patches up: a fresh 10 GB hipMalloc takes 0.3 ms
    // on most leases and 1-3 SECONDS on some (profiles/r05/first_iteration_stall.txt: one run in five, inside the first iteration's temporal projection)
    // (AHEAD of time only while a quarter of the device stays free: the buffers every patch's fit and updates must have are allocated later, and 64 patches
    //  reserving down to the builders' 8 GB left none for them -- tests/test_gpu_zconfigs.py::test_c5_whole_on_one_gpu; below that the builders decide as before)
    auto reserve = [&](DevBuf &b, size_t bytes) -> int {
        if (b.cap >= bytes) return 0;
        size_t fr = 0, tot = 0;
        CK(hipMemGetInfo(&fr, &tot));
        if (fr >= bytes + std::max((size_t)8 << 30, tot / 4)) RET(b.ensure(bytes));
        return 0;
    };
    // (only a recording whose fits use every frame keeps digit planes: the stride floor(T / min(T, 100 pmax)) of fit_ring_model.m:84-87 starts at pmax = p and only grows)
    const bool stride1 = P->T / std::max<int64_t>(1, std::min<int64_t>(P->T, (int64_t)P->p * 100)) <= 1;
    if (!P->derived) {
        const bool planes = i8 && stride1 && ctx->opt("win_i8", 1) != 0;
        const bool pi8 = planes && ctx->opt("proj_i8", 1) != 0;      // the temporal projection on the int8 pipe reads pixel-major planes instead of the read-order copy
        if (planes) { RET(reserve(P->dig, (size_t)nblk * Tpad * BLKPX * sizeof(float))); RET(P->dig_sc.ensure((size_t)nblk * BLKPX * sizeof(double))); }
        if (ctx->opt("r1_virtual", 1) != 0) {
            if (pi8) RET(reserve(P->digp, (size_t)nblk * Tpad * BLKPX * sizeof(float)));
            else if (ctx->opt("proj_tiled", 1) != 0) RET(reserve(P->yt4, (size_t)nblk * ((P->Tc + 15) >> 4) * 64 * 64 * sizeof(float4)));
        }
    }
    return 0;
}

int bg_fit_ring(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val,
                const float *C, int c_order, int with_projection, float *b0_out, int64_t info[4], int b0_only, double thresh_outlier) {
    HostTrace ht(ctx, "fit_ring");
    const bool outl = thresh_outlier == thresh_outlier;      // ~isnan(thresh_outlier), :50
    if (outl && !P->sn_ready && !(b0_only & 1)) return fail(CNMFE_ESTATE, "fit_ring_model with thresh_outlier needs the noise levels of the block (cnmfe_set_noise)");
    const int64_t T = P->T;
    const int p = P->p;
    DevBuf &dC = ctx->tmp[0], &dCc = ctx->tmp[1], &dCm = ctx->tmp[2], &dArow = ctx->tmp[3], &dAcol = ctx->tmp[4], &dAval = ctx->tmp[5],
           &dMisc = ctx->tmp[6], &dAsum = ctx->tmp[7], &dActive = ctx->tmp[8], &dPairs = ctx->tmp[9], &dPairOf = ctx->tmp[10];
    // isempty(A) -> A = ones(d,1), C = zeros(1,T)  (fit_ring_model.m:15-17): Bf = Y - Ymean, b0 = Ymean
    const bool has_a = K > 0 && A_colptr[K] > 0;
    const bool a_empty = K == 0;
    int64_t ldc = 4;
    HostCSR csr;
    bool trace_i8_done = false;                                // the int8 window projection's trace digits and K x K Gram are already queued
    if (has_a) {
        RET(upload_centered(ctx, dC, C, K, T, c_order, dCc, dCm, &ldc));
        // win_proj_i8.hpp: what depends on the traces alone goes out NOW -- the host's footprint block lists (0.2 ms at the headline size) are built underneath it
        // instead of in front of an idle GPU (a fit that turns out to use a frame stride > 1 has queued 0.1 ms for nothing)
        if (P->dig_valid && !outl && T <= 24576 && ctx->opt("win_i8", 1) != 0 && ctx->opt("gram_incremental", 1) != 0 && !(P->base_valid && P->base_kstride > 1)) {      // (a strided fit keeps the fp64 window kernel)
            const int64_t T16 = P->dig_T16;
            RET(ctx->tdig.ensure((size_t)K * T16 * 4 * sizeof(uint4)));
            RET(ctx->tscale.ensure((size_t)K * sizeof(double)));
            RET(ctx->gk.ensure((size_t)K * K * sizeof(double)));
            LAUNCH(ctx, "bg_trace_dig", k_trace_dig, dim3(K), dim3(256), 0, dCc.as<float>(), ldc, (int64_t)T, T16, ctx->tdig.as<uint4>(), ctx->tscale.as<double>());
            const int ntk = ((int)K + 15) >> 4;
            LAUNCH(ctx, "bg_trace_gram", k_trace_gram, dim3((unsigned)(ntk * (ntk + 1) / 2)), dim3(64 * TG_W), 0, dCc.as<float>(), ldc, (int64_t)T, (int)K, ctx->gk.as<double>());
            trace_i8_done = true;
        }
    }
    // the CSR rows of A (b0, ind_active, the table corrections): built AFTER the window projection is queued when that can go first (below)
    auto upload_csr = [&]() -> int {
        if (!has_a) return 0;
        csc_to_csr(P->d_b, K, A_colptr, A_rowidx, A_val, csr);
        RET(to_dev(ctx, dArow, csr.rowptr.data(), csr.rowptr.size()));
        RET(to_dev(ctx, dAcol, csr.col.data(), csr.col.size()));
        RET(to_dev(ctx, dAval, csr.val.data(), csr.val.size()));
        return 0;
    };
    ht.mark("traces");
    if (b0_only & 1) {
        RET(upload_csr());                                    // bg_ssub > 1: b0 = mean(Y - A*C, 2) on the patch (update_background_parallel.m:222-223)
        BgGeom g0{};
        g0.nr = P->nr; g0.nc = P->nc; g0.nr_b = P->nr_b; g0.nc_b = P->nc_b; g0.roff = P->roff; g0.coff = P->coff; g0.d = P->d; g0.d_b = P->d_b;
        LAUNCH(ctx, "bg_b0", k_b0, dim3((unsigned)((P->d + 255) / 256)), dim3(256), 0, P->ymean_d.as<double>(), g0,
               has_a ? dArow.as<int>() : nullptr, dAcol.as<int>(), dAval.as<float>(), dCm.as<double>(), P->b0.as<double>());
        // (no drain: every upload above went through the pinned arena -- or, too large for it, waited itself -- and what reads b0 is stream-ordered behind this)
        P->ysig_valid = false;
        return 0;
    }
    // ---- first-run test (:25): row 1 of W_old has exactly two distinct values (0 and 1/count) ----
    // Both questions about W_old were answered behind the call that produced it (ring_stats_enqueue): no drain of the stream here, so with
    // several patches per context the host sets up patch m + 1 while the GPU still solves patch m.
    bool first_run = false;
    int pmax = 0;
    RET(ring_stats_get(ctx, P, &pmax, &first_run));
    RET(dMisc.ensure(64));
    ht.mark("first_run + pmax");
    int kstride = 1;
    if (with_projection) {                                // :84-87
        int64_t nk = std::min<int64_t>(T, (int64_t)pmax * 100);
        if (nk < 1) nk = 1;
        kstride = (int)(T / nk);
        if (kstride < 1) kstride = 1;
    }
    // with a threshold the frames are SELECTED instead (:62-67) and nmax = nnz(ind_frames) = size(Bf, 2) makes k = 1 at :84-85
    // (also when nmax >= T, where nk = min(T, nmax) = T)
    if (outl) kstride = 1;
    BgGeom g;
    g.nr = P->nr; g.nc = P->nc; g.nr_b = P->nr_b; g.nc_b = P->nc_b; g.roff = P->roff; g.coff = P->coff;
    g.r0_abs = P->brect[0]; g.c0_abs = P->brect[2]; g.d1 = P->d1; g.d2 = P->d2;
    g.nbr = (P->nr_b + BLK - 1) / BLK; g.nbc = (P->nc_b + BLK - 1) / BLK;
    g.d = P->d; g.d_b = P->d_b; g.T = T; g.kstride = kstride;
    g.Tp = (T + kstride - 1) / kstride;                   // numel(1:k:T)
    g.nbr = (P->nr_b + BLK - 1) / BLK; g.nbc = (P->nc_b + BLK - 1) / BLK;
    // ---- incremental Gram (k_win_proj / k_cov_correct above): per 16x16 block, the footprints with a pixel within two blocks of it ----
    g.p_radius = 0;
    for (int i = 0; i < P->p; ++i) g.p_radius = std::max(g.p_radius, std::max(std::abs(P->dr[i]), std::abs(P->dc[i])));
    g.nbw = ((2 * g.p_radius) >> 4) + 2;
    g.p = p;
    // two ring pixels of one centre are up to 2*p_radius apart along an axis: their 16x16 blocks up to maxd apart (2 for radius <= 16, 3 up to 24)
    const int maxd = (2 * g.p_radius + 15) >> 4, nrel = nrel_of(maxd);
    if (maxd > 3 || g.nbw > 4) return fail(CNMFE_EUNSUPPORTED, "fit_ring_model: ring offsets up to %d pixels (the block-pair table covers <= 24)", g.p_radius);
    bool incr = !outl && ctx->opt("gram_incremental", 1) != 0 && K < 32768;   // the clipped Bf is not linear in the video: direct Gram   // (derived low-resolution patches of bg_ssub included: their video is built once)
    std::vector<int> lst_ptr, lst_k, blk_nt[4];
    std::vector<short> slot_of;
    static thread_local std::vector<int> nbox;               // 4 per neuron: first / last row, first / last column of its footprint in the block region
    if (incr) {
        // (flat arrays, two passes: a vector of vectors cost 0.4 ms of allocator churn per fit at the headline size, in front of the window projection)
        const int nblk_ = g.nbr * g.nbc;
        static thread_local std::vector<int> own, seen, cnt, pairs_b, pairs_k, mark;
        own.assign(nblk_, -1); seen.assign(nblk_, -1); cnt.assign(nblk_ + 1, 0);
        pairs_b.clear(); pairs_k.clear();
        nbox.assign((size_t)4 * std::max(1, K), 0);
        for (int k = 0; k < K && has_a; ++k) {
            mark.clear();
            int bx[4] = {1 << 30, -1, 1 << 30, -1};                                 // the footprint's bounding box in the block region: rows, columns
            // (the entries of a column ascend with the pixel index: one step per run of an image column of the block region, not per entry -- this list
            //  building sits in front of the window projection's launch)
            for (int64_t e = A_colptr[k]; e < A_colptr[k + 1];) {
                const int q0 = A_rowidx[e], cb = q0 / P->nr_b, col_end = (cb + 1) * P->nr_b;
                int qlast = q0;
                ++e;
                while (e < A_colptr[k + 1] && A_rowidx[e] < col_end && A_rowidx[e] >= qlast) { qlast = A_rowidx[e]; ++e; }
                bx[0] = std::min(bx[0], q0 - cb * P->nr_b); bx[1] = std::max(bx[1], qlast - cb * P->nr_b); bx[2] = std::min(bx[2], cb); bx[3] = std::max(bx[3], cb);
                for (int bi = (q0 - cb * P->nr_b) >> 4; bi <= (qlast - cb * P->nr_b) >> 4; ++bi) {
                    const int b_ = (cb >> 4) * g.nbr + bi;
                    if (own[b_] != k) { own[b_] = k; mark.push_back(b_); }
                }
            }
            for (int i = 0; i < 4; ++i) nbox[(size_t)4 * k + i] = bx[i];
            for (int b_ : mark)
                for (int dj = -maxd; dj <= maxd; ++dj)
                    for (int di = -maxd; di <= maxd; ++di) {
                        const int i2 = b_ % g.nbr + di, j2 = b_ / g.nbr + dj;
                        if (i2 < 0 || i2 >= g.nbr || j2 < 0 || j2 >= g.nbc) continue;
                        const int nb_ = j2 * g.nbr + i2;
                        if (seen[nb_] != k) { seen[nb_] = k; pairs_b.push_back(nb_); pairs_k.push_back(k); ++cnt[nb_ + 1]; }
                    }
        }
        lst_ptr.assign(nblk_ + 1, 0);
        for (int b_ = 0; b_ < nblk_; ++b_) {
            if (cnt[b_ + 1] > WIN_NLB) { incr = false; break; }                     // denser than the window kernel is built for: direct Gram
            lst_ptr[b_ + 1] = lst_ptr[b_] + cnt[b_ + 1];
        }
        if (incr) {
            lst_k.resize(pairs_k.size());
            slot_of.assign((size_t)nblk_ * std::max(1, K), (short)-1);
            std::vector<int> &fill = cnt;                                           // next free slot per block
            for (int b_ = 0; b_ < nblk_; ++b_) fill[b_] = 0;
            for (size_t i = 0; i < pairs_k.size(); ++i) {                           // pairs are k-major: every list comes out in ascending k, as before
                const int b_ = pairs_b[i], k = pairs_k[i], s_ = fill[b_]++;
                lst_k[lst_ptr[b_] + s_] = k; slot_of[(size_t)b_ * K + k] = (short)s_;
            }
            for (int b_ = 0; b_ < nblk_; ++b_) { const int n = lst_ptr[b_ + 1] - lst_ptr[b_]; if (n) blk_nt[(n - 1) >> 4].push_back(b_); }
        }
    }
    if (incr && P->base_valid && P->base_kstride != kstride) {
        // the table in place is of another stride: it moves to the second slot, and what that slot held -- the table of THIS stride, if the recording
        // alternates -- comes forward (otherwise its memory is where the new table is built)
        P->cov_base.swap(P->cov_base_alt); P->rowsum_base.swap(P->rowsum_base_alt);
        std::swap(P->base_valid, P->base_alt_valid); std::swap(P->base_kstride, P->base_alt_kstride);
        P->sys.swap(P->sys_alt); std::swap(P->sys_valid, P->sys_alt_valid);          // (the packed systems belong to their table)
        P->kinv_valid = false; P->kinv_lam_valid = false; P->kinv_fits = 0;          // (and the inverses to the packed systems)
    }
    const bool build_base = incr && !(P->base_valid && P->base_kstride == kstride);
    if (build_base) { P->sys_valid = false; P->kinv_valid = false; P->kinv_lam_valid = false; P->kinv_fits = 0; }
    ht.mark("footprint block lists");
    g.bf4 = 1;
    // round 5 (gram_i8.hpp): the Gram -- the VIDEO's table of the incremental path, or the direct Gram of Bf where that path does not apply -- on the int8 matrix
    // pipe, exact up to a 32-bit quantisation of the data (option gram_i8, default 1; int32 range: <= 24576 used frames).  The outlier branch (a clipped,
    // frame-selected Bf) keeps the fp64 kernel.
    const bool use_i8 = (build_base || !incr) && !outl && ctx->opt("gram_i8", 1) != 0 && g.Tp <= 24576;
    if (use_i8) g.bf4 = 3;
    g.Tpad = g.bf4 == 3 ? (g.Tp + 4 * GK - 1) / (4 * GK) * (4 * GK) : (g.Tp + GK - 1) / GK * GK;   // int8: steps of four 16-frame stages
    const int nblk = g.nbr * g.nbc;

    // ---- the window projection first: it needs the footprint lists and the traces, nothing else, and takes 5 ms at the headline size -- the host
    // builds the CSR rows of A, ind_active and the pair tables underneath it (a later fit of a patch: the video's table exists)
    // keep_pt: this fit's window projection is the P = Yc Cc' table the next spatial update wants (all frames, the block's neurons, the same centred traces):
    // the raw sums are kept with the patch, and so are the lists that index them (then the fit works out of the patch's copies)
    const bool keep_pt = incr && has_a && kstride == 1 && !P->derived && !(b0_only & 2) && ctx->opt("r1_virtual", 1) != 0;
    DevBuf &dLp = keep_pt ? P->pt_lp : ctx->inc[0], &dLk = ctx->inc[1], &dSlot = keep_pt ? P->pt_slot : ctx->inc[2], &dBl = ctx->inc[3], &dUt = ctx->inc[4], &dCsum = ctx->inc[5], &dGb = ctx->inc[6];
    if (keep_pt) P->pt_valid = false;                        // (until the new table is queued)
    std::vector<int> blall;
    int nsg = 1; int64_t ut_stride = 0, gb_stride = 0;
    bool proj_queued = false;
    bool stage_ok = false;                                     // ring_solve_staged.hpp: this fit's neuron windows are built
    bool win_i8 = false;                                       // the window projection ran on the int8 pipe: k_win_fix takes G from the K x K matrix
    auto queue_projection = [&]() -> int {
        for (int t = 3; t >= 0; --t) blall.insert(blall.end(), blk_nt[t].begin(), blk_nt[t].end());      // longest lists first
        RET(to_dev(ctx, dLp, lst_ptr.data(), lst_ptr.size()));
        RET(to_dev(ctx, dLk, lst_k.data(), lst_k.size()));
        RET(to_dev(ctx, dBl, blall.data(), blall.size()));
        RET(dCsum.ensure((size_t)K * sizeof(double)));
        LAUNCH(ctx, "bg_trace_subsum", k_trace_subsum, dim3(K), dim3(256), 0, dCc.as<float>(), ldc, g.Tp, g.kstride, dCsum.as<double>());
        const int nb_ = (int)blall.size();
        // frame segments per block: enough workgroups to fill the chip; small patches (few blocks) get more, shorter segments
        nsg = nb_ >= 512 ? std::max(1, std::min(8, (2048 + nb_ - 1) / nb_)) : std::max(1, std::min(16, (4096 + nb_ - 1) / std::max(1, nb_)));
        ut_stride = (int64_t)std::max<size_t>(1, lst_k.size()) * BLKPX; gb_stride = (int64_t)nblk * WIN_NLB * WIN_NLB;
        RET(dUt.ensure((size_t)nsg * ut_stride * sizeof(double)));
        RET(dGb.ensure((size_t)nsg * gb_stride * sizeof(double)));
        if (P->dig_valid && g.kstride == 1 && g.Tp <= 24576 && ctx->opt("win_i8", 1) != 0 && K > 0) {
            // round 5 (win_proj_i8.hpp): the same sums on the int8 matrix pipe out of the resident digit planes; G = Cc Cc' once, K x K
            // (round 6 ran a strided fit here too -- planes of every frame, trace digits zeroed on the skipped frames: correct, and no faster than the fp64 kernel on
            //  patches of 128 x 128, which reads its 2 GB at 5.5 TB/s already, while the K x K trace Gram became a kernel of its own: profiles/r06/c5shard_i8.txt)
            const int64_t T16 = P->dig_T16;
            if (!trace_i8_done) {                                // (the first fit of a patch: the digit planes did not exist when the traces went up)
                RET(ctx->tdig.ensure((size_t)K * T16 * 4 * sizeof(uint4)));
                RET(ctx->tscale.ensure((size_t)K * sizeof(double)));
                RET(ctx->gk.ensure((size_t)K * K * sizeof(double)));
                LAUNCH(ctx, "bg_trace_dig", k_trace_dig, dim3(K), dim3(256), 0, dCc.as<float>(), ldc, (int64_t)T, T16, ctx->tdig.as<uint4>(), ctx->tscale.as<double>());
                const int ntk = ((int)K + 15) >> 4;
                LAUNCH(ctx, "bg_trace_gram", k_trace_gram, dim3((unsigned)(ntk * (ntk + 1) / 2)), dim3(64 * TG_W), 0, dCc.as<float>(), ldc, (int64_t)T, (int)K, ctx->gk.as<double>());
            }
            std::vector<int> items;
            for (int b_ : blall) {
                const int ntl = (lst_ptr[b_ + 1] - lst_ptr[b_] + 15) >> 4;
                for (int gq = 0; gq < (ntl + 1) / 2; ++gq) items.push_back(b_ | (gq << 24));
            }
            RET(to_dev(ctx, ctx->win_items, items.data(), items.size()));
            if (!items.empty())
                if (win_planes3(ctx, T16)) { LAUNCH(ctx, "bg_win_proj", k_win_proj_i8<1>, dim3((unsigned)(items.size() * nsg)), dim3(512), 0, P->dig.as<uint4>(), T16, P->dig_sc.as<double>(), ctx->tdig.as<uint4>(),
                       ctx->tscale.as<double>(), dLp.as<int>(), dLk.as<int>(), ctx->win_items.as<int>(), nsg, dUt.as<double>(), ut_stride); } else { LAUNCH(ctx, "bg_win_proj", k_win_proj_i8<0>, dim3((unsigned)(items.size() * nsg)), dim3(512), 0, P->dig.as<uint4>(), T16, P->dig_sc.as<double>(), ctx->tdig.as<uint4>(),
                       ctx->tscale.as<double>(), dLp.as<int>(), dLk.as<int>(), ctx->win_items.as<int>(), nsg, dUt.as<double>(), ut_stride); }
            win_i8 = true;
        } else if (g.kstride == 1 || g.kstride == 2 || g.kstride == 4) {
            const int nbig = (int)blk_nt[3].size();              // blall starts with the longest lists
            if (nbig)
                LAUNCH(ctx, "bg_win_proj", k_win_proj4<true>, dim3((unsigned)(nbig * nsg)), dim3(256), 0, P->Yc4.as<float4>(), g, dCc.as<float>(), ldc, dLp.as<int>(), dLk.as<int>(),
                       dBl.as<int>(), nsg, dUt.as<double>(), ut_stride, dGb.as<double>(), gb_stride);
            if (nb_ > nbig)
                LAUNCH(ctx, "bg_win_proj", k_win_proj4<false>, dim3((unsigned)((nb_ - nbig) * nsg)), dim3(256), 0, P->Yc4.as<float4>(), g, dCc.as<float>(), ldc, dLp.as<int>(),
                       dLk.as<int>(), dBl.as<int>() + nbig, nsg, dUt.as<double>(), ut_stride, dGb.as<double>(), gb_stride);
        } else
        LAUNCH(ctx, "bg_win_proj", k_win_proj, dim3((unsigned)(nb_ * nsg)), dim3(256), 0, P->Yc4.as<float4>(), g, dCc.as<float>(), ldc, dLp.as<int>(), dLk.as<int>(),
               dBl.as<int>(), nsg, dUt.as<double>(), ut_stride, dGb.as<double>(), gb_stride);
        proj_queued = true;
        return 0;
    };
    if (incr && has_a && !build_base) RET(queue_projection());
    ht.mark("projection queued");
    RET(upload_csr());
    ht.mark("A csr");

    // ---- ind_active (:25-29) ----
    RET(dActive.ensure(P->d));
    int64_t nactive = P->d;
    if (first_run) {
        CK(hipMemsetAsync(dActive.p, 1, P->d, ctx->st()));
    } else {
        std::vector<float> asum(P->d_b, 0.f);
        // isempty(A) -> A = ones(d,1) (:14-16).  Bit 1 of b0_only: the reference passes A = [] because it subtracted A*C itself (bg_ssub > 1)
        if (a_empty || (b0_only & 2)) std::fill(asum.begin(), asum.end(), 1.0f);
        else if (has_a) for (int64_t q = 0; q < P->d_b; ++q) { double s = 0; for (int64_t e = csr.rowptr[q]; e < csr.rowptr[q + 1]; ++e) s += csr.val[e]; asum[q] = (float)s; }
        RET(to_dev(ctx, dAsum, asum.data(), asum.size()));
        CK(hipMemsetAsync(dMisc.p, 0, 64, ctx->st()));
        LAUNCH(ctx, "bg_active", k_active, dim3((unsigned)((P->d + 255) / 256)), dim3(256), 0, P->W.as<float>(), g,
               P->ring_dr.as<int>(), P->ring_dc.as<int>(), dAsum.as<float>(), dActive.as<unsigned char>(), dMisc.as<int>());
        // the count is only reported (info[2]): it lands in pinned memory and is read if the call ends with a drain anyway (b0_out)
        CK(hipMemcpyAsync((char *)P->stat_host + 8, dMisc.p, sizeof(int), hipMemcpyDeviceToHost, ctx->st()));
        nactive = -1;
    }
    // ---- b0 (:44) ----
    LAUNCH(ctx, "bg_b0", k_b0, dim3((unsigned)((P->d + 255) / 256)), dim3(256), 0, P->ymean_d.as<double>(), g,
           has_a ? dArow.as<int>() : nullptr, dAcol.as<int>(), dAval.as<float>(), dCm.as<double>(), P->b0.as<double>());

    ht.mark("ind_active + b0");
    {   // (pixels that are not active keep their weights: the solve kernel skips them; a patch without any costs the sweep of its tables)
        // ---- pair list: blocks that hold ring pixels of some patch pixel (or patch pixels), displacement within +-maxd ----
        std::vector<char> touched(nblk, 0);
        {
            const int pi0 = std::max(0, P->roff - g.p_radius) / BLK, pi1 = std::min(P->nr_b - 1, P->roff + P->nr - 1 + g.p_radius) / BLK;
            const int pj0 = std::max(0, P->coff - g.p_radius) / BLK, pj1 = std::min(P->nc_b - 1, P->coff + P->nc - 1 + g.p_radius) / BLK;
            for (int j = pj0; j <= pj1; ++j)
                for (int i = pi0; i <= pi1; ++i) touched[j * g.nbr + i] = 1;
        }
        std::vector<int4> pairs; std::vector<int> pair_of((size_t)nblk * nrel, -1);
        for (int j = 0; j < g.nbc; ++j)
            for (int i = 0; i < g.nbr; ++i) {
                if (!touched[j * g.nbr + i]) continue;
                for (int dC = 0; dC <= maxd; ++dC)
                    for (int dR = (dC == 0 ? 0 : -maxd); dR <= maxd; ++dR) {
                        int i2 = i + dR, j2 = j + dC;
                        if (i2 < 0 || i2 >= g.nbr || j2 >= g.nbc || !touched[j2 * g.nbr + i2]) continue;
                        const int rel = rel_index(dR, dC, maxd);
                        pair_of[(size_t)(j * g.nbr + i) * nrel + rel] = (int)pairs.size();
                        pairs.push_back(make_int4(j * g.nbr + i, j2 * g.nbr + i2, rel, 0));
                    }
            }
        const int npairs = (int)pairs.size();
        // needed 16x16 sub-tiles: displacement set D = (O - O) u O u -O of the ring offsets O
        const int DB = 2 * g.p_radius, DM = 2 * DB + 1;
        std::vector<char> Dm((size_t)DM * DM, 0);
        auto dset = [&](int dr, int dc) { if (dr >= -DB && dr <= DB && dc >= -DB && dc <= DB) Dm[(size_t)(dr + DB) * DM + dc + DB] = 1; };
        for (int a = 0; a < p; ++a) {
            dset(P->dr[a], P->dc[a]); dset(-P->dr[a], -P->dc[a]);
            for (int b = 0; b < p; ++b) dset(P->dr[b] - P->dr[a], P->dc[b] - P->dc[a]);
        }
        std::vector<int> rel_dR(nrel), rel_dC(nrel);
        for (int dC = 0; dC <= maxd; ++dC)
            for (int dR = (dC == 0 ? 0 : -maxd); dR <= maxd; ++dR) { rel_dR[rel_index(dR, dC, maxd)] = dR; rel_dC[rel_index(dR, dC, maxd)] = dC; }
        std::vector<unsigned short> needmask((size_t)nrel * 16, 0);
        for (int rel = 0; rel < nrel; ++rel) {
            const int dC = rel_dC[rel], dR = rel_dR[rel];
            for (int pi = 0; pi < 16; ++pi)
                for (int pj = 0; pj < 16; ++pj) {
                    if (rel == 0 && pi > pj) continue;       // self pair: upper patch triangle (cov_lookup swaps)
                    const int ri = (pi & 3) * 4, ci = (pi >> 2) * 4, rj = (pj & 3) * 4 + dR * 16, cj = (pj >> 2) * 4 + dC * 16;
                    bool nd = false;
                    for (int x = -3; x <= 3 && !nd; ++x)
                        for (int y = -3; y <= 3; ++y) {
                            const int ddr = rj - ri + x, ddc = cj - ci + y;
                            if (ddr >= -DB && ddr <= DB && ddc >= -DB && ddc <= DB && Dm[(size_t)(ddr + DB) * DM + ddc + DB]) { nd = true; break; }
                        }
                    if (nd) needmask[rel * 16 + pi] |= (unsigned short)(1u << pj);
                }
        }
        // work order: pair-major (the 13 displacement classes of one I block are consecutive), so heavy and light
        // quadrants are mixed in time.  (Measured: class-major order, which makes concurrent workgroups equal-cost,
        // was slower -- 150 vs 131 ms at 512x512x10000 -- and did not raise the L2 hit rate.)
        std::vector<int> work;
        for (int pp = 0; pp < npairs; ++pp)
            for (int q = 0; q < 4; ++q) {
                const int ih = q & 1, jh = q >> 1;
                bool any = false;
                for (int pi = ih * 8; pi < ih * 8 + 8; ++pi) if ((needmask[pairs[pp].z * 16 + pi] >> (jh * 8)) & 0xff) any = true;
                if (any) work.push_back(pp * 4 + q);
            }
        const int nwork = (int)work.size();
    ht.mark("pair / need / work tables");
        DevBuf &dWork = ctx->tmp[11], &dNeed = ctx->tmp[7];
        RET(to_dev(ctx, dPairs, pairs.data(), pairs.size()));
        RET(to_dev(ctx, dPairOf, pair_of.data(), pair_of.size()));
        RET(to_dev(ctx, dWork, work.data(), work.size()));
        RET(to_dev(ctx, dNeed, needmask.data(), needmask.size()));
        RET(ctx->cov.ensure((size_t)npairs * BLKPX * BLKPX * sizeof(double)));
        if (ctx->opt("debug", 0)) CK(hipMemsetAsync(ctx->cov.p, 0xff, (size_t)npairs * BLKPX * BLKPX * sizeof(double), ctx->st()));   // NaN-poison skipped sub-tiles
        int nwg = (nwork + 7) / 8 * 8;                      // multiple of 8 for the XCD remap (extra workgroups exit)
        DevBuf &dTcnt = ctx->tmp[12], &dTl = ctx->tmp[13];
        {
            // tile lists per (displacement class, quadrant): needed 16x16 sub-tiles as i | j << 4 (quadrant coordinates)
            std::vector<int> tcnt((size_t)nrel * 4, 0);
            std::vector<int> tlist((size_t)nrel * 4 * 64, 0);
            for (int rel = 0; rel < nrel; ++rel)
                for (int q = 0; q < 4; ++q) {
                    const int ih = q & 1, jh = q >> 1;
                    int n = 0;
                    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j)
                        if ((needmask[rel * 16 + ih * 8 + i] >> (jh * 8 + j)) & 1) tlist[(size_t)(rel * 4 + q) * 64 + n++] = i | (j << 4);
                    tcnt[rel * 4 + q] = n;
                }
            RET(to_dev(ctx, dTcnt, tcnt.data(), tcnt.size()));
            RET(to_dev(ctx, dTl, tlist.data(), tlist.size()));
        }
        // (every table above went through the pinned arena: no drain here, the big launches below queue behind whatever the stream still holds)
        ht.mark("tile lists + uploads (sync)");
        // incremental: the table of the video alone is built once (fp64 pipe) and kept with the patch; later fits skip B1 / B2a entirely
        DevBuf &covT = incr ? P->cov_base : ctx->cov, &rsT = incr ? P->rowsum_base : ctx->rowsum;
        const bool has_a_bf = has_a && !incr;
        if (incr) {
            RET(P->cov_base.ensure((size_t)npairs * BLKPX * BLKPX * sizeof(double)));
            if (build_base && ctx->opt("debug", 0)) CK(hipMemsetAsync(P->cov_base.p, 0xff, (size_t)npairs * BLKPX * BLKPX * sizeof(double), ctx->st()));
        }
        if (!incr || build_base) {
        // ---- B1: Bf tiled ----
        RET(ctx->bf.ensure((size_t)nblk * g.Tpad * BLKPX * sizeof(float)));
        const int tchunk = (int)((std::max<int64_t>(64, (g.Tpad + 15) / 16) + 7) & ~int64_t(7));
        dim3 gb(nblk, (unsigned)((g.Tpad + tchunk - 1) / tchunk));
        RET(rsT.ensure((size_t)nblk * BLKPX * sizeof(double)));
        uint4 *digp = nullptr; double *digs = nullptr;          // where the digit planes of this build live
        if (use_i8) {
            // the VIDEO's planes stay resident with the patch (the fits' window projection reads them, win_proj_i8.hpp) when one more video's worth of memory
            // leaves 8 GB free and the stride is 1; otherwise, and for the direct Gram of Bf, in the context's scratch
            const size_t dbytes = (size_t)nblk * g.Tpad * BLKPX * sizeof(float);
            bool resident = incr && kstride == 1 && !P->derived && ctx->opt("win_i8", 1) != 0;
            if (resident && P->dig.cap < dbytes) {
                size_t fr = 0, tot = 0;
                CK(hipMemGetInfo(&fr, &tot));
                if (fr < dbytes + ((size_t)8 << 30)) resident = false;
            }
            if (resident) {
                RET(P->dig.ensure(dbytes)); RET(P->dig_sc.ensure((size_t)nblk * BLKPX * sizeof(double)));
                digp = P->dig.as<uint4>(); digs = P->dig_sc.as<double>();
                P->dig_T16 = g.Tpad >> 4; P->dig_valid = true; P->digp_valid = false;
            } else {
                RET(ctx->dig_scale.ensure((size_t)nblk * BLKPX * sizeof(double)));
                digp = ctx->bf.as<uint4>(); digs = ctx->dig_scale.as<double>();
            }
            RET(ctx->dig_smax.ensure((size_t)nblk * BLKPX * sizeof(unsigned)));
            CK(hipMemsetAsync(ctx->dig_smax.p, 0, (size_t)nblk * BLKPX * sizeof(unsigned), ctx->st()));
            DigA da{has_a_bf ? dArow.as<int>() : nullptr, dAcol.as<int>(), dAval.as<float>(), dCc.as<float>(), ldc};
            const int tchunk16 = (int)((std::max<int64_t>(64, (g.Tpad + 15) / 16) + 15) & ~int64_t(15));
            const dim3 gd(nblk, (unsigned)((g.Tpad + tchunk16 - 1) / tchunk16));
            LAUNCH(ctx, "bg_dig_scale", k_dig_scale, gd, dim3(256), 0, P->Yc4.as<float4>(), P->Tc, g, da, tchunk16, ctx->dig_smax.as<unsigned>());
            RET(ctx->dig_rspart.ensure((size_t)gd.y * nblk * BLKPX * sizeof(double)));
            LAUNCH(ctx, "bg_build_dig", k_build_dig, gd, dim3(256), 0, P->Yc4.as<float4>(), P->Tc, g, da, ctx->dig_smax.as<unsigned>(), digs, digp, tchunk16, ctx->dig_rspart.as<double>());
            LAUNCH(ctx, "bg_rs_reduce", k_rs_reduce, dim3((unsigned)nblk), dim3(256), 0, ctx->dig_rspart.as<double>(), (int)gd.y, (int64_t)nblk * BLKPX, rsT.as<double>());
            // round 6: a fit on every kstride-th frame (T > 100 pmax: BASELINE configs[4], T = 20000) built the planes of ITS frames above, in the scratch, for the table.
            // The planes the spatial update's table and the temporal projection read hold EVERY frame and are built here, once per recording, under the same memory
            // rule (until round 5 a strided recording ran both on the fp64 pipe: 2.9-3.0 ms each for a rank's share of configs[4], profiles/r05/bench_c5shard_v3.json;
            // 2.5-2.6 ms now -- at that patch size all three video passes move their bytes at 5.5 TB/s whatever the pipe)
            // (round 6, late: up to I8_SEG_FRAMES * 16 frames -- the projections that contract over FRAMES keep every int32 sum inside one frame segment of at most
            //  I8_SEG_FRAMES frames, vproj.hip; the temporal projection contracts over pixels and has no such limit)
            if (incr && kstride > 1 && !resident && !P->derived && !has_a_bf && T <= (int64_t)I8_SEG_FRAMES * 16 && ctx->opt("win_i8", 1) != 0) {
                BgGeom g1 = g; g1.kstride = 1; g1.Tp = T; g1.Tpad = (T + 4 * GK - 1) / (4 * GK) * (4 * GK);
                const size_t d1bytes = (size_t)nblk * g1.Tpad * BLKPX * sizeof(float);
                bool ok = true;
                if (P->dig.cap < d1bytes) { size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); if (fr < d1bytes + ((size_t)8 << 30)) ok = false; }
                if (ok) {
                    RET(P->dig.ensure(d1bytes)); RET(P->dig_sc.ensure((size_t)nblk * BLKPX * sizeof(double)));
                    CK(hipMemsetAsync(ctx->dig_smax.p, 0, (size_t)nblk * BLKPX * sizeof(unsigned), ctx->st()));
                    const int tch1 = (int)((std::max<int64_t>(64, (g1.Tpad + 15) / 16) + 15) & ~int64_t(15));
                    const dim3 gd1(nblk, (unsigned)((g1.Tpad + tch1 - 1) / tch1));
                    DigA dv{nullptr, nullptr, nullptr, nullptr, 0};
                    LAUNCH(ctx, "bg_dig_scale", k_dig_scale, gd1, dim3(256), 0, P->Yc4.as<float4>(), P->Tc, g1, dv, tch1, ctx->dig_smax.as<unsigned>());
                    LAUNCH(ctx, "bg_build_dig", k_build_dig, gd1, dim3(256), 0, P->Yc4.as<float4>(), P->Tc, g1, dv, ctx->dig_smax.as<unsigned>(), P->dig_sc.as<double>(), P->dig.as<uint4>(), tch1,
                           (double *)nullptr);
                    P->dig_T16 = g1.Tpad >> 4; P->dig_valid = true; P->digp_valid = false;
                }
            }
        } else
        LAUNCH(ctx, "bg_build_bf", k_build_bf, gb, dim3(256), 0, P->Yc4.as<float4>(), P->Tc, g,
               has_a_bf ? dArow.as<int>() : nullptr, dAcol.as<int>(), dAval.as<float>(), dCc.as<float>(), ldc, ctx->bf.as<float>(), tchunk, rsT.as<double>());
        if (outl) {
            // ---- :50-56 clip, :60-67 frame selection ----
            const size_t bfbytes = (size_t)nblk * g.Tpad * BLKPX * sizeof(float);
            RET(ctx->bf2.ensure(bfbytes));
            RET(ctx->outl_cnt.ensure((size_t)T * sizeof(int)));
            CK(hipMemcpyAsync(ctx->bf2.p, ctx->bf.p, bfbytes, hipMemcpyDeviceToDevice, ctx->st()));
            CK(hipMemsetAsync(ctx->outl_cnt.p, 0, (size_t)T * sizeof(int), ctx->st()));
            LAUNCH(ctx, "bg_outlier_clip", k_outlier_clip, dim3((unsigned)((P->d + 255) / 256), (unsigned)(g.Tpad >> 2)), dim3(256), 0, ctx->bf.as<float4>(),
                   ctx->bf2.as<float4>(), g, P->ring_dr.as<int>(), P->ring_dc.as<int>(), P->W.as<float>(), P->sn_b.as<float>(), thresh_outlier,
                   ctx->outl_cnt.as<int>());
            std::vector<int> sel;
            const int64_t nmax = (int64_t)pmax * 100;                       // :61
            if (nmax < T) {                                                 // :62
                std::vector<int> cnt((size_t)T);
                CK(hipMemcpyAsync(cnt.data(), ctx->outl_cnt.p, (size_t)T * sizeof(int), hipMemcpyDeviceToHost, ctx->st()));
                CK(hipStreamSynchronize(ctx->st()));
                const double qv = matlab_quantile(cnt, (double)nmax / (double)T);   // :64
                for (int64_t t = 0; t < T; ++t) if ((double)cnt[t] <= qv) sel.push_back((int)t);
            } else for (int64_t t = 0; t < T; ++t) sel.push_back((int)t);
            const int64_t Tpad_src = g.Tpad;
            g.Tp = (int64_t)sel.size();
            g.Tpad = (g.Tp + GK - 1) / GK * GK;
            RET(to_dev(ctx, ctx->outl_sel, sel.data(), sel.size()));
            LAUNCH(ctx, "bg_select_frames", k_select_frames, dim3(nblk, (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, g.Tpad >> 2))), dim3(256), 0,
                   ctx->bf2.as<float>(), Tpad_src, ctx->bf.as<float>(), g.Tpad, ctx->outl_sel.as<int>(), g.Tp);
            CK(hipStreamSynchronize(ctx->st()));                           // `sel` is staged from this scope
        }
        if (g.bf4 < 2)
            LAUNCH(ctx, "bg_rowsum", k_rowsum, dim3(nblk), dim3(256), 0, ctx->bf.as<float>(), g.Tpad, rsT.as<double>(), g.bf4);

        {
            const size_t shmem = (size_t)G4_NBUF * G4_STAGE_F * sizeof(float);
            static bool attr4 = false;
            if (!attr4) {
                CK(hipFuncSetAttribute((const void *)k_gram4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
                attr4 = true;
            }
            if (use_i8) {
                static bool attr8 = false;
                if (!attr8) { CK(hipFuncSetAttribute((const void *)k_gram_i8, hipFuncAttributeMaxDynamicSharedMemorySize, GI_NBUF * GI_STAGE_B)); attr8 = true; }
                LAUNCH(ctx, "bg_gram_i8", k_gram_i8, dim3(nwg), dim3(512), (size_t)GI_NBUF * GI_STAGE_B, digp, g.Tpad >> 4, dPairs.as<int4>(), dWork.as<int>(), nwork,
                       dTcnt.as<int>(), dTl.as<int>(), digs, covT.as<double>());
            } else
                LAUNCH(ctx, "bg_gram_f64", k_gram4, dim3(nwg), dim3(256), shmem, ctx->bf.as<float>(), g.Tpad, dPairs.as<int4>(), dWork.as<int>(), nwork,
                       dTcnt.as<int>(), dTl.as<int>(), 0, covT.as<double>());
        }
        if (incr) { P->base_valid = true; P->base_kstride = kstride; }
        }
        const int nt = (p + 15) / 16;
        if (nt < 1 || nt > 8) return fail(CNMFE_EUNSUPPORTED, "fit_ring_model: %d ring neighbours (<= %d supported)", p, PMAX_RING);
        DevBuf &dFill = ctx->solve_fill;                     // the fill values {0, 1} of missing neighbours, in global memory (ring_solve.hpp)
        const double fillv[2] = {0.0, 1.0};
        RET(to_dev(ctx, dFill, fillv, 2));
        CovTab tab; tab.cov = ctx->cov.as<double>(); tab.pair_of = dPairOf.as<int>(); tab.nbr = g.nbr; tab.nbc = g.nbc; tab.maxd = maxd; tab.nrel = nrel;
        DevBuf &dWcodes = ctx->wcodes;
        const int woff = (g.p_radius + 15) >> 4;            // window origins reach -ceil(p_radius / 16) blocks (1 for rings up to 16 pixels, 2 up to 32)
        const int nwin = (g.nbr + 2 * woff) * (g.nbc + 2 * woff);
        // ---- packed systems (ring_solve_packed.hpp): the video's table re-laid once per pixel in the solve's register-tile order; the fits then apply the
        // footprints' corrections in registers and neither sweep the table (k_cov_correct) nor gather from it.  Conditions: the incremental table,
        // the memory (43 KB per pixel at p = 96), one launch (no split solve)
        bool packed = incr && ctx->opt("solve_packed", 1) != 0 && (int64_t)nblk * std::max(1, K) < (int64_t)1 << 31 &&
                      (int64_t)lst_k.size() * BLKPX < (int64_t)1 << 31;
        const size_t sys_bytes = (size_t)P->d * ((size_t)(nt * (nt + 1) / 2) * 256 + 16 * nt) * sizeof(double);
        if (packed && !(P->sys_valid && P->sys.cap >= sys_bytes)) {
            if (P->sys.cap < sys_bytes) {
                size_t fr = 0, tot = 0;
                CK(hipMemGetInfo(&fr, &tot));
                if (fr < sys_bytes + ((size_t)8 << 30)) packed = false;          // (not worth the last gigabytes: the table path needs none of this)
            }
            if (packed) {
                RET(P->sys.ensure(sys_bytes));
                RET(dWcodes.ensure((size_t)nwin * 256 * sizeof(int)));
                LAUNCH(ctx, "bg_win_codes", k_win_codes, dim3((unsigned)nwin), dim3(256), 0, dPairOf.as<int>(), g.nbr, g.nbc, woff, maxd, nrel, dWcodes.as<int>());
                CovTab tb = tab; tb.cov = P->cov_base.as<double>(); tb.wcodes = dWcodes.as<int>(); tb.woff = woff;
                int *dErrP = nullptr;
                RET(ctx_errflag(ctx, &dErrP));
#define RSP_CASE(NT_) case NT_: LAUNCH(ctx, "bg_sys_pack", (k_sys_pack<NT_>), dim3((unsigned)P->d), dim3(64), 0, tb, g, P->ring_dr.as<int>(), P->ring_dc.as<int>(), \
                                        P->rowsum_base.as<double>(), (const unsigned char *)nullptr, (float *)nullptr, dErrP, 0, (const int *)nullptr, dFill.as<double>(), P->sys.as<double>()); break;
                switch (nt) { RSP_CASE(1) RSP_CASE(2) RSP_CASE(3) RSP_CASE(4) RSP_CASE(5) RSP_CASE(6) RSP_CASE(7) RSP_CASE(8) default: break; }
#undef RSP_CASE
                P->sys_valid = true; P->kinv_valid = false; P->kinv_lam_valid = false; P->kinv_fits = 0;
            }
        }
        if (incr) {
            // ---- B2a': U~ on the blocks near footprints, then one sweep base -> cov (table path) or nothing (packed path: the solve corrects in registers) ----
            RET(ctx->rowsum.ensure((size_t)nblk * BLKPX * sizeof(double)));
            if (has_a) {
                if (!proj_queued) RET(queue_projection());             // (first fit of the patch: behind the video's table)
                RET(to_dev(ctx, dSlot, slot_of.data(), slot_of.size()));
                if (keep_pt) RET(P->pt_tab.ensure(std::max<size_t>(1, lst_k.size()) * BLKPX * sizeof(double)));
                LAUNCH(ctx, "bg_win_fix", k_win_fix, dim3((unsigned)blall.size(), WIN_NLB / WF_S), dim3(256), 0, g, (int)K, dArow.as<int>(), dAcol.as<int>(), dAval.as<float>(), dLp.as<int>(),
                       dSlot.as<short>(), dBl.as<int>(), nsg, dUt.as<double>(), ut_stride, dGb.as<double>(), gb_stride, keep_pt ? P->pt_tab.as<double>() : nullptr,
                       win_i8 ? ctx->gk.as<double>() : (const double *)nullptr, dLk.as<int>());
                if (keep_pt) {
                    P->pt_K = K; P->pt_lp_h = lst_ptr; P->pt_slot_h = slot_of;
                    P->pt_gen = bound_rows_of(ctx, C, c_order, K, P->pt_rows) ? ctx->bound_gen : -1;
                    P->pt_valid = true;
                }
                const int csplit = npairs >= 8192 ? 1 : npairs >= 4096 ? 2 : 4;      // (small patches: a pair's sweep is a long serial loop, one workgroup per pair leaves the chip idle)
                if (!packed)
                LAUNCH(ctx, "bg_cov_correct", k_cov_correct, dim3((unsigned)npairs, (unsigned)csplit), dim3(256), 0, P->cov_base.as<double>(), ctx->cov.as<double>(), dPairs.as<int4>(),
                       dNeed.as<unsigned short>(), g, (int)K, dArow.as<int>(), dAcol.as<int>(), dAval.as<float>(), dLp.as<int>(), dSlot.as<short>(), dUt.as<double>());
                LAUNCH(ctx, "bg_rowsum_correct", k_rowsum_correct, dim3(nblk), dim3(256), 0, P->rowsum_base.as<double>(), ctx->rowsum.as<double>(), g,
                       dArow.as<int>(), dAcol.as<int>(), dAval.as<float>(), dCsum.as<double>());
                // ---- round 6 (ring_solve_staged.hpp): U~ and A of every neuron as dense images over its footprint's bounding box dilated by the ring radius,
                // and per list position the neuron's window: the solve samples them with index arithmetic instead of walking CSR rows and slot tables ----
                if (packed && ctx->opt("solve_staged", 1) != 0 && P->nr_b < 32768 && P->nc_b < 32768) {
                    std::vector<int> nmeta((size_t)8 * K, 0), lmeta((size_t)8 * std::max<size_t>(1, lst_k.size()), 0);
                    int64_t tot = 0;
                    stage_ok = true;
                    for (int k = 0; k < K; ++k) {
                        int *mk = &nmeta[(size_t)8 * k];
                        if (nbox[(size_t)4 * k + 1] < 0) continue;                   // (an empty footprint: on no list)
                        // a centre whose ring reaches the footprint lies within one radius of its bounding box (the candidate test), and that ring's pixels within
                        // two: U~ = Yc Cc' - ... is non-zero wherever the video is, and the correction pairs it with A on ANOTHER ring pixel
                        const int R2 = 2 * g.p_radius;
                        const int r0 = std::max(0, nbox[(size_t)4 * k] - R2), r1 = std::min(P->nr_b - 1, nbox[(size_t)4 * k + 1] + R2);
                        const int c0 = std::max(0, nbox[(size_t)4 * k + 2] - R2), c1 = std::min(P->nc_b - 1, nbox[(size_t)4 * k + 3] + R2);
                        mk[5] = std::max(0, nbox[(size_t)4 * k] - g.p_radius) | (std::min(P->nr_b - 1, nbox[(size_t)4 * k + 1] + g.p_radius) << 16);
                        mk[6] = std::max(0, nbox[(size_t)4 * k + 2] - g.p_radius) | (std::min(P->nc_b - 1, nbox[(size_t)4 * k + 3] + g.p_radius) << 16);
                        const int64_t n = (int64_t)(r1 - r0 + 1) * (c1 - c0 + 1);
                        if (n > NWIN_MAX || tot + n > ((int64_t)1 << 30)) { stage_ok = false; break; }
                        mk[0] = r0; mk[1] = c0; mk[2] = r1 - r0 + 1; mk[3] = c1 - c0 + 1; mk[4] = (int)tot;
                        tot += n;
                    }
                    if (stage_ok) {
                        for (size_t i = 0; i < lst_k.size(); ++i) {
                            const int *mk = &nmeta[(size_t)8 * lst_k[i]];
                            int *ml = &lmeta[8 * i];
                            ml[0] = lst_k[i]; ml[1] = mk[0]; ml[2] = mk[1]; ml[3] = mk[2]; ml[4] = mk[3]; ml[5] = mk[4]; ml[6] = mk[5]; ml[7] = mk[6];
                        }
                        RET(to_dev(ctx, ctx->stg[0], nmeta.data(), nmeta.size()));
                        RET(to_dev(ctx, ctx->stg[1], lmeta.data(), lmeta.size()));
                        RET(ctx->stg[2].ensure((size_t)std::max<int64_t>(1, tot) * sizeof(double)));
                        RET(ctx->stg[3].ensure((size_t)std::max<int64_t>(1, tot) * sizeof(float)));
                        int *dErrW = nullptr;
                        RET(ctx_errflag(ctx, &dErrW));
                        LAUNCH(ctx, "bg_neuron_windows", k_nwin_build, dim3((unsigned)K), dim3(256), 0, ctx->stg[0].as<int>(), g, (int)K, dArow.as<int>(), dAcol.as<int>(), dAval.as<float>(),
                               dLp.as<int>(), dSlot.as<short>(), dUt.as<double>(), ctx->stg[2].as<double>(), ctx->stg[3].as<float>(), dErrW);
                    }
                }
            } else {                                                   // no footprints: Bf is the centred video itself
                if (!packed)
                CK(hipMemcpyAsync(ctx->cov.p, P->cov_base.p, (size_t)npairs * BLKPX * BLKPX * sizeof(double), hipMemcpyDeviceToDevice, ctx->st()));
                CK(hipMemcpyAsync(ctx->rowsum.p, P->rowsum_base.p, (size_t)nblk * BLKPX * sizeof(double), hipMemcpyDeviceToDevice, ctx->st()));
            }
        }
        ht.mark("base / correction launches");
        // ---- B2b ----
        const unsigned char *act = first_run ? nullptr : dActive.as<unsigned char>();
        const int probe = (int)ctx->opt("solve_probe", 0);
        if (packed) {
            PackArgs pa{};
            if (has_a) {
                pa.arow = dArow.as<int>(); pa.acol = dAcol.as<int>(); pa.aval = dAval.as<float>(); pa.lst_ptr = dLp.as<int>(); pa.lst_k = dLk.as<int>();
                pa.slot_of = dSlot.as<short>(); pa.K = (int)K; pa.Ut = dUt.as<double>();
            }
            int *dErrS = nullptr;
            RET(ctx_errflag(ctx, &dErrS));
            // ---- round 6 (ring_solve_inv.hpp): the fit out of the cached explicit inverses.  solve_inv: 0 off; 1 (default) the inverses are built in front of a
            // patch's SECOND fit with footprints (a recording fitted once never pays the build) at the ridge its first fit left; 2 in front of the first ----
            const int inv_mode = (int)ctx->opt("solve_inv", 0);             // (off by default: in the steady-state iteration the ridge drifts every fit, the series takes 1-2 terms per pixel and the fit is no faster than the factorising kernel -- DESIGN.md)
            const size_t kinv_bytes = (size_t)P->d * (size_t)ri_stride(nt) * sizeof(double);
            bool inv_ok = inv_mode != 0 && has_a && nt <= 6 && !(probe & 7);
            if (inv_ok && P->kinv.cap < kinv_bytes) {
                size_t fr = 0, tot = 0;
                CK(hipMemGetInfo(&fr, &tot));
                if (fr < kinv_bytes + ((size_t)8 << 30)) inv_ok = false;
            }
            double *lam_arr = nullptr;
            if (inv_ok) {
                if (P->kinv_lam.cap < (size_t)P->d * sizeof(double)) { RET(P->kinv_lam.ensure((size_t)P->d * sizeof(double))); P->kinv_lam_valid = false; }
                RET(P->kinv_list.ensure(((size_t)2 * P->d + 4) * sizeof(int)));
                if (!P->kinv_lam_valid) { CK(hipMemsetAsync(P->kinv_lam.p, 0, (size_t)P->d * sizeof(double), ctx->st())); P->kinv_lam_valid = true; }
                lam_arr = P->kinv_lam.as<double>();
                if (!P->kinv_valid && (P->kinv_fits >= 1 || inv_mode >= 2)) {
                    RET(P->kinv.ensure(kinv_bytes));
#define RI_CASE(NT_) case NT_: LAUNCH(ctx, "bg_ring_inverse", (k_ring_inverse<NT_>), dim3((unsigned)P->d), dim3(64), 0, P->sys.as<double>(), g, P->ring_dr.as<int>(), \
                                        P->ring_dc.as<int>(), P->rowsum_base.as<double>(), (const double *)lam_arr, P->kinv.as<double>()); break;
                    switch (nt) { RI_CASE(1) RI_CASE(2) RI_CASE(3) RI_CASE(4) RI_CASE(5) RI_CASE(6) default: break; }
#undef RI_CASE
                    P->kinv_valid = true;
                }
                ++P->kinv_fits;
            }
            if (inv_ok && P->kinv_valid) {
                int *flist = P->kinv_list.as<int>(), *rlist = flist + P->d, *fcnt = flist + 2 * P->d;
                CK(hipMemsetAsync(fcnt, 0, 4 * sizeof(int), ctx->st()));
                InvArgs ia{};
                ia.kp = P->kinv.as<double>(); ia.sys = P->sys.as<double>(); ia.csum = dCsum.as<double>(); ia.lam_out = lam_arr;
                ia.flist = flist; ia.rlist = rlist; ia.fcnt = fcnt; ia.maxit = (int)ctx->opt("solve_inv_terms", 5);
                const unsigned nlist = (unsigned)std::min<int64_t>(P->d, 2048);
#define RA_CASE(NT_) case NT_: \
                LAUNCH(ctx, "bg_ring_solve", (k_ring_apply<NT_>), dim3((unsigned)P->d), dim3(64), 0, ia, pa, g, P->ring_dr.as<int>(), P->ring_dc.as<int>(), ctx->rowsum.as<double>(), \
                       act, P->W.as<float>(), dErrS, probe, (const int *)nullptr); \
                LAUNCH(ctx, "bg_ring_solve_rest", (k_ring_solve6<NT_>), dim3((unsigned)P->d), dim3(64), 0, P->sys.as<double>(), pa, g, P->ring_dr.as<int>(), P->ring_dc.as<int>(), \
                       ctx->rowsum.as<double>(), (const unsigned char *)nullptr, P->W.as<float>(), dErrS, probe, (const int *)flist, (const int *)fcnt, lam_arr); \
                LAUNCH(ctx, "bg_ring_inverse_rest", (k_ring_inverse_list<NT_>), dim3(nlist), dim3(64), 0, P->sys.as<double>(), g, P->ring_dr.as<int>(), P->ring_dc.as<int>(), \
                       P->rowsum_base.as<double>(), (const double *)lam_arr, P->kinv.as<double>(), (const int *)rlist, (const int *)(fcnt + 3)); break;
                switch (nt) { RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) default: break; }
#undef RA_CASE
            } else if (has_a && stage_ok && nt <= 6) {
                StageArgs sg{dLp.as<int>(), ctx->stg[1].as<int>(), ctx->stg[2].as<double>(), ctx->stg[3].as<float>()};
#define RS8_CASE(NT_) case NT_: LAUNCH(ctx, "bg_ring_solve", (k_ring_solve8<NT_>), dim3((unsigned)P->d), dim3(64), 0, P->sys.as<double>(), sg, g, P->ring_dr.as<int>(), \
                                        P->ring_dc.as<int>(), ctx->rowsum.as<double>(), act, P->W.as<float>(), dErrS, probe, (const int *)nullptr, lam_arr); break;
                switch (nt) { RS8_CASE(1) RS8_CASE(2) RS8_CASE(3) RS8_CASE(4) RS8_CASE(5) RS8_CASE(6) default: break; }
#undef RS8_CASE
            } else {
#define RS6_CASE(NT_) case NT_: LAUNCH(ctx, "bg_ring_solve", (k_ring_solve6<NT_>), dim3((unsigned)P->d), dim3(64), 0, P->sys.as<double>(), pa, g, P->ring_dr.as<int>(), \
                                        P->ring_dc.as<int>(), ctx->rowsum.as<double>(), act, P->W.as<float>(), dErrS, probe, (const int *)nullptr, (const int *)nullptr, lam_arr); break;
            switch (nt) { RS6_CASE(1) RS6_CASE(2) RS6_CASE(3) RS6_CASE(4) RS6_CASE(5) RS6_CASE(6) RS6_CASE(7) RS6_CASE(8) default: break; }
#undef RS6_CASE
            }
        } else {
        RET(dWcodes.ensure((size_t)nwin * 256 * sizeof(int)));
        LAUNCH(ctx, "bg_win_codes", k_win_codes, dim3((unsigned)nwin), dim3(256), 0, dPairOf.as<int>(), g.nbr, g.nbc, woff, maxd, nrel, dWcodes.as<int>());
        tab.wcodes = dWcodes.as<int>(); tab.woff = woff;
        RET(solve_launch(ctx, P, tab, g, ctx->rowsum.as<double>(), act, nt, probe, dFill.as<double>()));
        }
    }
    RET(ring_stats_enqueue(ctx, P));                         // what the NEXT fit of this patch asks of the W being written now
    ht.mark("solve launch");
    if (b0_out) {
        std::vector<double> tmp(P->d);
        CK(hipMemcpyAsync(tmp.data(), P->b0.p, P->d * sizeof(double), hipMemcpyDeviceToHost, ctx->st()));
        CK(hipStreamSynchronize(ctx->st()));
        for (int64_t i = 0; i < P->d; ++i) b0_out[i] = (float)tmp[i];
        if (nactive < 0) nactive = *reinterpret_cast<const int *>((const char *)P->stat_host + 8);
    }
    // without b0_out the call returns with the fit in flight (every later engine call is stream-ordered behind it)
    if (info) { info[0] = first_run ? 1 : 0; info[1] = kstride; info[2] = nactive; info[3] = pmax; }
    P->ysig_valid = false;
    return 0;
}

// the window projection of ALL frames on the int8 pipe out of a patch's resident digit planes, for callers outside this file (vproj.hip: the spatial update's table):
// digits of the centred traces dCc (K rows, ldc apart), the (block, trace-group pair) items of the lists lst_ptr / blall, one launch; Ut[seg * ut_stride + ...] as
// k_win_proj_i8 leaves it (per-segment partial sums in real units)
int win_i8_table(cnmfe_ctx *ctx, Patch *P, const char *name_dig, const char *name_proj, int K, const float *dCc, int64_t ldc, const std::vector<int> &lst_ptr, const std::vector<int> &blall,
                 const int *dLp, const int *dLk, int nsg, double *dUt, int64_t ut_stride) {
    const int64_t T16 = P->dig_T16;
    RET(ctx->tdig.ensure((size_t)K * T16 * 4 * sizeof(uint4)));
    RET(ctx->tscale.ensure((size_t)K * sizeof(double)));
    LAUNCH(ctx, name_dig, k_trace_dig, dim3(K), dim3(256), 0, dCc, ldc, (int64_t)P->T, T16, ctx->tdig.as<uint4>(), ctx->tscale.as<double>());
    std::vector<int> items;
    for (int b_ : blall) {
        const int ntl = (lst_ptr[b_ + 1] - lst_ptr[b_] + 15) >> 4;
        for (int gq = 0; gq < (ntl + 1) / 2; ++gq) items.push_back(b_ | (gq << 24));
    }
    RET(to_dev(ctx, ctx->win_items, items.data(), items.size()));
    if (!items.empty())
        if (win_planes3(ctx, T16)) { LAUNCH(ctx, name_proj, k_win_proj_i8<1>, dim3((unsigned)(items.size() * nsg)), dim3(512), 0, P->dig.as<uint4>(), T16, P->dig_sc.as<double>(), ctx->tdig.as<uint4>(),
               ctx->tscale.as<double>(), dLp, dLk, ctx->win_items.as<int>(), nsg, dUt, ut_stride); } else { LAUNCH(ctx, name_proj, k_win_proj_i8<0>, dim3((unsigned)(items.size() * nsg)), dim3(512), 0, P->dig.as<uint4>(), T16, P->dig_sc.as<double>(), ctx->tdig.as<uint4>(),
               ctx->tscale.as<double>(), dLp, dLk, ctx->win_items.as<int>(), nsg, dUt, ut_stride); }
    return 0;
}

// loads this translation unit's code object now (HIP loads it at the first launch of one of its kernels -- milliseconds each that would otherwise fall into the first iteration): cnmfe_create
int tu_warm_bg() { hipFuncAttributes at; return hipFuncGetAttributes(&at, (const void *)k_rowsum) == hipSuccess ? 0 : -1; }

}  // namespace cnmfe
