// R1: the residual / background-subtraction kernel.
//   Ysig = Y(patch,:) - b0 - W * (R - mean_t R),   R = Y_block - A_prev * C_prev
// (update_spatial_parallel.m:162-166 == update_temporal_parallel.m:149-152, rearranged so that
//  only centred quantities are accumulated in fp32).
//
// HBM-bound by design: one read of the block video + one write of Ysig.  A workgroup owns a
// TR x TC tile of centre pixels for a whole segment of frames.  The 96..128 ring weights of a
// thread's pixel live in VGPRs for the lifetime of the workgroup; per chunk of 4 frames the
// (TR+2r) x (TC+2r) halo of the centred residual is staged once in LDS as float4-per-pixel
// and every ring neighbour is one conflict-free ds_read_b128 + 4 FMAs.
#include "common.hpp"

namespace cnmfe {

// row means (double) and centred copy of a [k][ldc] trace matrix; with `idx`, row k of the result is row idx[k] of C (the selected rows of the bound
// matrix are centred where they lie: no gathered copy in between)
__global__ void __launch_bounds__(256) k_center_traces(const float *__restrict__ C, int64_t ldc, int64_t T, float *__restrict__ Cc, double *__restrict__ Cmean, const int *__restrict__ idx) {
    int k = blockIdx.x;
    const float4 *row = reinterpret_cast<const float4 *>(C + (int64_t)(idx ? idx[k] : k) * ldc);      // ldc is a multiple of 4, the base 256-byte aligned
    __shared__ double red[4];
    const int64_t n4 = ldc >> 2;
    // 16-byte loads, four partial sums per thread: a row is 40 KB and one workgroup's, so the kernel lasts as long as one thread's chain of loads
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t c = threadIdx.x; c < n4; c += 256) {
        const float4 v = row[c];
        const int64_t t = 4 * c;
        s0 += v.x; s1 += t + 1 < T ? v.y : 0.f; s2 += t + 2 < T ? v.z : 0.f; s3 += t + 3 < T ? v.w : 0.f;     // (t < T for every c: ldc - T < 4)
    }
    double s = (s0 + s1) + (s2 + s3);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const double mean = ((red[0] + red[1]) + (red[2] + red[3])) / (double)T;
    if (threadIdx.x == 0) Cmean[k] = mean;
    float4 *out = reinterpret_cast<float4 *>(Cc + (int64_t)k * ldc);
    for (int64_t c = threadIdx.x; c < n4; c += 256) {
        const float4 v = row[c];
        const int64_t t = 4 * c;
        float4 o;
        o.x = (float)((double)v.x - mean);
        o.y = t + 1 < T ? (float)((double)v.y - mean) : 0.f;
        o.z = t + 2 < T ? (float)((double)v.z - mean) : 0.f;
        o.w = t + 3 < T ? (float)((double)v.w - mean) : 0.f;
        out[c] = o;
    }
}

int center_traces(cnmfe_ctx *ctx, const float *C, int64_t ldc, int32_t K, int64_t T, DevBuf &Cc, DevBuf &Cmean) {
    RET(Cc.ensure_hw(std::max<int64_t>(1, (int64_t)K * ldc) * sizeof(float), ctx->hw_cc));
    RET(Cmean.ensure_hw(std::max<int32_t>(1, K) * sizeof(double), ctx->hw_cm));
    if (K > 0) LAUNCH(ctx, "center_traces", k_center_traces, dim3(K), dim3(256), 0, C, ldc, T, Cc.as<float>(), Cmean.as<double>(), (const int *)nullptr);
    return 0;
}

// upload_traces + center_traces for a caller that only wants the centred copy: rows of the bound matrix (CNMFE_BOUND / CNMFE_BOUND_ROWS) are centred
// straight out of it -- one launch, no K x T staging copy; a host matrix goes through `stage` as before
int upload_centered(cnmfe_ctx *ctx, DevBuf &stage, const float *C, int32_t K, int64_t T, int order, DevBuf &Cc, DevBuf &Cmean, int64_t *ldc_out) {
    const int64_t ldc = (T + 3) & ~int64_t(3);
    if (K > 0 && (order == CNMFE_BOUND || order == CNMFE_BOUND_ROWS)) {
        *ldc_out = ldc;
        if (!ctx->bound_valid || ctx->bound_T != T || (order == CNMFE_BOUND && ctx->bound_K != K))
            return fail(CNMFE_ESTATE, "no bound trace matrix of %d x %lld (cnmfe_traces_bind)", K, (long long)T);
        const int *didx = nullptr;
        if (order == CNMFE_BOUND_ROWS) {
            if (!C) return fail(CNMFE_EINVAL, "null trace matrix");
            const int32_t *rows = reinterpret_cast<const int32_t *>(C);
            for (int32_t k = 0; k < K; ++k)
                if (rows[k] < 0 || rows[k] >= ctx->bound_K) return fail(CNMFE_EINVAL, "row %d of the bound trace matrix (%d rows) does not exist", rows[k], ctx->bound_K);
            DevBuf &dIdx = ctx->tmp[15];
            RET(to_dev(ctx, dIdx, rows, (size_t)K));
            didx = dIdx.as<int>();
        }
        RET(Cc.ensure_hw(std::max<int64_t>(1, (int64_t)K * ldc) * sizeof(float), ctx->hw_cc));
        RET(Cmean.ensure_hw(std::max<int32_t>(1, K) * sizeof(double), ctx->hw_cm));
        LAUNCH(ctx, "center_traces", k_center_traces, dim3(K), dim3(256), 0, ctx->bound.as<float>(), ldc, T, Cc.as<float>(), Cmean.as<double>(), didx);
        return 0;
    }
    RET(upload_traces(ctx, stage, C, K, T, order, ldc_out));
    return center_traces(ctx, stage.as<float>(), *ldc_out, K, T, Cc, Cmean);
}

// Ysig4 [T/4][d] float4  ->  frame-major [T][d]
__global__ void k_ysig_unpack(const float4 *__restrict__ src, int64_t d, int64_t T, float *__restrict__ dst) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= d) return;
    const int64_t c = blockIdx.y, t = 4 * c;
    const float4 v = src[c * d + m];
    dst[t * d + m] = v.x;
    if (t + 1 < T) dst[(t + 1) * d + m] = v.y;
    if (t + 2 < T) dst[(t + 2) * d + m] = v.z;
    if (t + 3 < T) dst[(t + 3) * d + m] = v.w;
}

// dlt[m] = ymean_f[q(m)] - b0[m]   (double arithmetic, float result: it is small, = A*Cmean at m)
__global__ void k_dlt(const float *__restrict__ ymean_f, const double *__restrict__ b0, float *__restrict__ dlt,
                      int64_t d, int nr, int nr_b, int roff, int coff) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= d) return;
    int64_t q = (int64_t)((int)(m / nr) + coff) * nr_b + (int)(m % nr) + roff;
    dlt[m] = (float)((double)ymean_f[q] - b0[m]);
}

// W*A_prev per patch pixel (ELL rows): wa(m,k) = sum_i W(m,i) * A_prev(m + o_i, k), accumulated in ring order.
// One thread per pixel; its first WA_CAP (k, value) slots live in registers (4) and LDS ([slot][thread], conflict-free), slots beyond that -- a ring over more than
// 32 footprints: rare -- where the table itself lies (round 6: `cap` rows are allocated, max(32, min(K, WA_CAP_MAX)); the reference has no such limit,
// fit_ring_model.m:28 / update_spatial_parallel.m:162-166; only a ring over more than WA_CAP_MAX footprints still raises the error flag).
constexpr int WA_CAP_ = 32;
constexpr int WA_CAP_MAX = 256;
// STAGE: the (column, value) arrays of the block's CSR are copied to LDS first (dynamic, 8 B per entry) and the per-entry loop reads them there.  A pixel
// whose ring crosses footprints walks tens of entries one dependent load after the other; out of L2 that walk made the kernel's duration (70-130 us for
// a 128 x 128 patch, as much as for the 512 x 512 frame) -- what a small patch's footprints need fits the LDS many times over.
template <bool STAGE>
__global__ void __launch_bounds__(128) k_ring_wa(const float *__restrict__ W, int64_t d, int nr, int nr_b, int nc_b, int roff, int coff, int p,
                                                 const int *__restrict__ dr, const int *__restrict__ dc, const int *__restrict__ arow,
                                                 const int *__restrict__ acol_g, const float *__restrict__ aval_g, int nnz,
                                                 int *__restrict__ wa_cnt, int *__restrict__ wa_k, float *__restrict__ wa_v, int *__restrict__ overflow, int cap) {
    __shared__ int tk[WA_CAP_][128];
    __shared__ float tv[WA_CAP_][128];
    extern __shared__ __attribute__((aligned(16))) char wa_dyn[];
    const int *acol = acol_g; const float *aval = aval_g;
    if constexpr (STAGE) {
        int *sc = reinterpret_cast<int *>(wa_dyn); float *sv = reinterpret_cast<float *>(wa_dyn) + nnz;
        for (int e = threadIdx.x; e < nnz; e += 128) { sc[e] = acol_g[e]; sv[e] = aval_g[e]; }
        __syncthreads();
        acol = sc; aval = sv;
    }
    const int64_t m = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (m >= d) return;
    const int t = threadIdx.x;
    const int rbm = (int)(m % nr) + roff, cbm = (int)(m / nr) + coff;
    int n = 0;
    // slots 0..3 live in registers (a ring rarely meets more footprints), slots 4.. in LDS: the per-entry step was a serial walk of LDS reads -- find
    // the slot, read-modify-write its value -- and the wave runs it for every (offset, entry) ANY of its 64 pixels has, ~150 times in a row
    int rk0 = -1, rk1 = -1, rk2 = -1, rk3 = -1; float rv0 = 0.f, rv1 = 0.f, rv2 = 0.f, rv3 = 0.f;
    // eight ring offsets at a time: their weights and row extents are independent loads (the thread's serial depth was 96 x three dependent
    // loads, and most neighbours lie under no footprint at all); the accumulation itself stays in ring order
    for (int i0 = 0; i0 < p; i0 += 8) {
        float w8[8]; int e0[8], e1[8];
        // (every load unconditional, out-of-range neighbours clamped to pixel 0 and masked afterwards: a load behind a branch is a wait per offset)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u < p ? i0 + u : p - 1;
            const int rb = rbm + dr[i], cb = cbm + dc[i];
            const bool in = i0 + u < p && rb >= 0 && rb < nr_b && cb >= 0 && cb < nc_b;
            const int64_t q = in ? (int64_t)cb * nr_b + rb : 0;
            w8[u] = W[(int64_t)i * d + m];
            e0[u] = arow[q]; e1[u] = arow[q + 1];
            if (!in) w8[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (w8[u] == 0.f) e1[u] = e0[u];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            for (int e = e0[u]; e < e1[u]; ++e) {
                const int k = acol[e];
                const float a = aval[e], w = w8[u];
                const bool m0 = rk0 == k, m1 = rk1 == k, m2 = rk2 == k, m3 = rk3 == k;
                if (m0 | m1 | m2 | m3) {
                    rv0 = m0 ? fmaf(w, a, rv0) : rv0; rv1 = m1 ? fmaf(w, a, rv1) : rv1; rv2 = m2 ? fmaf(w, a, rv2) : rv2; rv3 = m3 ? fmaf(w, a, rv3) : rv3;
                } else if (n < 4) {
                    const float v = fmaf(w, a, 0.f);
                    if (n == 0) { rk0 = k; rv0 = v; } else if (n == 1) { rk1 = k; rv1 = v; } else if (n == 2) { rk2 = k; rv2 = v; } else { rk3 = k; rv3 = v; }
                    ++n;
                } else {
                    int s = 4;
                    while (s < n && s < WA_CAP_ && tk[s][t] != k) ++s;
                    if (s == WA_CAP_) while (s < n && wa_k[(int64_t)s * d + m] != k) ++s;          // (beyond the LDS slots: the table's own rows, this thread's column)
                    if (s == n) {
                        if (n == cap) { atomicOr(overflow, 2); continue; }         // (bit 1 of the context's error flag: reported at the next wait)
                        if (s < WA_CAP_) { tk[s][t] = k; tv[s][t] = 0.f; } else { wa_k[(int64_t)s * d + m] = k; wa_v[(int64_t)s * d + m] = 0.f; }
                        ++n;
                    }
                    if (s < WA_CAP_) tv[s][t] = fmaf(w, a, tv[s][t]);
                    else wa_v[(int64_t)s * d + m] = fmaf(w, a, wa_v[(int64_t)s * d + m]);
                }
            }
    }
    wa_cnt[m] = n;
    if (n > 0) { wa_k[m] = rk0; wa_v[m] = rv0; }
    if (n > 1) { wa_k[d + m] = rk1; wa_v[d + m] = rv1; }
    if (n > 2) { wa_k[2 * d + m] = rk2; wa_v[2 * d + m] = rv2; }
    if (n > 3) { wa_k[3 * d + m] = rk3; wa_v[3 * d + m] = rv3; }
    for (int s = 4; s < n && s < WA_CAP_; ++s) { wa_k[(int64_t)s * d + m] = tk[s][t]; wa_v[(int64_t)s * d + m] = tv[s][t]; }
}

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int WA_CAP = WA_CAP_;   // max neurons whose footprint a pixel's ring may touch

struct R1Args {
    const float4 *Y4; int64_t d_b; int nr_b, nc_b;      // resident video: centred, [T/4][d_b] float4
    int nr, nc, roff, coff; int64_t d;
    int64_t T, tseg;
    const float *W; int p, h;
    const int *offs;                 // generic path only: p entries (dc*HR + dr)
    const float *ymean_f; const float *dlt;
    // A_prev*C_prev enters through linearity: W*(A*Cc) = (W*A)*Cc.  wa_* is the per-pixel ELL table of W*A
    // ([slot][pixel], WA_CAP slots), wa_cnt[pixel] its length; Cc the centred traces [k][ldc].
    const int *wa_cnt; const int *wa_k; const float *wa_v; const float *Cc; int64_t ldc;
    float4 *Ysig4;                   // output, [T/4][d] float4
    int ntile_r;
    const int *tile_map;         // dispatch slot -> tile_r | tile_c << 16 (XCD-compact order, built on the host)
    int probe;                   // timing ablations of the arc-DMA kernel (option r1_probe): 1 no DMA after the first chunk, 2 no ring product,
                                 // 4 no partial-sum exchange, 8 no store.  Results are garbage with any bit set.
};

constexpr int R1_TKMAX = 64;                // traces staged per tile and chunk (one DMA instruction)
constexpr int R1_KBM = 8192;                // neurons the tile-list bitmap covers (beyond: per-pixel global loads)
constexpr int R1_OVF = 384;                 // workgroup-wide list of (W*A) entries beyond those a pixel keeps in registers

// compile-time ring of get_nhood(R) in MATLAB find() order (column offset slow, row offset fast)
template <int R> struct RingTab { int n; int dr[PMAX_RING]; int dc[PMAX_RING]; };
template <int R> constexpr RingTab<R> make_ring() {
    RingTab<R> t{};
    int n = 0;
    for (int c = -R; c <= R; ++c)
        for (int r = -R; r <= R; ++r) {
            int d2 = c * c + r * r;
            if (d2 >= R * R && d2 < (R + 1) * (R + 1)) { t.dr[n] = r; t.dc[n] = c; ++n; }
        }
    t.n = n;
    return t;
}
template <int R> struct RingConst { static constexpr RingTab<R> tab = make_ring<R>(); };

// stage the centred video halo of the 4-frame chunk c4: one 16-byte load per halo pixel
template <int NT>
__device__ __forceinline__ void stage_halo(const R1Args &a, float4 *halo, int tid, int HR, int NH, int hr0, int hc0, int64_t c4) {
#pragma unroll 2
    for (int idx = tid; idx < NH; idx += NT) {
        const int hr = idx % HR, hc = idx / HR;
        const int rb = hr0 + hr, cb = hc0 + hc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rb >= 0 && rb < a.nr_b && cb >= 0 && cb < a.nc_b) v = a.Y4[c4 * a.d_b + (int64_t)cb * a.nr_b + rb];
        halo[idx] = v;
    }
}

// (W*A)*Cc at pixel m for frames t0..t0+3
__device__ __forceinline__ float4 wa_term(const R1Args &a, int64_t m, int64_t t0) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n = a.wa_cnt[m];
    for (int e = 0; e < n; ++e) {
        const float v = a.wa_v[(int64_t)e * a.d + m];
        const float4 c4 = *reinterpret_cast<const float4 *>(a.Cc + (int64_t)a.wa_k[(int64_t)e * a.d + m] * a.ldc + t0);
        s.x = fmaf(v, c4.x, s.x); s.y = fmaf(v, c4.y, s.y); s.z = fmaf(v, c4.z, s.z); s.w = fmaf(v, c4.w, s.w);
    }
    return s;
}

// uniform base + 32-bit per-lane byte offset: lowers to `global_load v, v_off, s[base:base+1]`, so a
// frame-invariant address costs ONE VGPR instead of a hoisted 64-bit pointer pair per array.
__device__ __forceinline__ float ld_off(const float *base, uint32_t byteoff) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byteoff);
}
__device__ __forceinline__ float4 ld4_off(const float4 *base, uint32_t byteoff) {
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byteoff);
}
__device__ __forceinline__ void st4_off(float4 *base, uint32_t byteoff, float4 v) {
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + byteoff) = v;
}

// write-through store that DROPS the line from the XCD's L2 (sc1; MI355X_MICROARCH.md, stores of each flavour): Ysig is written once and read
// by later kernels -- kept in L2 it only evicts the halo lines neighbouring tiles are about to re-read (TCC hit rate of the sweep: 56 %).
// Inline asm: invisible to hipcc's vmcnt bookkeeping -- only for kernels that count vmcnt by hand (the LDS-DMA ones do).
typedef float f4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_off_wt(float4 *base, uint32_t byteoff, float4 v) {
    const f4v_t x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" :: "v"(byteoff), "v"(x), "s"(base) : "memory");
}

// ---- specialised kernel, LDS-DMA staging (r1_variant 10) -------------------------------------------------
// One centre pixel per thread, every neighbour address `thread base + immediate` (one ds_read_b128 and four FMAs per neighbour, no address
// arithmetic), the P weights in VGPRs for the workgroup's lifetime; the halo goes global -> LDS directly (global_load_lds_dwordx4): no staging
// VGPRs, no ds_write commit phase (49 KB per chunk through the 79 B/clk store path), and the halo of chunk c+2 is in
// flight while chunk c is consumed (three LDS buffers, ONE barrier per chunk).  The DMA writes lane-linearly, so a
// buffer is 48 wave-instructions x 64 slots of 16 B; slots past the halo and halo pixels outside the block fetch a
// clamped address: their ring weights are exactly 0 (every neighbour outside the block is outside the FOV).

template <int R, int TR, int TC, bool HAS_AC, int NBUF>
__global__ void __launch_bounds__(TR *TC, 2) k_residual_dma(R1Args a) {
    constexpr int NT = TR * TC, NW = NT / 64;
    constexpr int P = RingConst<R>::tab.n;
    constexpr int HR = TR + 2 * R, HC = TC + 2 * R, NH = HR * HC;
    constexpr int NIT = (NH + NT - 1) / NT;                 // DMA instructions per wave and chunk
    constexpr int NHp = NIT * NT;                           // slots per buffer; NBUF buffers: chunk c+NBUF-1 in flight under chunk c
    // the only LDS object: NBUF halo buffers | NBUF trace buffers of R1_TKMAX float4 | tile-list scratch (HAS_AC)
    extern __shared__ __attribute__((aligned(16))) float4 halo[];
    float4 *tbuf = halo + NBUF * NHp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tmap = a.tile_map[blockIdx.x];
    const int tile_r = tmap & 0xffff, tile_c = tmap >> 16;
    const int tr = tid % TR, tc = tid / TR;
    const int pr = tile_r * TR + tr, pc = tile_c * TC + tc;
    const bool valid = pr < a.nr && pc < a.nc;
    const int64_t m = valid ? (int64_t)pc * a.nr + pr : 0;
    const int hr0 = tile_r * TR + a.roff - R, hc0 = tile_c * TC + a.coff - R;
    const int hbase = tc * HR + tr;                     // biased base: neighbour (dr,dc) at halo[hbase + (dc+R)*HR + (dr+R)]
    // DMA plan: instruction j of wave w fills slots (j*NW + w)*64 + lane
    uint32_t qoff[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int idx = (j * NW + wave) * 64 + lane;
        const int hr = idx % HR, hc = idx / HR;
        int rb = hr0 + hr, cb = hc0 + hc;
        rb = rb < 0 ? 0 : (rb >= a.nr_b ? a.nr_b - 1 : rb);
        cb = cb < 0 ? 0 : (cb >= a.nc_b ? a.nc_b - 1 : cb);
        qoff[j] = idx < NH ? (uint32_t)(cb * a.nr_b + rb) * 16u : 0u;
    }
    const unsigned ldsA = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float4 *)halo;
    const unsigned lds0 = ldsA + (unsigned)wave * 1024u;
    const uint32_t mb = (uint32_t)m * 4u;
    // ---- (W*A_prev) rows: the tile's traces are staged in LDS once per chunk (wave 0, one DMA instruction), every
    // pixel reads its <= WA_PRE samples from there.  The tile list is the union of wa_k over the tile's pixels, built
    // here with a bitmap; entries it cannot hold (more than R1_TKMAX traces per tile, more than WA_PRE per pixel,
    // neuron index >= R1_KBM) fall back to per-pixel global loads in the epilogue.
    constexpr int WA_PRE = 8;                            // entries per pixel held in registers; the rest go through the overflow list
    int wsl[WA_PRE]; float wvv[WA_PRE]; int nwa = 0; bool spill = false;
    int ov0 = 0, ov1 = 0;                                // this pixel's range in the workgroup's overflow list (entries >= WA_PRE)
    uint32_t tkoff = 0;                                  // wave 0: byte offset of trace tk[lane] in Cc
    unsigned *bm = reinterpret_cast<unsigned *>(tbuf + NBUF * R1_TKMAX);      // R1_KBM/32 words
    int *pre = reinterpret_cast<int *>(bm + R1_KBM / 32);                    // exclusive popcount prefix, R1_KBM/32 + 1; then the list counter
    int *tk = pre + R1_KBM / 32 + 2;                                         // R1_TKMAX neuron ids
    int2 *ovl = reinterpret_cast<int2 *>(tk + R1_TKMAX);                     // R1_OVF (trace slot, weight bits)
    if (HAS_AC) {
        for (int w = tid; w < R1_KBM / 32; w += NT) bm[w] = 0u;
        if (tid < R1_TKMAX) tk[tid] = 0;
        if (tid == 0) pre[R1_KBM / 32 + 1] = 0;
        __syncthreads();
        nwa = valid ? a.wa_cnt[m] : 0;
        for (int e = 0; e < nwa; ++e) {                  // every trace this pixel needs goes on the tile list
            const int k = a.wa_k[(int64_t)e * a.d + m];
            if (k < R1_KBM) atomicOr(&bm[k >> 5], 1u << (k & 31));
        }
        __syncthreads();
        if (tid == 0) { int s_ = 0; for (int w = 0; w < R1_KBM / 32; ++w) { pre[w] = s_; s_ += __popc(bm[w]); } pre[R1_KBM / 32] = s_; }
        __syncthreads();
        for (int k = tid; k < R1_KBM; k += NT) {
            const unsigned word = bm[k >> 5];
            if ((word >> (k & 31)) & 1u) { const int pos = pre[k >> 5] + __popc(word & ((1u << (k & 31)) - 1u)); if (pos < R1_TKMAX) tk[pos] = k; }
        }
        auto slot_of = [&](int k) {
            if (k >= R1_KBM) return -1;
            const unsigned word = bm[k >> 5];
            const int sl = pre[k >> 5] + __popc(word & ((1u << (k & 31)) - 1u));
            return sl < R1_TKMAX ? sl : -1;
        };
#pragma unroll
        for (int e = 0; e < WA_PRE; ++e) {
            wsl[e] = -1; wvv[e] = 0.f;
            if (e < nwa) {
                const int sl = slot_of(a.wa_k[(int64_t)e * a.d + m]);
                if (sl >= 0) { wsl[e] = sl; wvv[e] = a.wa_v[(int64_t)e * a.d + m]; }
                else { wsl[e] = -2; spill = true; }           // not on the tile list: global fallback in the epilogue
            }
        }
        if (nwa > WA_PRE) {
            const int n = nwa - WA_PRE;
            ov0 = atomicAdd(&pre[R1_KBM / 32 + 1], n); ov1 = ov0 + n;
            bool ok = ov1 <= R1_OVF;
            for (int e = WA_PRE; e < nwa && ok; ++e) {
                const int sl = slot_of(a.wa_k[(int64_t)e * a.d + m]);
                if (sl < 0) ok = false;
                else ovl[ov0 + e - WA_PRE] = make_int2(sl, __float_as_int(a.wa_v[(int64_t)e * a.d + m]));
            }
            if (!ok) { ov0 = ov1 = 0; spill = true; }            // list full / trace not staged: this pixel's tail goes the global way
        }
        __syncthreads();
        tkoff = (uint32_t)tk[lane] * (uint32_t)(a.ldc * 4);
    }
    static_assert(P % 2 == 0, "ring size must be even (weights are held as pairs)");
    f2 wp[P / 2];
#pragma unroll
    for (int i = 0; i < P / 2; ++i) {                   // threads off the patch read pixel 0 and never store
        wp[i].x = ld_off(a.W + (int64_t)(2 * i) * a.d, mb);
        wp[i].y = ld_off(a.W + (int64_t)(2 * i + 1) * a.d, mb);
    }
    const float dl = ld_off(a.dlt, mb);
    const int64_t cbeg = ((int64_t)blockIdx.y * a.tseg) >> 2;
    const int64_t tend = (int64_t)blockIdx.y * a.tseg + a.tseg < a.T ? (int64_t)blockIdx.y * a.tseg + a.tseg : a.T;
    const int64_t cend = (tend + 3) >> 2;                   // chunks [cbeg, cend)
    const bool tw = HAS_AC && wave == 0;                    // the wave that also moves the traces
    auto issue = [&](int64_t c) {
        const int64_t cc = c < cend ? c : cend - 1;         // clamp: the vmcnt arithmetic stays uniform
        const float4 *y4 = a.Y4 + cc * a.d_b;
        const int b = (int)((c - cbeg) % NBUF);
        const unsigned dst = lds0 + (unsigned)b * (unsigned)(NHp * 16);
#pragma unroll
        for (int j = 0; j < NIT; ++j) glds16(y4, qoff[j], dst + (unsigned)(j * NW) * 1024u);
        if (tw) glds16(a.Cc + 4 * cc, tkoff, ldsA + (unsigned)(NBUF * NHp + b * R1_TKMAX) * 16u);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the loads above are the compiler's; start the hand count from zero
#pragma unroll
    for (int q = 0; q < NBUF - 1; ++q) issue(cbeg + q);
    for (int64_t c = cbeg; c < cend; ++c) {
        // outstanding after this wait: at most the loads of the NBUF-2 younger chunks (stores may retire out of order
        // with loads: counting only the younger LOADS is safe either way)
        if (tw) asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NIT + 1) * (NBUF - 2)) : "memory");
        else    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NIT * (NBUF - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(c + NBUF - 1);
        const int64_t t0 = c << 2;
        const int cb_ = (int)((c - cbeg) % NBUF);
        const float4 *hb = halo + cb_ * NHp + hbase;
        f2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
        constexpr int G = 4, NG = P / G, D = 3;
        static_assert(P % G == 0, "ring size must be a multiple of the read group");
        float4 r[D + 1][G];
#pragma unroll
        for (int g = 0; g < D; ++g)
#pragma unroll
            for (int j = 0; j < G; ++j)
                r[g][j] = hb[(RingConst<R>::tab.dc[g * G + j] + R) * HR + (RingConst<R>::tab.dr[g * G + j] + R)];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + D < NG) {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    r[(g + D) % (D + 1)][j] = hb[(RingConst<R>::tab.dc[(g + D) * G + j] + R) * HR + (RingConst<R>::tab.dr[(g + D) * G + j] + R)];
            }
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const float4 rv = r[g % (D + 1)][j];
                const f2 r01 = {rv.x, rv.y}, r23 = {rv.z, rv.w};
                const f2 wv = wp[(g * G + j) >> 1];
                if (((g * G + j) & 1) == 0) {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc01) : "v"(wv), "v"(r01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc23) : "v"(wv), "v"(r23));
                } else {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc01) : "v"(wv), "v"(r01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc23) : "v"(wv), "v"(r23));
                }
            }
            asm volatile("" : "+v"(acc01), "+v"(acc23) : : "memory");
        }
        const float4 acc = make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
        float4 cv = hb[R * HR + R];
        if (HAS_AC) {
            const float4 *tb = tbuf + cb_ * R1_TKMAX;
#pragma unroll
            for (int e = 0; e < WA_PRE; ++e) {
                const float4 c4 = tb[wsl[e] >= 0 ? wsl[e] : 0];
                cv.x = fmaf(wvv[e], c4.x, cv.x); cv.y = fmaf(wvv[e], c4.y, cv.y);
                cv.z = fmaf(wvv[e], c4.z, cv.z); cv.w = fmaf(wvv[e], c4.w, cv.w);
            }
            for (int q = ov0; q < ov1; ++q) {               // entries beyond WA_PRE: (slot, weight) pairs in LDS, no global access
                const int2 en = ovl[q];
                const float4 c4 = tb[en.x]; const float v = __int_as_float(en.y);
                cv.x = fmaf(v, c4.x, cv.x); cv.y = fmaf(v, c4.y, cv.y); cv.z = fmaf(v, c4.z, cv.z); cv.w = fmaf(v, c4.w, cv.w);
            }
        }
        if (valid) {
            if (HAS_AC && spill) {                          // rare: entries neither the tile list nor the overflow list could hold
                for (int e = 0; e < nwa; ++e) {
                    if (e < WA_PRE ? wsl[e] != -2 : ov1 > ov0) continue;
                    const float v = a.wa_v[(int64_t)e * a.d + m];
                    const float4 c4 = *reinterpret_cast<const float4 *>(a.Cc + (int64_t)a.wa_k[(int64_t)e * a.d + m] * a.ldc + t0);
                    cv.x = fmaf(v, c4.x, cv.x); cv.y = fmaf(v, c4.y, cv.y); cv.z = fmaf(v, c4.z, cv.z); cv.w = fmaf(v, c4.w, cv.w);
                }
            }
            st4_off(a.Ysig4 + c * a.d, mb * 4u, make_float4(cv.x + dl - acc.x, cv.y + dl - acc.y, cv.z + dl - acc.z, cv.w + dl - acc.w));
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the clamped tail loads before the LDS is released
}

template <int R, int TR, int TC>
static int launch_r1_dma(cnmfe_ctx *ctx, const R1Args &a, bool has_ac, dim3 grid) {
    constexpr int NT = TR * TC, NH = (TR + 2 * R) * (TC + 2 * R), NIT = (NH + NT - 1) / NT;
    constexpr int NBUF = (size_t)3 * (NIT * NT + R1_TKMAX) * sizeof(float4) + 4096 <= 160 * 1024 ? 3 : 2;
    const size_t shmem = (size_t)NBUF * NIT * NT * sizeof(float4) + (size_t)NBUF * R1_TKMAX * sizeof(float4) +
                         (size_t)(R1_KBM / 32 + R1_KBM / 32 + 2 + R1_TKMAX + 2 * R1_OVF) * sizeof(int);
    if (shmem > 160 * 1024) return fail(CNMFE_EUNSUPPORTED, "ring radius %d: LDS-DMA halo needs %zu B", R, shmem);
    CK(hipFuncSetAttribute((const void *)k_residual_dma<R, TR, TC, true, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    CK(hipFuncSetAttribute((const void *)k_residual_dma<R, TR, TC, false, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    if (has_ac) LAUNCH(ctx, "residual_r1", (k_residual_dma<R, TR, TC, true, NBUF>), grid, dim3(NT), shmem, a);
    else        LAUNCH(ctx, "residual_r1", (k_residual_dma<R, TR, TC, false, NBUF>), grid, dim3(NT), shmem, a);
    return 0;
}

// Dispatch order of the tiles.  Workgroup b of a launch runs on XCD b % 8 and each XCD starts its workgroups in
// order, 32 resident at a time (one per CU: the halo buffers fill the LDS).  The 32 tiles an XCD works on
// together stream the same frames at the same pace, so the halo pixels they share are fetched from the fabric
// once and served from that XCD's L2 afterwards: the fabric-side read amplification is the halo'd area of the
// *union* of those 32 tiles over its own area.  order 1 therefore hands every XCD compact, near-square groups
// of 32 tiles (super-tiles); order 0 is the plain column-major strip.
static int build_tile_map(cnmfe_ctx *ctx, DevBuf &buf, int ntr, int ntc, int TR, int TC, int h, int order, int per_xcd = 32) {
    const int n = ntr * ntc;
    std::vector<int32_t> logical; logical.reserve(n);
    if (order == 0) {
        for (int c = 0; c < ntc; ++c) for (int r = 0; r < ntr; ++r) logical.push_back(r | (c << 16));
    } else {
        int sr = per_xcd, sc = 1; double best = 1e300;          // per_xcd workgroups of an XCD are resident together (32 CUs x workgroups per CU)
        for (int a2 = 1; a2 <= per_xcd; a2 *= 2) {
            const int b2 = per_xcd / a2;
            const double area = (double)(std::min(a2, ntr) * TR + 2 * h) * (std::min(b2, ntc) * TC + 2 * h);
            if (area < best) { best = area; sr = a2; sc = b2; }
        }
        for (int C = 0; C < ntc; C += sc) for (int Rr = 0; Rr < ntr; Rr += sr)
            for (int c = C; c < std::min(C + sc, ntc); ++c) for (int r = Rr; r < std::min(Rr + sr, ntr); ++r)
                logical.push_back(r | (c << 16));
    }
    std::vector<int32_t> map(n);
    for (int b = 0; b < n; ++b) map[b] = logical[n % 8 == 0 ? (b % 8) * (n / 8) + b / 8 : b];
    return to_dev(ctx, buf, map.data(), map.size());
}

// ---- generic kernel: any ring (runtime offsets, weights streamed from L2) --------------------------
template <bool HAS_AC>
__global__ void __launch_bounds__(256) k_residual_gen(R1Args a) {
    constexpr int NT = 256, TR = 16;
    extern __shared__ __attribute__((aligned(16))) float4 halo[];
    const int h = a.h, HR = TR + 2 * h, HC = TR + 2 * h, NH = HR * HC;
    const int tid = threadIdx.x;
    const int tile_r = blockIdx.x % a.ntile_r, tile_c = blockIdx.x / a.ntile_r;
    const int tr = tid % TR, tc = tid / TR;
    const int pr = tile_r * TR + tr, pc = tile_c * TR + tc;
    const bool valid = pr < a.nr && pc < a.nc;
    const int64_t m = valid ? (int64_t)pc * a.nr + pr : 0;
    const int64_t qc = (int64_t)(pc + a.coff) * a.nr_b + (pr + a.roff);
    const int hr0 = tile_r * TR + a.roff - h, hc0 = tile_c * TR + a.coff - h;
    const int base = (tc + h) * HR + (tr + h);
    const float dl = valid ? a.dlt[m] : 0.f;
    (void)qc;
    const int64_t tbeg = (int64_t)blockIdx.y * a.tseg;
    const int64_t tend = tbeg + a.tseg < a.T ? tbeg + a.tseg : a.T;
    for (int64_t t0 = tbeg; t0 < tend; t0 += 4) {
        stage_halo<NT>(a, halo, tid, HR, NH, hr0, hc0, t0 >> 2);
        __syncthreads();
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            if (HAS_AC) { const float4 s4 = wa_term(a, m, t0); acc.x = -s4.x; acc.y = -s4.y; acc.z = -s4.z; acc.w = -s4.w; }
            for (int i = 0; i < a.p; ++i) {
                const float wi = a.W[(int64_t)i * a.d + m];
                const float4 r = halo[base + a.offs[i]];
                acc.x = fmaf(wi, r.x, acc.x); acc.y = fmaf(wi, r.y, acc.y);
                acc.z = fmaf(wi, r.z, acc.z); acc.w = fmaf(wi, r.w, acc.w);
            }
            const float4 c = halo[base];                 // (Y - Ymean) at the centre
            a.Ysig4[(t0 >> 2) * a.d + m] = make_float4(c.x + dl - acc.x, c.y + dl - acc.y, c.z + dl - acc.z, c.w + dl - acc.w);
        }
        __syncthreads();
    }
}

// ---- incremental residual -------------------------------------------------------------------------------------
// Ysig = Y - b0 - W (R - mean R) with R = Y - A C splits into a part that depends on the video, W and b0 only and the footprint term
// (W A)(C - mean C).  An iteration asks for two residuals under the same W, b0 (update_spatial_parallel.m:162-166 with the halo-only
// neurons, update_temporal_parallel.m:149-152 with all of them): the second one is the resident Ysig plus the DIFFERENCE of two
// footprint terms -- one streaming pass (read + write Ysig) instead of a second ring sweep.
// A wave owns 64 consecutive pixels of a column; their rings touch only a dozen footprints between them, so the wave keeps ONE list of
// traces (wave-uniform row offsets -> scalar loads, one per trace and chunk) and every lane its weight for each of them (0 if its ring
// misses the footprint).  Per-lane float4 gathers of the same traces cost 16 L1 cycles per entry and wave (measured: 11 ms against
// 4.4 ms for the bare stream); they remain as the fallback for a wave that meets more than DELTA_NL traces or a pixel with more than
// DELTA_NE entries.
constexpr int DELTA_NE = 12, DELTA_NL = 16;
struct DeltaPlan { int nl; bool over; int koff[DELTA_NL]; float w[DELTA_NL]; };
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}
// the wave's trace list in ascending k, and this lane's weights on it
__device__ __forceinline__ void delta_plan(int n, const int (&ke)[DELTA_NE], const float (&ve)[DELTA_NE], int64_t ldc, float sign, DeltaPlan &pl) {
    int last = -1;
    pl.nl = 0;
#pragma unroll
    for (int j = 0; j < DELTA_NL; ++j) {
        int kmin = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < DELTA_NE; ++e) if (e < n && ke[e] > last) kmin = min(kmin, ke[e]);
        const int km = wave_min(kmin);
        pl.koff[j] = 0; pl.w[j] = 0.f;
        if (km != 0x7fffffff) {
            float wv = 0.f;
#pragma unroll
            for (int e = 0; e < DELTA_NE; ++e) if (e < n && ke[e] == km) wv += ve[e];
            pl.w[j] = sign * wv; pl.koff[j] = (int)((int64_t)km * ldc); pl.nl = j + 1; last = km;
        }
    }
    int rest = n > DELTA_NE ? 0 : 0x7fffffff;                       // anything not on the list: a trace beyond DELTA_NL or an entry beyond DELTA_NE
#pragma unroll
    for (int e = 0; e < DELTA_NE; ++e) if (e < n && ke[e] > last) rest = 0;
    pl.over = wave_min(rest) == 0;
}
template <bool HAS2, bool HAS1>
__global__ void __launch_bounds__(256) k_residual_delta(float4 *__restrict__ ysig4, int64_t d, int64_t Tc, int64_t cseg,
                                                        const int *__restrict__ cnt2, const int *__restrict__ k2, const float *__restrict__ v2,
                                                        const float *__restrict__ Cc2, int64_t ldc2,
                                                        const int *__restrict__ cnt1, const int *__restrict__ k1, const float *__restrict__ v1,
                                                        const float *__restrict__ Cc1, int64_t ldc1, int probe) {
    typedef float nt4 __attribute__((ext_vector_type(4)));
    const int64_t m0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = m0 < d;
    const int64_t m = valid ? m0 : d - 1;
    const int n2 = HAS2 && valid ? cnt2[m] : 0, n1 = HAS1 && valid ? cnt1[m] : 0;
    int ke2[DELTA_NE], ke1[DELTA_NE];
    float ve2[DELTA_NE], ve1[DELTA_NE];
#pragma unroll
    for (int e = 0; e < DELTA_NE; ++e) {
        ke2[e] = 0; ve2[e] = 0.f; ke1[e] = 0; ve1[e] = 0.f;
        if (HAS2 && e < n2) { ke2[e] = k2[(int64_t)e * d + m]; ve2[e] = v2[(int64_t)e * d + m]; }
        if (HAS1 && e < n1) { ke1[e] = k1[(int64_t)e * d + m]; ve1[e] = v1[(int64_t)e * d + m]; }
    }
    DeltaPlan p2, p1;
    p2.nl = p1.nl = 0; p2.over = p1.over = false;
    if (HAS2) delta_plan(n2, ke2, ve2, ldc2, 1.f, p2);
    if (HAS1) delta_plan(n1, ke1, ve1, ldc1, -1.f, p1);
    const int64_t c0 = (int64_t)blockIdx.y * cseg, c1 = c0 + cseg < Tc ? c0 + cseg : Tc;
    if (!(p2.over || p1.over) && !(probe & 2)) {
        const int nl2 = probe & 1 ? 0 : p2.nl, nl1 = probe & 1 ? 0 : p1.nl;
        for (int64_t c = c0; c < c1; ++c) {
            nt4 *yp = reinterpret_cast<nt4 *>(ysig4 + c * d + m);
            const nt4 yv = __builtin_nontemporal_load(yp);
            float4 y = make_float4(yv.x, yv.y, yv.z, yv.w);
            // the two terms are summed separately and their DIFFERENCE is added: asking again for the term Ysig already contains changes nothing
            float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s2;
#pragma unroll
            for (int j = 0; j < DELTA_NL; ++j)
                if (HAS2 && j < nl2) {
                    const float4 t = *reinterpret_cast<const float4 *>(Cc2 + p2.koff[j] + 4 * c);         // wave-uniform address: a scalar load
                    s2.x = fmaf(p2.w[j], t.x, s2.x); s2.y = fmaf(p2.w[j], t.y, s2.y); s2.z = fmaf(p2.w[j], t.z, s2.z); s2.w = fmaf(p2.w[j], t.w, s2.w);
                }
#pragma unroll
            for (int j = 0; j < DELTA_NL; ++j)
                if (HAS1 && j < nl1) {
                    const float4 t = *reinterpret_cast<const float4 *>(Cc1 + p1.koff[j] + 4 * c);
                    s1.x = fmaf(-p1.w[j], t.x, s1.x); s1.y = fmaf(-p1.w[j], t.y, s1.y); s1.z = fmaf(-p1.w[j], t.z, s1.z); s1.w = fmaf(-p1.w[j], t.w, s1.w);
                }
            y.x += s2.x - s1.x; y.y += s2.y - s1.y; y.z += s2.z - s1.z; y.w += s2.w - s1.w;
            if (valid) __builtin_nontemporal_store((nt4){y.x, y.y, y.z, y.w}, yp);
        }
        return;
    }
    // fallback: per-lane gathers, entries beyond DELTA_NE re-read from the ELL rows
    int x2 = n2 < DELTA_NE ? n2 : DELTA_NE, x1 = n1 < DELTA_NE ? n1 : DELTA_NE;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { x2 = max(x2, __shfl_xor(x2, o)); x1 = max(x1, __shfl_xor(x1, o)); }
    const int nm2 = __builtin_amdgcn_readfirstlane(x2), nm1 = __builtin_amdgcn_readfirstlane(x1);
    for (int64_t c = c0; c < c1; ++c) {
        float4 y = ysig4[c * d + m];
        float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s2;
#pragma unroll
        for (int e = 0; e < DELTA_NE; ++e)
            if (HAS2 && e < nm2) {
                const float4 t = *reinterpret_cast<const float4 *>(Cc2 + (int64_t)ke2[e] * ldc2 + 4 * c);
                s2.x = fmaf(ve2[e], t.x, s2.x); s2.y = fmaf(ve2[e], t.y, s2.y); s2.z = fmaf(ve2[e], t.z, s2.z); s2.w = fmaf(ve2[e], t.w, s2.w);
            }
        for (int e = DELTA_NE; e < n2; ++e) {
            const float w = v2[(int64_t)e * d + m];
            const float4 t = *reinterpret_cast<const float4 *>(Cc2 + (int64_t)k2[(int64_t)e * d + m] * ldc2 + 4 * c);
            s2.x = fmaf(w, t.x, s2.x); s2.y = fmaf(w, t.y, s2.y); s2.z = fmaf(w, t.z, s2.z); s2.w = fmaf(w, t.w, s2.w);
        }
#pragma unroll
        for (int e = 0; e < DELTA_NE; ++e)
            if (HAS1 && e < nm1) {
                const float4 t = *reinterpret_cast<const float4 *>(Cc1 + (int64_t)ke1[e] * ldc1 + 4 * c);
                s1.x = fmaf(ve1[e], t.x, s1.x); s1.y = fmaf(ve1[e], t.y, s1.y); s1.z = fmaf(ve1[e], t.z, s1.z); s1.w = fmaf(ve1[e], t.w, s1.w);
            }
        for (int e = DELTA_NE; e < n1; ++e) {
            const float w = v1[(int64_t)e * d + m];
            const float4 t = *reinterpret_cast<const float4 *>(Cc1 + (int64_t)k1[(int64_t)e * d + m] * ldc1 + 4 * c);
            s1.x = fmaf(w, t.x, s1.x); s1.y = fmaf(w, t.y, s1.y); s1.z = fmaf(w, t.z, s1.z); s1.w = fmaf(w, t.w, s1.w);
        }
        y.x += s2.x - s1.x; y.y += s2.y - s1.y; y.z += s2.z - s1.z; y.w += s2.w - s1.w;
        if (valid) ysig4[c * d + m] = y;
    }
}

}  // namespace cnmfe
#include "resid_arc.hpp"
#include "resid_duo.hpp"
namespace cnmfe {

// r1_variant: 14 = duo roles (resid_duo.hpp; radius 15, no footprint term inside the sweep: the default), 11 = four arc roles on a 16 x 32 tile with the
// A_prev flavour (resid_arc.hpp; radius 15), 10 = one pixel per thread (radius 18 and the low-resolution radii of bg_ssub), -1 = generic kernel.
// Measured and removed in round 3 (numbers in profiles/r02/, profiles/r03/README.md): register-staged tiles (0-4), register-staged arc roles (5-9),
// quad roles (12), four roles with one barrier per chunk (13, and its unequal-arc variants).
template <int R>
static int launch_r1_v(cnmfe_ctx *ctx, int variant, const R1Args &a, bool has_ac, int ntile_c, int64_t nseg) {
    if constexpr (R == 15) {                              // arc kernels: radius 15 only (16-bit ds_read immediates, LDS size)
        if (variant == 11) return launch_r1_arc_dma<R>(ctx, a, has_ac, ntile_c, nseg);
        if (variant == 14) return launch_r1_duo<R>(ctx, a, ntile_c, nseg);
    }
    // note: a.ntile_r / grid depend on the tile shape, set by the caller through tile_shape()
    dim3 grid((unsigned)((int64_t)a.ntile_r * ntile_c), (unsigned)nseg);
    return launch_r1_dma<R, 32, 16>(ctx, a, has_ac, grid);
}

static void tile_shape(int variant, int &TR, int &TC) {
    TR = 32; TC = 16;                                     // variant 10
    if (variant == 11) { TR = ARC_TR; TC = ARC_TC; }
    else if (variant == 14) { TR = DUO_T; TC = DUO_T; }
}

// the ABI hands Ysig out frame-major (d x T column-major); resident it is 4-frame interleaved
int ysig_export(cnmfe_ctx *ctx, Patch *P, DevBuf &ysig, float *Ysig_out, int out_memspace) {
    const int64_t T = P->T;
    float *dstp = Ysig_out;
    if (out_memspace != CNMFE_DEVICE) { RET(ctx->stage.ensure((size_t)P->d * T * sizeof(float))); dstp = ctx->stage.as<float>(); }
    LAUNCH(ctx, "ysig_unpack", k_ysig_unpack, dim3((unsigned)((P->d + 255) / 256), (unsigned)P->Tc), dim3(256), 0,
           ysig.as<float4>(), P->d, T, dstp);
    if (out_memspace != CNMFE_DEVICE)
        CK(hipMemcpyAsync(Ysig_out, dstp, (size_t)P->d * T * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    RET(ctx_check_errflag(ctx));
    return 0;
}

// Ysig += (pending term) - (applied term); the pending term becomes the applied one
int residual_materialize(cnmfe_ctx *ctx, Patch *P) {
    RET(residual_realize(ctx, P));                           // a virtual residual: the sweep runs now (every caller is about to read Ysig itself)
    if (!P->pend) return 0;
    if (!P->ysig_valid || !P->ysig.p) return fail(CNMFE_ESTATE, "pending footprint term without a resident residual");
    if (P->pend_ac || P->res_ac) {
        const int64_t nblk = (P->d + 255) / 256;
        int64_t nseg = std::max<int64_t>(1, std::min<int64_t>(P->Tc, (8192 + nblk - 1) / nblk));
        const int64_t cseg = (P->Tc + nseg - 1) / nseg;
        nseg = (P->Tc + cseg - 1) / cseg;
        const dim3 grid((unsigned)nblk, (unsigned)nseg);
#define DELTA_ARGS P->ysig.as<float4>(), P->d, P->Tc, cseg, P->pendCnt.as<int>(), P->pendK.as<int>(), P->pendV.as<float>(), P->pendCc.as<float>(), P->pend_ldc, \
                   P->resCnt.as<int>(), P->resK.as<int>(), P->resV.as<float>(), P->resCc.as<float>(), P->res_ldc, (int)ctx->opt("r1_probe", 0)
        if (P->pend_ac && P->res_ac) LAUNCH(ctx, "residual_delta", (k_residual_delta<true, true>), grid, dim3(256), 0, DELTA_ARGS);
        else if (P->pend_ac)         LAUNCH(ctx, "residual_delta", (k_residual_delta<true, false>), grid, dim3(256), 0, DELTA_ARGS);
        else                         LAUNCH(ctx, "residual_delta", (k_residual_delta<false, true>), grid, dim3(256), 0, DELTA_ARGS);
#undef DELTA_ARGS
    }
    P->res_ac = P->pend_ac; P->res_ldc = P->pend_ldc; P->res_K = P->pend_K;
    if (P->pend_ac) { P->resCnt.swap(P->pendCnt); P->resK.swap(P->pendK); P->resV.swap(P->pendV); P->resCc.swap(P->pendCc); P->resCm.swap(P->pendCm); }
    P->pend = false;
    return 0;
}

// U(k,:) += sign * sum_l M(k,l) Cc_l,  M(k,l) = sum_m A(m,k) (W A_term)(m,l): the footprint term of the residual seen through A'.
// One workgroup per neuron k; deterministic (no floating-point atomics): the traces l that the rings of k's pixels touch are found with a
// bitmap, listed in ascending order, and each M(k,l) is summed over k's pixels in storage order by one thread.
constexpr int TERM_KMAX = 8192;
__global__ void __launch_bounds__(256) k_term_project(const int64_t *__restrict__ colptr, const int *__restrict__ erow, const float *__restrict__ aval, int64_t d,
                                                      const int *__restrict__ cnt, const int *__restrict__ wk, const float *__restrict__ wv,
                                                      const float *__restrict__ Cc, int64_t ldcc, int Kterm, float sign, float *__restrict__ U, int64_t ldu, int *__restrict__ overflow) {
    __shared__ unsigned bm[TERM_KMAX / 32];
    __shared__ int lst[512];
    __shared__ float Mv[512];
    __shared__ int nl_s;
    const int k = blockIdx.x, tid = threadIdx.x;
    const int64_t e0 = colptr[k], e1 = colptr[k + 1];
    if (e0 == e1) return;
    for (int w = tid; w < TERM_KMAX / 32; w += 256) bm[w] = 0u;
    __syncthreads();
    for (int64_t e = e0 + tid; e < e1; e += 256) {
        const int m = erow[e], n = cnt[m];
        for (int j = 0; j < n; ++j) { const int l = wk[(int64_t)j * d + m]; atomicOr(&bm[l >> 5], 1u << (l & 31)); }
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < (Kterm + 31) / 32; ++w) {
            unsigned x = bm[w];
            while (x) { if (n == 512) { *overflow = 1; break; } const int b = __ffs(x) - 1; x &= x - 1; lst[n++] = w * 32 + b; }
        }
        nl_s = n;
    }
    __syncthreads();
    const int nl = nl_s;
    // M(k, l): sixteen threads per list entry, each over a contiguous sixteenth of k's pixels in storage order, the sixteen partial sums added in
    // order by the first of them: a fixed association (bit-reproducible), 16 x shorter than one thread per entry
    __shared__ float part16[16][17];
    const int sub = tid & 15, slot = tid >> 4;
    const int64_t ne = e1 - e0, per = (ne + 15) / 16;
    const int64_t ea = e0 + sub * per, eb = ea + per < e1 ? ea + per : e1;
    for (int i0 = 0; i0 < nl; i0 += 16) {
        const int i = i0 + slot;
        float s = 0.f;
        if (i < nl) {
            const int l = lst[i];
            for (int64_t e = ea; e < eb; ++e) {
                const int m = erow[e], n = cnt[m];
                for (int j = 0; j < n; ++j) if (wk[(int64_t)j * d + m] == l) { s = fmaf(aval[e], wv[(int64_t)j * d + m], s); break; }
            }
        }
        part16[slot][sub] = s;
        __syncthreads();
        if (sub == 0 && i < nl) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += part16[slot][q];
            Mv[i] = sign * tot;
        }
        __syncthreads();
    }
    float4 *urow = reinterpret_cast<float4 *>(U + (int64_t)k * ldu);
    for (int64_t c = tid; c < (ldu >> 2); c += 256) {
        float4 a4 = urow[c];
        for (int i = 0; i < nl; ++i) {
            const float w = Mv[i];
            const float4 t = *reinterpret_cast<const float4 *>(Cc + (int64_t)lst[i] * ldcc + 4 * c);
            a4.x = fmaf(w, t.x, a4.x); a4.y = fmaf(w, t.y, a4.y); a4.z = fmaf(w, t.z, a4.z); a4.w = fmaf(w, t.w, a4.w);
        }
        urow[c] = a4;
    }
}

// A' Ysig for the residual the caller asked for: dU holds A' (resident Ysig); add the pending footprint term and take the applied one out
int residual_term_project(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *dColptr, const int *dErow, const float *dAval, float *dU, int64_t ldu, int *dOverflow) {
    if (!P->pend) return 0;
    // more than 512 traces near one footprint, or trace ids beyond the bitmap: fold the term into Ysig instead (the caller re-projects)
    if ((P->pend_ac && (P->pend_K > TERM_KMAX || P->pend_ldc != ldu)) || (P->res_ac && (P->res_K > TERM_KMAX || P->res_ldc != ldu))) return 1;
    if (P->pend_ac)
        LAUNCH(ctx, "temporal_term_project", k_term_project, dim3(K), dim3(256), 0, dColptr, dErow, dAval, P->d, P->pendCnt.as<int>(), P->pendK.as<int>(),
               P->pendV.as<float>(), P->pendCc.as<float>(), P->pend_ldc, (int)P->pend_K, 1.f, dU, ldu, dOverflow);
    if (P->res_ac)
        LAUNCH(ctx, "temporal_term_project", k_term_project, dim3(K), dim3(256), 0, dColptr, dErow, dAval, P->d, P->resCnt.as<int>(), P->resK.as<int>(),
               P->resV.as<float>(), P->resCc.as<float>(), P->res_ldc, (int)P->res_K, -1.f, dU, ldu, dOverflow);
    return 0;
}

// The same term seen through the SPATIAL projection U(m,k) = sum_t Ysig(m,t) Cc(k,t) on the entries of the search mask:
//   U(m,k) += sign * sum_j (W A_term)(m, l_j) <Cc_term(l_j,:), Cc(k,:)>
// -- a K_term x K matrix of inner products (k_cross_gram) and one pass over the mask's entries.
__global__ void __launch_bounds__(256) k_cross_gram(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb, int64_t T, int KB, float *__restrict__ G) {
    const int k = blockIdx.x, l = blockIdx.y;
    const float4 *a = reinterpret_cast<const float4 *>(A + (int64_t)l * lda), *b = reinterpret_cast<const float4 *>(B + (int64_t)k * ldb);
    double s = 0;
    for (int64_t c = threadIdx.x; 4 * c < T; c += 256) {     // (both matrices are zero beyond T up to their row stride)
        const float4 x = a[c], y = b[c];
        s += ((double)x.x * y.x + (double)x.y * y.y) + ((double)x.z * y.z + (double)x.w * y.w);
    }
    __shared__ double red[4];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) G[(int64_t)l * KB + k] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ void __launch_bounds__(256) k_term_fold_spatial(const int *__restrict__ erow, const int *__restrict__ ecol, int64_t nnz, int64_t d, const int *__restrict__ cnt,
                                                           const int *__restrict__ wk, const float *__restrict__ wv, const float *__restrict__ G, int KB, float sign,
                                                           float *__restrict__ U) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const int m = erow[e], n = cnt[m];
    if (n == 0) return;
    const int k = ecol[e];
    float s = 0.f;
    for (int j = 0; j < n; ++j) s = fmaf(wv[(int64_t)j * d + m], G[(int64_t)wk[(int64_t)j * d + m] * KB + k], s);
    U[e] += sign * s;
}
constexpr int64_t FOLD_MAX_PAIRS = int64_t(1) << 16;
// can the pending / applied terms of P be taken in through a spatial projection onto K traces with row stride ldc?
bool residual_term_foldable_spatial(const Patch *P, int32_t K, int64_t ldc) {
    if (!P->pend) return true;
    if (P->pend_ac && (P->pend_ldc != ldc || (int64_t)P->pend_K * K > FOLD_MAX_PAIRS)) return false;
    if (P->res_ac && (P->res_ldc != ldc || (int64_t)P->res_K * K > FOLD_MAX_PAIRS)) return false;
    return true;
}
int residual_term_fold_spatial(cnmfe_ctx *ctx, Patch *P, int32_t K, int64_t nnz, const int *dErow, const int *dEcol, const float *dCc, int64_t ldc, float *dU, DevBuf &dG) {
    if (!P->pend || nnz == 0) return 0;
    const int64_t T = P->T;
    auto one = [&](int Kt, const DevBuf &cnt, const DevBuf &wk, const DevBuf &wv, const DevBuf &cc, float sign) -> int {
        if (Kt <= 0) return 0;
        RET(dG.ensure((size_t)Kt * K * sizeof(float)));
        LAUNCH(ctx, "spatial_term_gram", k_cross_gram, dim3((unsigned)K, (unsigned)Kt), dim3(256), 0, cc.as<float>(), ldc, dCc, ldc, T, (int)K, dG.as<float>());
        LAUNCH(ctx, "spatial_term_fold", k_term_fold_spatial, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, dErow, dEcol, nnz, P->d, cnt.as<int>(), wk.as<int>(),
               wv.as<float>(), dG.as<float>(), (int)K, sign, dU);
        return 0;
    };
    if (P->pend_ac) RET(one(P->pend_K, P->pendCnt, P->pendK, P->pendV, P->pendCc, 1.f));
    if (P->res_ac)  RET(one(P->res_K, P->resCnt, P->resK, P->resV, P->resCc, -1.f));
    return 0;
}

// ---- compute_RSS (Sources2D.m:1358-1510), ring model, bg_ssub = 1 ------------------------------------------------------------
// E = Y(patch) - A C - (W (Y_block - b0_block - A_prev C_prev) + b0_new).  With the resident residual of (A_prev, C_prev),
// Ysig = Yc + (Ymean - b0) - W Yc + (W A_prev)(C_prev - mean):  E = Ysig + kappa - A C,
// kappa = b0 - b0_new - W (Ymean_block - b0_block) + (W A_prev) mean(C_prev)  -- a per-pixel constant (k_rss_const), so the sum of squares is
// ONE streaming read of Ysig with the footprint rows A(m,:) applied through the wave-uniform trace lists of the delta kernel.
__global__ void __launch_bounds__(256) k_rss_const(int64_t d, int nr, int nr_b, int nc_b, int roff, int coff, int p, const int *__restrict__ dr,
                                                   const int *__restrict__ dc, const float *__restrict__ W, const float *__restrict__ ymean_f,
                                                   const double *__restrict__ b0, const float *__restrict__ b0blk, const float *__restrict__ b0new,
                                                   const int *__restrict__ cnt, const int *__restrict__ wk, const float *__restrict__ wv,
                                                   const double *__restrict__ cm, float *__restrict__ kappa) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= d) return;
    const int rbm = (int)(m % nr) + roff, cbm = (int)(m / nr) + coff;
    double s = 0.0;
    for (int i = 0; i < p; ++i) {
        const float w = W[(int64_t)i * d + m];
        const int rb = rbm + dr[i], cb = cbm + dc[i];
        if (w == 0.f || rb < 0 || rb >= nr_b || cb < 0 || cb >= nc_b) continue;
        const int64_t q = (int64_t)cb * nr_b + rb;
        s += (double)w * ((double)ymean_f[q] - (double)b0blk[q]);
    }
    double t = 0.0;
    if (cnt) { const int n = cnt[m]; for (int e = 0; e < n; ++e) t += (double)wv[(int64_t)e * d + m] * cm[wk[(int64_t)e * d + m]]; }
    kappa[m] = (float)(b0[m] - (double)b0new[m] - s + t);
}

__global__ void __launch_bounds__(256) k_rss(const float4 *__restrict__ ysig4, int64_t d, int64_t T, int64_t Tc, int64_t cseg, const float *__restrict__ kappa,
                                             const int *__restrict__ cnt, const int *__restrict__ ak, const float *__restrict__ av,
                                             const float *__restrict__ C, int64_t ldc, double *__restrict__ partial) {
    __shared__ double red[4];
    const int64_t m0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = m0 < d;
    const int64_t m = valid ? m0 : d - 1;
    const int n = (cnt && valid) ? cnt[m] : 0;
    int ke[DELTA_NE]; float ve[DELTA_NE];
#pragma unroll
    for (int e = 0; e < DELTA_NE; ++e) { ke[e] = 0; ve[e] = 0.f; if (e < n) { ke[e] = ak[(int64_t)e * d + m]; ve[e] = av[(int64_t)e * d + m]; } }
    DeltaPlan pl; pl.nl = 0; pl.over = false;
    if (cnt) delta_plan(n, ke, ve, ldc, -1.f, pl);
    const float kap = valid ? kappa[m] : 0.f;
    const int64_t c0 = (int64_t)blockIdx.y * cseg, c1 = c0 + cseg < Tc ? c0 + cseg : Tc;
    double acc = 0.0;
    for (int64_t c = c0; c < c1; ++c) {
        float4 y = ysig4[c * d + m];
        y.x += kap; y.y += kap; y.z += kap; y.w += kap;
        if (!pl.over) {
#pragma unroll
            for (int j = 0; j < DELTA_NL; ++j)
                if (j < pl.nl) {
                    const float4 t = *reinterpret_cast<const float4 *>(C + pl.koff[j] + 4 * c);
                    y.x = fmaf(pl.w[j], t.x, y.x); y.y = fmaf(pl.w[j], t.y, y.y); y.z = fmaf(pl.w[j], t.z, y.z); y.w = fmaf(pl.w[j], t.w, y.w);
                }
        } else {
            for (int e = 0; e < n; ++e) {
                const float w = -av[(int64_t)e * d + m];
                const float4 t = *reinterpret_cast<const float4 *>(C + (int64_t)ak[(int64_t)e * d + m] * ldc + 4 * c);
                y.x = fmaf(w, t.x, y.x); y.y = fmaf(w, t.y, y.y); y.z = fmaf(w, t.z, y.z); y.w = fmaf(w, t.w, y.w);
            }
        }
        if (valid) {
            const int64_t t0 = 4 * c;
            acc += (double)y.x * y.x;
            if (t0 + 1 < T) acc += (double)y.y * y.y;
            if (t0 + 2 < T) acc += (double)y.z * y.z;
            if (t0 + 3 < T) acc += (double)y.w * y.w;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

int rss_run(cnmfe_ctx *ctx, Patch *P, int32_t K, const int64_t *A_colptr, const int32_t *A_rowidx, const float *A_val, const float *C, int c_order,
            const float *b0_block, const float *b0_new, double *rss_out) {
    RET(residual_materialize(ctx, P));
    const int64_t T = P->T, d = P->d;
    DevBuf &dC = ctx->tmp[0], &dCnt = ctx->tmp[3], &dK = ctx->tmp[4], &dV = ctx->tmp[5], &dB0b = ctx->tmp[6], &dB0n = ctx->tmp[7], &dKap = ctx->tmp[12], &dPart = ctx->tmp[13];
    const bool has_a = K > 0 && A_colptr[K] > 0;
    int64_t ldc = 4;
    std::vector<int> cnt; std::vector<int> ek; std::vector<float> ev;
    if (has_a) {
        RET(upload_traces(ctx, dC, C, K, T, c_order, &ldc));
        cnt.assign((size_t)d, 0);
        for (int32_t k = 0; k < K; ++k) for (int64_t e = A_colptr[k]; e < A_colptr[k + 1]; ++e) cnt[A_rowidx[e]]++;
        int cap = 1; for (int64_t m = 0; m < d; ++m) cap = std::max(cap, cnt[m]);
        ek.assign((size_t)cap * d, 0); ev.assign((size_t)cap * d, 0.f);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int32_t k = 0; k < K; ++k)
            for (int64_t e = A_colptr[k]; e < A_colptr[k + 1]; ++e) { const int64_t m = A_rowidx[e]; const int s_ = cnt[m]++; ek[(size_t)s_ * d + m] = k; ev[(size_t)s_ * d + m] = A_val[e]; }
        RET(to_dev(ctx, dCnt, cnt.data(), cnt.size())); RET(to_dev(ctx, dK, ek.data(), ek.size())); RET(to_dev(ctx, dV, ev.data(), ev.size()));
    }
    RET(to_dev(ctx, dB0b, b0_block, (size_t)P->d_b)); RET(to_dev(ctx, dB0n, b0_new, (size_t)d));
    RET(dKap.ensure((size_t)d * sizeof(float)));
    LAUNCH(ctx, "rss_const", k_rss_const, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, d, P->nr, P->nr_b, P->nc_b, P->roff, P->coff, P->p, P->ring_dr.as<int>(),
           P->ring_dc.as<int>(), P->W.as<float>(), P->ymean_f.as<float>(), P->b0.as<double>(), dB0b.as<float>(), dB0n.as<float>(),
           P->res_ac ? P->resCnt.as<int>() : nullptr, P->resK.as<int>(), P->resV.as<float>(), P->resCm.as<double>(), dKap.as<float>());
    const int64_t nblk = (d + 255) / 256;
    int64_t nseg = std::max<int64_t>(1, std::min<int64_t>(P->Tc, (8192 + nblk - 1) / nblk));
    const int64_t cseg = (P->Tc + nseg - 1) / nseg;
    nseg = (P->Tc + cseg - 1) / cseg;
    RET(dPart.ensure((size_t)nblk * nseg * sizeof(double)));
    LAUNCH(ctx, "rss_sweep", k_rss, dim3((unsigned)nblk, (unsigned)nseg), dim3(256), 0, P->ysig.as<float4>(), d, T, P->Tc, cseg, dKap.as<float>(),
           has_a ? dCnt.as<int>() : nullptr, dK.as<int>(), dV.as<float>(), dC.as<float>(), ldc, dPart.as<double>());
    std::vector<double> part((size_t)nblk * nseg);
    CK(hipMemcpyAsync(part.data(), dPart.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->st()));
    RET(ctx_check_errflag(ctx));
    double s = 0.0;
    for (double v : part) s += v;                              // fixed order: reproducible
    *rss_out = s;
    return 0;
}

// ---- reconstruct_background (Sources2D.m:1247-1355), ring model, bg_ssub = 1 --------------------------------------------------
// Ybg = W (Y_block - b0_block - A_prev C_prev) + b0_new = Y(patch) - Ysig - kappa  with the resident residual of (A_prev, C_prev) and the
// per-pixel constant kappa of compute_RSS: one streaming pass, frames [frame0, frame0 + nframes) written frame-major (d x nframes).
__global__ void __launch_bounds__(256) k_bg_out(const float4 *__restrict__ yc4, int64_t d_b, int nr, int nr_b, int roff, int coff, const float4 *__restrict__ ysig4,
                                                int64_t d, const float *__restrict__ ymean_f, const float *__restrict__ kappa, int64_t frame0, int64_t nframes,
                                                float *__restrict__ out) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= d) return;
    const int64_t q = (int64_t)((int)(m / nr) + coff) * nr_b + (int)(m % nr) + roff;
    const float cst = ymean_f[q] - kappa[m];
    const int64_t t = frame0 + blockIdx.y, c = t >> 2; const int u = (int)(t & 3);
    const float y = reinterpret_cast<const float *>(yc4 + c * d_b + q)[u], s = reinterpret_cast<const float *>(ysig4 + c * d + m)[u];
    out[(int64_t)blockIdx.y * d + m] = y - s + cst;
}

int bg_reconstruct_run(cnmfe_ctx *ctx, Patch *P, const float *b0_block, const float *b0_new, int64_t frame0, int64_t nframes, float *out, int out_memspace) {
    RET(residual_materialize(ctx, P));
    const int64_t d = P->d;
    DevBuf &dB0b = ctx->tmp[6], &dB0n = ctx->tmp[7], &dKap = ctx->tmp[12];
    RET(to_dev(ctx, dB0b, b0_block, (size_t)P->d_b)); RET(to_dev(ctx, dB0n, b0_new, (size_t)d));
    RET(dKap.ensure((size_t)d * sizeof(float)));
    LAUNCH(ctx, "rss_const", k_rss_const, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, d, P->nr, P->nr_b, P->nc_b, P->roff, P->coff, P->p, P->ring_dr.as<int>(),
           P->ring_dc.as<int>(), P->W.as<float>(), P->ymean_f.as<float>(), P->b0.as<double>(), dB0b.as<float>(), dB0n.as<float>(),
           P->res_ac ? P->resCnt.as<int>() : nullptr, P->resK.as<int>(), P->resV.as<float>(), P->resCm.as<double>(), dKap.as<float>());
    float *dst = out;
    if (out_memspace != CNMFE_DEVICE) { RET(ctx->stage.ensure((size_t)d * nframes * sizeof(float))); dst = ctx->stage.as<float>(); }
    LAUNCH(ctx, "bg_reconstruct", k_bg_out, dim3((unsigned)((d + 255) / 256), (unsigned)nframes), dim3(256), 0, P->Yc4.as<float4>(), P->d_b, P->nr, P->nr_b, P->roff, P->coff,
           P->ysig.as<float4>(), d, P->ymean_f.as<float>(), dKap.as<float>(), frame0, nframes, dst);
    if (out_memspace != CNMFE_DEVICE) CK(hipMemcpyAsync(out, dst, (size_t)d * nframes * sizeof(float), hipMemcpyDeviceToHost, ctx->st()));
    return ctx_check_errflag(ctx);
}

// The ring sweep itself: Ysig = Yc(patch) + (Ymean - b0) - W Yc_block [+ (W A_prev)(C_prev - mean) when `has_ac`: the ELL rows of W A_prev in tmp[8..10], the
// centred traces in tmp[1]] into `ysig`.  can_defer: the caller can leave the footprint term PENDING beside Ysig instead (then *deferred is set and the sweep
// runs without it -- the fastest kernel, see below).
static int r1_sweep(cnmfe_ctx *ctx, Patch *P, DevBuf &ysig, bool has_ac, int64_t ldc, bool can_defer, bool *deferred) {
    const int64_t T = P->T;
    DevBuf &dCc = ctx->tmp[1], &dOffs = ctx->tmp[6], &dDlt = ctx->tmp[7], &dWaCnt = ctx->tmp[8], &dWaK = ctx->tmp[9], &dWaV = ctx->tmp[10];
    RET(dDlt.ensure(P->d * sizeof(float)));
    LAUNCH(ctx, "r1_dlt", k_dlt, dim3((unsigned)((P->d + 255) / 256)), dim3(256), 0,
           P->ymean_f.as<float>(), P->b0.as<double>(), dDlt.as<float>(), P->d, P->nr, P->nr_b, P->roff, P->coff);

    // is the resident ring the full get_nhood(radius) ring?  (then a compile-time kernel exists for 15 / 18)
    int variant = (int)ctx->opt("r1_variant", 14);
    const int h = P->radius;
    if (variant >= 0 && variant != 10 && variant != 11 && variant != 14) variant = 14;
    bool full_ring = true;
    { int n = 0;
      for (int c = -h; c <= h && full_ring; ++c) for (int r = -h; r <= h; ++r) { int d2 = c * c + r * r;
          if (d2 >= h * h && d2 < (h + 1) * (h + 1)) { if (n >= P->p || P->dr[n] != r || P->dc[n] != c) { full_ring = false; break; } ++n; } }
      if (n != P->p) full_ring = false; }
    // The footprint term (W A_prev)(C_prev - mean) of a patch with halo neurons need not go through the sweep at all: the sweep runs without it (so the
    // faster duo-role kernel serves patched runs too) and the term stays PENDING beside Ysig -- the spatial and the temporal update both take it in
    // algebraically, through their projections (residual_term_fold_spatial, residual_term_project); any other consumer materialises it first.
    const bool defer_term = has_ac && variant == 14 && h == 15 && full_ring && can_defer && ctx->opt("r1_lazy", 1) != 0 && ctx->opt("r1_defer", 1) != 0;
    const bool sweep_ac = has_ac && !defer_term;
    if (variant == 14 && (sweep_ac || h != 15)) variant = 11;   // duo roles (resid_duo.hpp): radius 15, no footprint term inside the sweep
    if (h == 18 && variant >= 11) variant = 10;               // arc kernels: radius 15 only (ds_read immediates)
    // the low-resolution rings of bg_ssub = 2, 3 (ceil(15/2) = 8, ceil(18/2) = 9, ceil(15/3) = 5, ceil(18/3) = 6): LDS-DMA kernel only
    const bool small_special = full_ring && (h == 5 || h == 6 || h == 8 || h == 9) && variant >= 0;
    if (small_special) variant = 10;
    const bool special = (full_ring && (h == 15 || h == 18) && variant >= 0) || small_special;
    int TR = 16, TC = 16;
    if (special) tile_shape(variant, TR, TC);
    const int HR = TR + 2 * h, HC = TC + 2 * h;

    R1Args a;
    a.Y4 = P->Yc4.as<float4>(); a.d_b = P->d_b; a.nr_b = P->nr_b; a.nc_b = P->nc_b;
    a.nr = P->nr; a.nc = P->nc; a.roff = P->roff; a.coff = P->coff; a.d = P->d;
    a.T = T;
    a.W = P->W.as<float>(); a.p = P->p; a.h = h; a.offs = nullptr;
    a.ymean_f = P->ymean_f.as<float>(); a.dlt = dDlt.as<float>();
    a.wa_cnt = sweep_ac ? dWaCnt.as<int>() : nullptr; a.wa_k = sweep_ac ? dWaK.as<int>() : nullptr;
    a.wa_v = sweep_ac ? dWaV.as<float>() : nullptr; a.Cc = sweep_ac ? dCc.as<float>() : nullptr; a.ldc = ldc;
    a.Ysig4 = ysig.as<float4>();
    a.ntile_r = (P->nr + TR - 1) / TR;
    a.probe = (int)ctx->opt("r1_probe", 0);
    const int ntile_c = (P->nc + TC - 1) / TC;
    const int64_t ntiles = (int64_t)a.ntile_r * ntile_c;
    // enough workgroups to fill 256 CUs several times over; segments are a multiple of 4 frames
    int64_t nseg = std::max<int64_t>(1, std::min<int64_t>((T + 255) / 256, (4096 + ntiles - 1) / ntiles));
    if (variant == 14) nseg = std::max<int64_t>(1, std::min<int64_t>((T + 255) / 256, (512 + ntiles - 1) / ntiles));   // 1024-pixel tiles: two rounds of the chip; every segment re-reads W
    int64_t tseg = ((T + nseg - 1) / nseg + 3) & ~int64_t(3);
    nseg = (T + tseg - 1) / tseg;
    a.tseg = tseg;
    int rc;
    if (special) {
        RET(build_tile_map(ctx, dOffs, a.ntile_r, ntile_c, TR, TC, h, 1));
        a.tile_map = dOffs.as<int>();
        const dim3 gridd((unsigned)((int64_t)a.ntile_r * ntile_c), (unsigned)nseg);
        if (h == 15) rc = launch_r1_v<15>(ctx, variant, a, sweep_ac, ntile_c, nseg);
        else if (h == 18) rc = launch_r1_v<18>(ctx, variant, a, sweep_ac, ntile_c, nseg);
        else if (h == 8) rc = launch_r1_dma<8, 32, 16>(ctx, a, sweep_ac, gridd);
        else if (h == 9) rc = launch_r1_dma<9, 32, 16>(ctx, a, sweep_ac, gridd);
        else if (h == 5) rc = launch_r1_dma<5, 32, 16>(ctx, a, sweep_ac, gridd);
        else rc = launch_r1_dma<6, 32, 16>(ctx, a, sweep_ac, gridd);
    } else {
        std::vector<int32_t> offs(std::max(1, P->p), 0);
        for (int i = 0; i < P->p; ++i) offs[i] = P->dc[i] * HR + P->dr[i];
        RET(to_dev(ctx, dOffs, offs.data(), offs.size()));
        a.offs = dOffs.as<int>();
        CK(hipStreamSynchronize(ctx->st()));
        size_t shmem = (size_t)HR * HC * sizeof(float4);
        if (shmem > 160 * 1024) return fail(CNMFE_EUNSUPPORTED, "ring radius %d needs %zu B of LDS per tile", h, shmem);
        if (shmem > 64 * 1024) {
            CK(hipFuncSetAttribute((const void *)k_residual_gen<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            CK(hipFuncSetAttribute((const void *)k_residual_gen<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        }
        dim3 grid((unsigned)ntiles, (unsigned)nseg);
        if (sweep_ac) LAUNCH(ctx, "residual_r1_generic", k_residual_gen<true>, grid, dim3(256), shmem, a);
        else          LAUNCH(ctx, "residual_r1_generic", k_residual_gen<false>, grid, dim3(256), shmem, a);
        rc = 0;
    }
    RET(rc);
    if (deferred) *deferred = defer_term;
    return 0;
}

// a virtual residual becomes a resident one: the sweep without any footprint term (the term, if any, is pending beside it either way)
int residual_realize(cnmfe_ctx *ctx, Patch *P) {
    if (!P->ysig_virtual) return 0;
    if (!P->ysig_valid) { P->ysig_virtual = false; return 0; }
    if (P->res_kind == 2) return ssub_realize(ctx, P);      // bg_ssub > 1: the low-resolution sweep and its upsample (ssub.hip)
    RET(P->ysig.ensure((size_t)P->d * P->Tc * sizeof(float4)));
    RET(r1_sweep(ctx, P, P->ysig, false, 4, false, nullptr));
    P->ysig_virtual = false;
    return 0;
}

int residual_run(cnmfe_ctx *ctx, Patch *P, int pid, int32_t Ksel, const int64_t *A_colptr, const int32_t *A_rowidx,
                 const float *A_val, const float *C, int c_order, float *Ysig_out, int out_memspace, DevBuf *outbuf, int tables_only) {
    const int64_t T = P->T;
    DevBuf &ysig = outbuf ? *outbuf : P->ysig;               // bg_ssub > 1 sweeps the low-resolution patch into its own buffer
    // the resident Ysig of this patch is still the residual under the current video, W and b0: only the footprint term changes
    const bool delta = !outbuf && P->ysig_valid && P->res_kind == 1 && (P->ysig.p || P->ysig_virtual) && ctx->opt("r1_delta", 1) != 0;
    DevBuf &dC = ctx->tmp[0], &dCc = ctx->tmp[1], &dCm = ctx->tmp[2], &dArow = ctx->tmp[3], &dAcol = ctx->tmp[4], &dAval = ctx->tmp[5],
           &dWaCnt = ctx->tmp[8], &dWaK = ctx->tmp[9], &dWaV = ctx->tmp[10];
    int64_t ldc = 4;
    bool has_ac = Ksel > 0 && A_colptr[Ksel] > 0;
    if (has_ac) {
        RET(upload_centered(ctx, dC, C, Ksel, T, c_order, dCc, dCm, &ldc));
        HostCSR csr; csc_to_csr(P->d_b, Ksel, A_colptr, A_rowidx, A_val, csr);
        RET(to_dev(ctx, dArow, csr.rowptr.data(), csr.rowptr.size()));
        RET(to_dev(ctx, dAcol, csr.col.data(), csr.col.size()));
        RET(to_dev(ctx, dAval, csr.val.data(), csr.val.size()));
        RET(dWaCnt.ensure_hw(P->d * sizeof(int), ctx->hw_wa[0]));
        const int wa_cap = std::max(WA_CAP, std::min((int)Ksel, WA_CAP_MAX));        // a ring cannot touch more footprints than there are
        RET(dWaK.ensure_hw((size_t)wa_cap * P->d * sizeof(int), ctx->hw_wa[1]));
        RET(dWaV.ensure_hw((size_t)wa_cap * P->d * sizeof(float), ctx->hw_wa[2]));
        // a ring that touches more than WA_CAP_MAX footprints raises the context's error flag (ctx_check_errflag at the next wait of this call chain: the
        // spatial / temporal update's own download) instead of costing a drain of the stream here -- with several patches per context that drain
        // was what kept the host from setting up patch m + 1 under patch m's kernels
        int *dErrWa = nullptr;
        RET(ctx_errflag(ctx, &dErrWa));
        const int nnzA = (int)csr.col.size();
        const size_t wa_stage = (size_t)nnzA * 8;
        if (wa_stage <= 96 * 1024) {                               // (32 KB static + this: one workgroup per CU above ~48 KB, still every CU of a small patch's grid)
            if (wa_stage > 32 * 1024)                                // (per device: set where it is needed, not once per process)
                CK(hipFuncSetAttribute((const void *)k_ring_wa<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            LAUNCH(ctx, "r1_ring_wa", k_ring_wa<true>, dim3((unsigned)((P->d + 127) / 128)), dim3(128), wa_stage, P->W.as<float>(), P->d, P->nr, P->nr_b, P->nc_b,
                   P->roff, P->coff, P->p, P->ring_dr.as<int>(), P->ring_dc.as<int>(), dArow.as<int>(), dAcol.as<int>(), dAval.as<float>(), nnzA,
                   dWaCnt.as<int>(), dWaK.as<int>(), dWaV.as<float>(), dErrWa, wa_cap);
        } else
            LAUNCH(ctx, "r1_ring_wa", k_ring_wa<false>, dim3((unsigned)((P->d + 127) / 128)), dim3(128), 0, P->W.as<float>(), P->d, P->nr, P->nr_b, P->nc_b,
                   P->roff, P->coff, P->p, P->ring_dr.as<int>(), P->ring_dc.as<int>(), dArow.as<int>(), dAcol.as<int>(), dAval.as<float>(), nnzA,
                   dWaCnt.as<int>(), dWaK.as<int>(), dWaV.as<float>(), dErrWa, wa_cap);
    }
    if (tables_only) { ctx->last_ldc = ldc; return 0; }       // bg_ssub: the caller only wants (W*A) and the centred traces (tmp[8..10], tmp[1])
    // the footprint term this call leaves applied is kept beside Ysig (the scratch buffers above change hands with the patch's)
    auto keep = [&]() {
        if (outbuf) return;
        P->res_ac = has_ac; P->res_ldc = ldc; P->res_kind = 1; P->res_K = Ksel; P->pend = false;
        if (has_ac) { P->resCnt.swap(dWaCnt); P->resK.swap(dWaK); P->resV.swap(dWaV); P->resCc.swap(dCc); P->resCm.swap(dCm); }
    };
    if (delta) {
        // lazy: keep the term pending; cnmfe_hals_temporal folds it in algebraically, anybody else materialises it
        if (ctx->opt("r1_lazy", 1) != 0 && !Ysig_out) {
            P->pend = true; P->pend_ac = has_ac; P->pend_ldc = ldc; P->pend_K = Ksel;
            if (has_ac) { P->pendCnt.swap(dWaCnt); P->pendK.swap(dWaK); P->pendV.swap(dWaV); P->pendCc.swap(dCc); P->pendCm.swap(dCm); }
            return 0;
        }
        P->pend = true; P->pend_ac = has_ac; P->pend_ldc = ldc; P->pend_K = Ksel;
        if (has_ac) { P->pendCnt.swap(dWaCnt); P->pendK.swap(dWaK); P->pendV.swap(dWaV); P->pendCc.swap(dCc); P->pendCm.swap(dCm); }
        RET(residual_materialize(ctx, P));
        if (Ysig_out) RET(ysig_export(ctx, P, ysig, Ysig_out, out_memspace));
        return 0;
    }
    // Sweep-free residual (round 4, option "r1_virtual", default 1): nobody has asked for Ysig itself -- the request is only recorded.  The spatial update
    // takes Ysig C' out of the table P = Yc Cc' (U = P - W P), the temporal update projects the centred video through B = A - W'A (vproj.hip), a footprint
    // term pends beside the virtual Ysig exactly as beside a resident one; every other consumer goes through residual_realize, which runs the sweep then.
    if (!outbuf && !Ysig_out && ctx->opt("r1_virtual", 1) != 0 && ctx->opt("r1_lazy", 1) != 0 && ctx->opt("r1_delta", 1) != 0) {
        P->res_ac = false; P->res_ldc = ldc; P->res_kind = 1; P->res_K = 0;
        P->pend = has_ac; P->pend_ac = has_ac; P->pend_ldc = ldc; P->pend_K = has_ac ? Ksel : 0;
        if (has_ac) { P->pendCnt.swap(dWaCnt); P->pendK.swap(dWaK); P->pendV.swap(dWaV); P->pendCc.swap(dCc); P->pendCm.swap(dCm); }
        ctx->last_ldc = ldc;
        P->ysig_valid = true; P->ysig_virtual = true;
        return 0;
    }
    RET(ysig.ensure((size_t)P->d * P->Tc * sizeof(float4)));
    bool defer_term = false;
    RET(r1_sweep(ctx, P, ysig, has_ac, ldc, !outbuf && !Ysig_out, &defer_term));
    if (!outbuf) P->ysig_virtual = false;
    if (defer_term) {                                        // Ysig carries no term; the one asked for waits beside it
        P->res_ac = false; P->res_ldc = ldc; P->res_kind = 1; P->res_K = 0;
        P->pend = true; P->pend_ac = true; P->pend_ldc = ldc; P->pend_K = Ksel;
        P->pendCnt.swap(dWaCnt); P->pendK.swap(dWaK); P->pendV.swap(dWaV); P->pendCc.swap(dCc); P->pendCm.swap(dCm);
    } else
        keep();
    ctx->last_ldc = ldc;
    P->ysig_valid = true;
    if (Ysig_out) RET(ysig_export(ctx, P, ysig, Ysig_out, out_memspace));
    // without an output buffer the call returns with the kernel in flight: every consumer of Ysig is an engine
    // call on the same stream, so the caller's host work overlaps the sweep (include/cnmfe.h, cnmfe_residual)
    return 0;
}

// loads this translation unit's code object now (HIP loads it at the first launch of one of its kernels -- milliseconds each that would otherwise fall into the first iteration): cnmfe_create
int tu_warm_resid() { hipFuncAttributes at; return hipFuncGetAttributes(&at, (const void *)k_center_traces) == hipSuccess ? 0 : -1; }

}  // namespace cnmfe
