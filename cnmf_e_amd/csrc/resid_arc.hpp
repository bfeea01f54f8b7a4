// R1 "arc roles": the ring product with LDS-value reuse.   (included by resid.hip; gfx950 only)
//
// In the one-pixel-per-thread kernel every FMA group needs its own 16-byte LDS read, so the LDS pipe
// (256 B/clk/CU) caps the kernel at ~40 % of HBM.  Here the ring is split into four arcs (left/right: runs of
// consecutive row offsets at a fixed column offset; top/bottom: runs of consecutive column offsets at a fixed
// row offset) and a thread owns P adjacent centre pixels ALONG the run direction of its arc.  One staged value
// R'(q) then feeds up to P centres (q - m in the arc for several of the thread's m), so a run of length L costs
// L+P-1 reads for P*L FMA groups.  Each centre is covered by four threads (one per arc); their partial sums meet
// in an LDS tile and a final pass adds the centre term and stores Ysig.
//
// Tile 16 rows x 32 columns (512 centres), halo row stride HRp odd (== 1 mod 16) so that both lane->column
// (vertical roles) and lane->row with a per-quarter rotation (horizontal roles) hit 16 distinct 16-byte slots in
// every ds_read_b128 lane group.  Threads = 4 roles x 512/P.
#pragma once
#include <type_traits>

namespace cnmfe {

constexpr int ARC_TR = 16, ARC_TC = 32;

template <int R> struct ArcTab {
    int n[4];                 // offsets per arc: 0 = left, 1 = right, 2 = top, 3 = bottom
    int ring[4][40];          // index into the full ring (W rows), in run order
    int nrun[4];
    int rfix[4][24];          // fixed offset of the run (dc for arcs 0/1, dr for arcs 2/3)
    int rs[4][24];            // first moving offset of the run
    int rl[4][24];            // run length
    int ra0[4][24];           // arc-local index of the run's first offset
};

template <int R> constexpr ArcTab<R> make_arcs() {
    ArcTab<R> t{};
    constexpr RingTab<R> ring = make_ring<R>();
    int fx[4][40] = {}, mv[4][40] = {}, id[4][40] = {};
    for (int a = 0; a < 4; ++a) t.n[a] = 0;
    for (int i = 0; i < ring.n; ++i) {
        const int dr = ring.dr[i], dc = ring.dc[i];
        const int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
        int a = 0;
        if (adc > adr) a = dc < 0 ? 0 : 1;
        else if (adr > adc) a = dr < 0 ? 2 : 3;
        else a = (dc < 0 && dr < 0) ? 0 : (dc > 0 && dr > 0) ? 1 : (dr < 0 ? 2 : 3);
        const int k = t.n[a]++;
        fx[a][k] = a < 2 ? dc : dr; mv[a][k] = a < 2 ? dr : dc; id[a][k] = i;
    }
    for (int a = 0; a < 4; ++a) {
        // sort by (fixed, moving)
        for (int i = 1; i < t.n[a]; ++i)
            for (int j = i; j > 0 && (fx[a][j] < fx[a][j - 1] || (fx[a][j] == fx[a][j - 1] && mv[a][j] < mv[a][j - 1])); --j) {
                int x = fx[a][j]; fx[a][j] = fx[a][j - 1]; fx[a][j - 1] = x;
                x = mv[a][j]; mv[a][j] = mv[a][j - 1]; mv[a][j - 1] = x;
                x = id[a][j]; id[a][j] = id[a][j - 1]; id[a][j - 1] = x;
            }
        t.nrun[a] = 0;
        for (int i = 0; i < t.n[a]; ++i) {
            t.ring[a][i] = id[a][i];
            if (i > 0 && fx[a][i] == fx[a][i - 1] && mv[a][i] == mv[a][i - 1] + 1) { t.rl[a][t.nrun[a] - 1]++; continue; }
            const int r = t.nrun[a]++;
            t.rfix[a][r] = fx[a][i]; t.rs[a][r] = mv[a][i]; t.rl[a][r] = 1; t.ra0[a][r] = i;
        }
    }
    return t;
}
template <int R> struct ArcConst { static constexpr ArcTab<R> tab = make_arcs<R>(); };

// flat per-role program for P centres per thread: the LDS reads in order, and for every read the (centre j,
// arc-local weight index a) pairs it feeds
template <int R, int P> struct ArcProg {
    int nl[4];
    int fix[4][96], mov[4][96];       // offsets of the value read: (dr, dc) = ARC < 2 ? (mov, fix) : (fix, mov)
    int nf[4][96];
    int fj[4][96][4], fa[4][96][4];
};
template <int R, int P> constexpr ArcProg<R, P> make_prog() {
    ArcProg<R, P> g{};
    constexpr ArcTab<R> t = make_arcs<R>();
    for (int arc = 0; arc < 4; ++arc) {
        int li = 0;
        for (int run = 0; run < t.nrun[arc]; ++run)
            for (int x = 0; x < t.rl[arc][run] + P - 1; ++x) {
                g.fix[arc][li] = t.rfix[arc][run]; g.mov[arc][li] = t.rs[arc][run] + x;
                int nf = 0;
                for (int j = 0; j < P; ++j) {
                    const int u = x - j;
                    if (u >= 0 && u < t.rl[arc][run]) { g.fj[arc][li][nf] = j; g.fa[arc][li][nf] = t.ra0[arc][run] + u; ++nf; }
                }
                g.nf[arc][li] = nf;
                ++li;
            }
        g.nl[arc] = li;
    }
    return g;
}
template <int R, int P> struct ProgConst { static constexpr ArcProg<R, P> tab = make_prog<R, P>(); };

// the ring product of one role: ARC in 0..3, P centres per thread.  hb = thread base in the halo:
//   vertical roles  (ARC 0/1): &halo[c * HRp + g*P]      -> value of (row g*P + i, col c + j) at hb[(j+R)*HRp + (i+R)]
//   horizontal roles(ARC 2/3): &halo[h*P * HRp + r]      -> value of (row r + i, col h*P + j) at the same expression
// wp[j][a/2] holds the arc weights of centre j as pairs; acc[j][0] = frames 0,1, acc[j][1] = frames 2,3.
template <int R, int ARC, int P, int HRp, int NW, int D = 4>
__device__ __forceinline__ void arc_product(const float4 *hb, const f2 (&wp)[P][NW], f2 (&acc)[P][2]) {
    using PC = ProgConst<R, P>;
    constexpr int NL = PC::tab.nl[ARC];                    // D = LDS reads in flight ahead of their FMAs
    float4 r[D + 1];
#define ARC_ADDR(li) (ARC < 2 ? hb + (PC::tab.fix[ARC][li] + R) * HRp + (PC::tab.mov[ARC][li] + R) \
                              : hb + (PC::tab.mov[ARC][li] + R) * HRp + (PC::tab.fix[ARC][li] + R))
#pragma unroll
    for (int li = 0; li < D; ++li) if (li < NL) r[li] = *ARC_ADDR(li);
#pragma unroll
    for (int li = 0; li < NL; ++li) {
        if (li + D < NL) r[(li + D) % (D + 1)] = *ARC_ADDR(li + D);
        const float4 rv = r[li % (D + 1)];
        const f2 r01 = {rv.x, rv.y}, r23 = {rv.z, rv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < PC::tab.nf[ARC][li]) {
                const int j = PC::tab.fj[ARC][li][q], a = PC::tab.fa[ARC][li][q];
                const f2 wv = wp[j][a >> 1];
                if ((a & 1) == 0) {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j][0]) : "v"(wv), "v"(r01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j][1]) : "v"(wv), "v"(r23));
                } else {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[j][0]) : "v"(wv), "v"(r01));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[j][1]) : "v"(wv), "v"(r23));
                }
            }
        }
        // keep later reads from being hoisted above these FMAs (register pressure), see k_residual_r
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]) : : "memory");
    }
#undef ARC_ADDR
}

// ---- arc roles on LDS-DMA staging (r1_variant 11) -------------------------------------------------------------
// The ring product of k_residual_arc (P = 4: half the LDS reads of the one-pixel-per-thread kernel) with the halo
// staged by global_load_lds_dwordx4 as in k_residual_dma: two halo buffers (the partial sums take the third's room),
// chunk c+1 in flight under chunk c, traces of the A_prev flavour staged in LDS.  Two barriers per chunk: halo
// landed / partial sums complete.
template <int R, bool HAS_AC, int ARC_D = 4>
__global__ void __launch_bounds__(ARC_TR *ARC_TC, 2) k_residual_arc_dma(R1Args a) {
    constexpr int P = 4;
    constexpr int TR = ARC_TR, TC = ARC_TC, NC = TR * TC, NT = NC, NWV = NT / 64;
    constexpr int HR = TR + 2 * R, HC = TC + 2 * R;
    constexpr int HRp = ((HR + 14) / 16) * 16 + 1;
    constexpr int NHp = HRp * HC;
    constexpr int NIT = (NHp + NT - 1) / NT, NHs = NIT * NT;          // DMA slots per buffer (lane-linear image of [HC][HRp])
    constexpr int NA = ArcConst<R>::tab.n[0];
    static_assert(ArcConst<R>::tab.n[1] == NA && ArcConst<R>::tab.n[2] == NA && ArcConst<R>::tab.n[3] == NA && NA % 2 == 0, "arcs must be balanced");
    constexpr int NW = NA / 2;
    constexpr int TRp = TR + 1, NCp = TRp * TC;
    constexpr int NBUF = 2;
    extern __shared__ __attribute__((aligned(16))) float4 lds[];      // halo[2][NHs] | part | tbuf[2][R1_TKMAX] | scratch
    float4 *halo = lds, *part = lds + NBUF * NHs, *tbuf = part + 2 * NCp + 2 * NC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tmap = a.tile_map[blockIdx.x];
    const int tile_r = tmap & 0xffff, tile_c = tmap >> 16;
    const int hr0 = tile_r * TR + a.roff - R, hc0 = tile_c * TC + a.coff - R;
    // ---- role geometry (as k_residual_arc) ----
    constexpr int TPR = NC / P;
    const int role = __builtin_amdgcn_readfirstlane(tid / TPR), rt = tid % TPR;
    int cr[P], cc[P];
    int hbase;
    if (role < 2) {
        const int c = rt & 31, g = rt >> 5;
#pragma unroll
        for (int j = 0; j < P; ++j) { cr[j] = g * P + j; cc[j] = c; }
        hbase = c * HRp + g * P;
    } else {
        const int q = rt >> 4, i = rt & 15;
        const int sq = (q * P * HRp) & 15;
        const int r = (i - sq) & 15;
#pragma unroll
        for (int j = 0; j < P; ++j) { cr[j] = r; cc[j] = q * P + j; }
        hbase = q * P * HRp + r;
    }
    f2 wp[P][NW];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int pr = tile_r * TR + cr[j], pc = tile_c * TC + cc[j];
        const int64_t m = (pr < a.nr && pc < a.nc) ? (int64_t)pc * a.nr + pr : 0;
        const uint32_t mb = (uint32_t)m * 4u;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            int i0, i1;
            if (role == 0) { i0 = ArcConst<R>::tab.ring[0][2 * k]; i1 = ArcConst<R>::tab.ring[0][2 * k + 1]; }
            else if (role == 1) { i0 = ArcConst<R>::tab.ring[1][2 * k]; i1 = ArcConst<R>::tab.ring[1][2 * k + 1]; }
            else if (role == 2) { i0 = ArcConst<R>::tab.ring[2][2 * k]; i1 = ArcConst<R>::tab.ring[2][2 * k + 1]; }
            else { i0 = ArcConst<R>::tab.ring[3][2 * k]; i1 = ArcConst<R>::tab.ring[3][2 * k + 1]; }
            wp[j][k].x = ld_off(a.W + (int64_t)i0 * a.d, mb);
            wp[j][k].y = ld_off(a.W + (int64_t)i1 * a.d, mb);
        }
    }
    // ---- DMA plan: instruction j of wave w fills slots (j*NWV + w)*64 + lane of the [HC][HRp] image ----
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
    // ---- final-pass centre of this thread ----
    const int fr = tid % TR, fc = tid / TR;
    const int fpr = tile_r * TR + fr, fpc = tile_c * TC + fc;
    const bool fvalid = fpr < a.nr && fpc < a.nc;
    const int64_t fm = fvalid ? (int64_t)fpc * a.nr + fpr : 0;
    const uint32_t fmb = (uint32_t)fm * 4u;
    const float dl = ld_off(a.dlt, fmb);
    // ---- (W*A_prev): tile trace list + per-pixel (slot, weight) entries, as in k_residual_dma ----
    constexpr int WA_PRE = 8;
    int wsl[WA_PRE]; float wvv[WA_PRE]; int nwa = 0; bool spill = false;
    int ov0 = 0, ov1 = 0;
    uint32_t tkoff = 0;
    unsigned *bm = reinterpret_cast<unsigned *>(tbuf + NBUF * R1_TKMAX);
    int *pre = reinterpret_cast<int *>(bm + R1_KBM / 32);
    int *tk = pre + R1_KBM / 32 + 2;
    int2 *ovl = reinterpret_cast<int2 *>(tk + R1_TKMAX);
    if (HAS_AC) {
        for (int w = tid; w < R1_KBM / 32; w += NT) bm[w] = 0u;
        if (tid < R1_TKMAX) tk[tid] = 0;
        if (tid == 0) pre[R1_KBM / 32 + 1] = 0;
        __syncthreads();
        nwa = fvalid ? a.wa_cnt[fm] : 0;
        for (int e = 0; e < nwa; ++e) {
            const int k = a.wa_k[(int64_t)e * a.d + fm];
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
                const int sl = slot_of(a.wa_k[(int64_t)e * a.d + fm]);
                if (sl >= 0) { wsl[e] = sl; wvv[e] = a.wa_v[(int64_t)e * a.d + fm]; }
                else { wsl[e] = -2; spill = true; }
            }
        }
        if (nwa > WA_PRE) {
            const int n = nwa - WA_PRE;
            ov0 = atomicAdd(&pre[R1_KBM / 32 + 1], n); ov1 = ov0 + n;
            bool ok = ov1 <= R1_OVF;
            for (int e = WA_PRE; e < nwa && ok; ++e) {
                const int sl = slot_of(a.wa_k[(int64_t)e * a.d + fm]);
                if (sl < 0) ok = false;
                else ovl[ov0 + e - WA_PRE] = make_int2(sl, __float_as_int(a.wa_v[(int64_t)e * a.d + fm]));
            }
            if (!ok) { ov0 = ov1 = 0; spill = true; }
        }
        __syncthreads();
        tkoff = (uint32_t)tk[lane] * (uint32_t)(a.ldc * 4);
    }
    const int64_t cbeg = ((int64_t)blockIdx.y * a.tseg) >> 2;
    const int64_t tend = (int64_t)blockIdx.y * a.tseg + a.tseg < a.T ? (int64_t)blockIdx.y * a.tseg + a.tseg : a.T;
    const int64_t cend = (tend + 3) >> 2;
    const bool tw = HAS_AC && wave == 0;
    const int probe = __builtin_amdgcn_readfirstlane(a.probe);
    auto issue = [&](int64_t c) {
        if ((probe & 1) && c > cbeg + 1) return;
        const int64_t cx = c < cend ? c : cend - 1;
        const float4 *y4 = a.Y4 + cx * a.d_b;
        const int b = (int)((c - cbeg) & 1);
        const unsigned dst = lds0 + (unsigned)b * (unsigned)(NHs * 16);
#pragma unroll
        for (int j = 0; j < NIT; ++j) glds16(y4, qoff[j], dst + (unsigned)(j * NWV) * 1024u);
        if (tw) glds16(a.Cc + 4 * cx, tkoff, ldsA + (unsigned)((NBUF * NHs + 2 * NCp + 2 * NC) + b * R1_TKMAX) * 16u);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue(cbeg);
    for (int64_t c = cbeg; c < cend; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's part of chunk c (and its stores of chunk c-1)
        __builtin_amdgcn_s_barrier();                           // halo(c) complete; everybody is past the final pass of c-1
        asm volatile("" ::: "memory");
        issue(c + 1);
        const int cb_ = (int)((c - cbeg) & 1);
        const float4 *hb = halo + cb_ * NHs + hbase;
        f2 acc[P][2];
#pragma unroll
        for (int j = 0; j < P; ++j) { acc[j][0] = (f2){0.f, 0.f}; acc[j][1] = (f2){0.f, 0.f}; }
        if (probe & 2) { }
        else if (role == 0) arc_product<R, 0, P, HRp, NW, ARC_D>(hb, wp, acc);
        else if (role == 1) arc_product<R, 1, P, HRp, NW, ARC_D>(hb, wp, acc);
        else if (role == 2) arc_product<R, 2, P, HRp, NW, ARC_D>(hb, wp, acc);
        else arc_product<R, 3, P, HRp, NW, ARC_D>(hb, wp, acc);
        if (!(probe & 4)) {
#pragma unroll
        for (int j = 0; j < P; ++j)
            part[role < 2 ? role * NCp + cc[j] * TRp + cr[j] : 2 * NCp + (role - 2) * NC + cc[j] * TR + cr[j]] =
                make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // partial sums complete
        asm volatile("" ::: "memory");
        }
        {
            const int ci = fc * TR + fr, cv_ = fc * TRp + fr;
            float4 p0 = make_float4(acc[0][0].x, acc[1][0].x, acc[2][0].x, acc[3][0].x), p1 = p0, p2 = p0, p3 = p0;
            if (!(probe & 4)) { p0 = part[cv_]; p1 = part[NCp + cv_]; p2 = part[2 * NCp + ci]; p3 = part[2 * NCp + NC + ci]; }
            float4 cv = halo[cb_ * NHs + (fc + R) * HRp + (fr + R)];
            if (HAS_AC) {
                const float4 *tb = tbuf + cb_ * R1_TKMAX;
#pragma unroll
                for (int e = 0; e < WA_PRE; ++e) {
                    const float4 c4 = tb[wsl[e] >= 0 ? wsl[e] : 0];
                    cv.x = fmaf(wvv[e], c4.x, cv.x); cv.y = fmaf(wvv[e], c4.y, cv.y);
                    cv.z = fmaf(wvv[e], c4.z, cv.z); cv.w = fmaf(wvv[e], c4.w, cv.w);
                }
                for (int q = ov0; q < ov1; ++q) {
                    const int2 en = ovl[q];
                    const float4 c4 = tb[en.x]; const float v = __int_as_float(en.y);
                    cv.x = fmaf(v, c4.x, cv.x); cv.y = fmaf(v, c4.y, cv.y); cv.z = fmaf(v, c4.z, cv.z); cv.w = fmaf(v, c4.w, cv.w);
                }
            }
            if (fvalid && !((probe & 8) && cv.x != 12345.f)) {
                if (HAS_AC && spill) {
                    for (int e = 0; e < nwa; ++e) {
                        if (e < WA_PRE ? wsl[e] != -2 : ov1 > ov0) continue;
                        const float v = a.wa_v[(int64_t)e * a.d + fm];
                        const float4 c4 = *reinterpret_cast<const float4 *>(a.Cc + (int64_t)a.wa_k[(int64_t)e * a.d + fm] * a.ldc + (c << 2));
                        cv.x = fmaf(v, c4.x, cv.x); cv.y = fmaf(v, c4.y, cv.y); cv.z = fmaf(v, c4.z, cv.z); cv.w = fmaf(v, c4.w, cv.w);
                    }
                }
                const float4 yo = make_float4(cv.x + dl - ((p0.x + p1.x) + (p2.x + p3.x)), cv.y + dl - ((p0.y + p1.y) + (p2.y + p3.y)),
                                              cv.z + dl - ((p0.z + p1.z) + (p2.z + p3.z)), cv.w + dl - ((p0.w + p1.w) + (p2.w + p3.w)));
                if (probe & 16) st4_off_wt(a.Ysig4 + c * a.d, fmb * 4u, yo);     // (A/B: write-through stores that drop the line from L2: +0.3 ms here)
                else st4_off(a.Ysig4 + c * a.d, fmb * 4u, yo);
            }
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int R>
static int launch_r1_arc_dma(cnmfe_ctx *ctx, const R1Args &a, bool has_ac, int ntile_c, int64_t nseg) {
    constexpr int HR = ARC_TR + 2 * R, HC = ARC_TC + 2 * R, HRp = ((HR + 14) / 16) * 16 + 1, NT = ARC_TR * ARC_TC;
    constexpr int NIT = (HRp * HC + NT - 1) / NT;
    constexpr size_t shmem = (2 * (size_t)NIT * NT + 2 * (size_t)(ARC_TR + 1) * ARC_TC + 2 * (size_t)ARC_TR * ARC_TC + 2 * R1_TKMAX) * sizeof(float4) +
                             (size_t)(R1_KBM / 32 + R1_KBM / 32 + 2 + R1_TKMAX + 2 * R1_OVF) * sizeof(int);
    static_assert(shmem <= 160 * 1024, "arc DMA kernel exceeds LDS");
    static_assert(((HC + R) * HRp + HR) * 16 < 65536, "ds_read immediate offset overflow");
    dim3 grid((unsigned)((int64_t)a.ntile_r * ntile_c), (unsigned)nseg);
    CK(hipFuncSetAttribute((const void *)k_residual_arc_dma<R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    CK(hipFuncSetAttribute((const void *)k_residual_arc_dma<R, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    if (has_ac) LAUNCH(ctx, "residual_r1", (k_residual_arc_dma<R, true>), grid, dim3(NT), shmem, a);
    else        LAUNCH(ctx, "residual_r1", (k_residual_arc_dma<R, false>), grid, dim3(NT), shmem, a);
    return 0;
}

}  // namespace cnmfe
