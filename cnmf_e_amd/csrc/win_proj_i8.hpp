// B2a', round 5: the fit's window projection  U(i, k) = sum_t Yc_i(t) Cc_k(t)  on the int8 matrix pipe.
//
// bg_win_proj (win_proj.hpp) is one read of the video against the traces near every 16 x 16 block, on the fp64 matrix pipe -- and co-bound by it: 2.9 ms at H,
// 2.0 ms with the MFMAs taken out (profiles/r05/video_pass_mfma_probe.txt).  Its result feeds the rank-2 corrections of the ring systems and needs their
// ~1e-10 relative accuracy, so reduced floating-point precision is out -- but the video's 32-bit fixed-point digit planes of gram_i8.hpp are exactly the A
// operand of v_mfma_i32_16x16x64_i8 for this product too (lane (pixel, frame group) holds 16 frames of one digit: one 16-byte global load, no LDS), and the
// centred traces get the same treatment per fit (k_trace_dig: K x T x 4 bytes).  13 int8 MFMAs (digit-pair classes p + r = 2 .. 6, exact int32 sums) replace
// 16 fp64 MFMAs of 64 clocks per 16 x 16 tile and 64 frames: a quarter of the matrix time, the kernel is left with its loads.
// The digit planes stay resident with the patch (P->dig, + one video's worth of HBM: only when that leaves 8 GB free; otherwise, and for frame strides > 1 or
// recordings beyond the int32 range, the fp64 kernel runs).  The list Gram matrix G = Cc Cc' that k_win_fix needs comes from ONE K x K fp64 product per fit
// (k_trace_gram) instead of per-block copies.
#pragma once

namespace cnmfe {

// tdig[(k * T16 + s) * 4 + plane] = 16 bytes: the digits of frames 16 s .. 16 s + 15 of centred trace k; tscale[k] = max |Cc_k| / (2^31 - 2^24)
__global__ void __launch_bounds__(256) k_trace_dig(const float *__restrict__ Cc, int64_t ldc, int64_t T, int64_t T16, uint4 *__restrict__ tdig, double *__restrict__ tscale) {
    const int k = blockIdx.x;
    const float *row = Cc + (int64_t)k * ldc;
    __shared__ float red[256];
    float m = 0.f;
    for (int64_t t = threadIdx.x; t < T; t += 256) m = fmaxf(m, fabsf(row[t]));
    red[threadIdx.x] = m; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    const double sc = red[0] > 0.f ? (double)red[0] / 2130706432.0 : 1.0;
    if (threadIdx.x == 0) tscale[k] = sc;
    const double inv = 1.0 / sc;
    for (int64_t s = threadIdx.x; s < T16; s += 256) {
        unsigned pl[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) pl[p][j] = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t t = 16 * s + j;
            int qv = t < T ? __double2int_rn((double)row[t] * inv) : 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int d = (int)(signed char)(qv & 0xff);
                pl[p][j >> 2] |= (unsigned)(d & 0xff) << (8 * (j & 3));
                qv = (qv - d) >> 8;
            }
        }
        uint4 *o = tdig + ((int64_t)k * T16 + s) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    }
}

// GK[k * K + l] = sum_t Cc_k(t) Cc_l(t), fp64 (exact fp32 products): one workgroup per 16 x 16 tile of the upper triangle (mirrored on store), its SIXTEEN waves
// a sixteenth of the frames each, eight chunk loads in flight per lane.  The kernel lasts as long as one wave's chain of dependent load -> MFMA steps whatever K is
// (one wave per tile: 0.34 ms at K = 500; a patch of a 4 x 4 decomposition has 15-30 tiles, nothing else hides the chain)
constexpr int TG_W = 16;
__global__ void __launch_bounds__(64 * TG_W) k_trace_gram(const float *__restrict__ Cc, int64_t ldc, int64_t T, int K, double *__restrict__ GK) {
    __shared__ double red[TG_W - 1][64][4];
    const int nt = (K + 15) >> 4;
    int ti = 0, rem = blockIdx.x;                          // blockIdx.x enumerates (ti <= tj)
    while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
    const int tj = ti + rem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 15, kq = lane >> 4;
    const int ka = ti * 16 + fi, kb = tj * 16 + fi;
    const float4 *ra = reinterpret_cast<const float4 *>(Cc + (int64_t)(ka < K ? ka : K - 1) * ldc);
    const float4 *rb = reinterpret_cast<const float4 *>(Cc + (int64_t)(kb < K ? kb : K - 1) * ldc);
    const int64_t nch = (T + 3) >> 2;                      // (the rows are padded to ldc >= 4 * nch and zero behind T: k_center_traces)
    const int64_t per = (((nch + TG_W - 1) / TG_W) + 3) & ~int64_t(3), c0 = wave * per, c1 = c0 + per < nch ? c0 + per : nch;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t s = c0; s < c1; s += 16) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t c = s + 4 * u + kq;
            a[u] = make_float4(0.f, 0.f, 0.f, 0.f); b[u] = a[u];
            if (c < c1) { a[u] = ra[c]; b[u] = rb[c]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[u].x, (double)b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[u].y, (double)b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[u].z, (double)b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[u].w, (double)b[u].w, acc, 0, 0, 0);
        }
    }
    if (wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][lane][r] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {                      // D layout (fp64 16x16): row = kq + 4 r, column = fi; the partial sums in wave order
            double v = acc[r];
#pragma unroll
            for (int w2 = 0; w2 < TG_W - 1; ++w2) v += red[w2][lane][r];
            const int k = ti * 16 + kq + 4 * r, l = tj * 16 + fi;
            if (k < K && l < K) { GK[(int64_t)k * K + l] = v; GK[(int64_t)l * K + k] = v; }
        }
    }
}

// workgroup = (block, frame segment, pair of 16-trace groups of the block's list); 8 waves x 2 pixel tiles = the block's 256 pixels
// V0 = 1 (option win_i8_planes; the default since the end of round 6 for sums over at least 2048 frames, bg.hip win_planes3): the video's lowest digit plane is neither loaded nor multiplied -- 24-bit samples, 3/4 of the
// video's bytes, 11 of the 13 MFMAs.  W against the float64 oracle: 5e-7 .. 1.8e-6 of its largest weight over the whole GPU suite (four planes: 4e-8 .. 2e-7; the tests allow
// 2e-6, SURVEY.md 8(c) asks 1e-3), A and C unchanged (profiles/r06/gputest_win3.txt, parity_observed_win3.json)
template <int NTG, int V0>
__device__ __forceinline__ void win_body_i8(const uint4 *__restrict__ dig, int64_t T16, const double *__restrict__ vscale, const uint4 *__restrict__ tdig, const double *__restrict__ tscale,
                                            int blk, int l0, int nl, int grp, const int *__restrict__ lst_k, int64_t st0, int64_t st1, double *__restrict__ Ut) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, kq = lane >> 4;
    const uint4 *vp[2]; const uint4 *tp[NTG];
#pragma unroll
    for (int a = 0; a < 2; ++a) vp[a] = dig + ((int64_t)blk * T16 * 4) * BLKPX + (wave * 2 + a) * 16 + fi;
    int kk[NTG];
#pragma unroll
    for (int b = 0; b < NTG; ++b) {
        const int sl = (grp * 2 + b) * 16 + fi;
        kk[b] = lst_k[l0 + (sl < nl ? sl : 0)];            // (slots behind the list redo its first trace: never stored)
        tp[b] = tdig + (int64_t)kk[b] * T16 * 4;
    }
    int4v_t acc[2][NTG][5];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTG; ++b)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[a][b][c] = (int4v_t){0, 0, 0, 0};
    struct Frag { int4v_t v[2][4]; int4v_t t[NTG][4]; };
    auto load = [&](int64_t st) {
        Frag f;
        const int64_t s16 = 4 * (st < st1 ? st : st1 - 1) + kq;          // (the step behind the segment re-reads its last one: no branch in the loop, never used)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int p = V0; p < 4; ++p) { const uint4 u = ld_stream(vp[a] + (s16 * 4 + p) * BLKPX); f.v[a][p] = (int4v_t){(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
#pragma unroll
        for (int b = 0; b < NTG; ++b)
#pragma unroll
            for (int p = 0; p < 4; ++p) { const uint4 u = tp[b][s16 * 4 + p]; f.t[b][p] = (int4v_t){(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
        return f;
    };
    auto mm = [&](const Frag &f) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NTG; ++b) {
                const int4v_t *x = f.v[a], *y = f.t[b];
                int4v_t (&c)[5] = acc[a][b];                              // class = p + r - 2; consecutive MFMAs on different accumulators
                c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[3], y[0], c[1], 0, 0, 0);
                c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[3], y[1], c[2], 0, 0, 0);
                c[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[3], y[2], c[3], 0, 0, 0);
                c[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[3], y[3], c[4], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[2], y[0], c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[2], y[1], c[1], 0, 0, 0);
                c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[2], y[2], c[2], 0, 0, 0);
                c[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[2], y[3], c[3], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[1], y[1], c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[1], y[2], c[1], 0, 0, 0);
                c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[1], y[3], c[2], 0, 0, 0);
                if constexpr (V0 == 0) {
                    c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[0], y[2], c[0], 0, 0, 0);
                    c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[0], y[3], c[1], 0, 0, 0);
                }
            }
    };
    Frag f0 = load(st0), f1;                                              // two named fragment sets: the next step's 24 loads go out before this step's MFMAs, no copies
    for (int64_t st = st0; st < st1; st += 2) {
        f1 = load(st + 1);
        mm(f0);
        f0 = load(st + 2);
        if (st + 1 < st1) mm(f1);
    }
    // D layout (int32 16x16): row = 4 kq + r (pixel of the tile), column = fi (trace of the group)
#pragma unroll
    for (int b = 0; b < NTG; ++b) {
        const int slot = (grp * 2 + b) * 16 + fi;
        if (slot >= nl) continue;
        const double ts = tscale[kk[b]];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int px = (wave * 2 + a) * 16 + 4 * kq + r;
                const double v = 65536.0 * (double)acc[a][b][0][r] + 16777216.0 * (double)acc[a][b][1][r] + 4294967296.0 * (double)acc[a][b][2][r] +
                                 1099511627776.0 * (double)acc[a][b][3][r] + 281474976710656.0 * (double)acc[a][b][4][r];
                Ut[(int64_t)(l0 + slot) * BLKPX + px] = v * vscale[(int64_t)blk * BLKPX + px] * ts;
            }
    }
}

// items[] = (block | group << 24) of the (block, trace-group pair) work items, longest lists first; blockIdx.x = item * nseg + segment
template <int V0>
__global__ void __launch_bounds__(512, 2) k_win_proj_i8(const uint4 *__restrict__ dig, int64_t T16, const double *__restrict__ vscale, const uint4 *__restrict__ tdig,
                                                        const double *__restrict__ tscale, const int *__restrict__ lst_ptr, const int *__restrict__ lst_k, const int *__restrict__ items,
                                                        int nseg, double *__restrict__ Ut, int64_t ut_stride) {
    const int it = items[blockIdx.x / nseg], seg = blockIdx.x % nseg;
    const int blk = it & 0xffffff, grp = it >> 24;
    const int l0 = lst_ptr[blk], nl = lst_ptr[blk + 1] - l0;
    const int64_t nstep = T16 >> 2, sseg = (nstep + nseg - 1) / nseg;
    const int64_t st0 = seg * sseg, st1 = st0 + sseg < nstep ? st0 + sseg : nstep;
    double *ut = Ut + seg * ut_stride;
    if (st0 >= st1) {                                        // (more segments than steps: this one contributes zeros)
        for (int b = 0; b < 2; ++b) {
            const int slot = (grp * 2 + b) * 16 + (int)(threadIdx.x & 15);
            if (slot < nl) for (int px = (int)(threadIdx.x >> 4); px < BLKPX; px += 32) ut[(int64_t)(l0 + slot) * BLKPX + px] = 0.0;
        }
        return;
    }
    if (nl - grp * 32 > 16) win_body_i8<2, V0>(dig, T16, vscale, tdig, tscale, blk, l0, nl, grp, lst_k, st0, st1, ut);
    else win_body_i8<1, V0>(dig, T16, vscale, tdig, tscale, blk, l0, nl, grp, lst_k, st0, st1, ut);
}

}  // namespace cnmfe
