// B2a, round 5: the covariance table of the VIDEO (and the direct Gram of Bf, the fallback) on the int8 matrix pipe -- exactly.
//
// The table  cov(a, b) = sum_t Yc_a(t) Yc_b(t)  is built once per recording (and frame stride) and was the cold path's largest item: 172 ms of fp64 MFMAs at
// 512 x 512 x 10000 (k_gram4<0>, 0.7 of the fp64 matrix peak).  Its consumers need ~1e-10 relative accuracy (the ridge systems have condition numbers of 1e4-1e5:
// the fp32 / split-bf16 modes of k_gram4 cost 2e-4 / 9e-5 of W), which rules out floating-point accumulation in less than fp64 -- but not INTEGER accumulation:
//   * every pixel's centred trace is scaled to 32-bit fixed point, q_a(t) = rint(Yc_a(t) / s_a), s_a = max_t |Yc_a| / (2^31 - 2^24) (k_dig_scale), and cut into
//     four balanced base-256 digits, q = d0 + 256 d1 + 256^2 d2 + 256^3 d3, d in [-128, 127] (k_build_dig: the same 4 bytes per sample as the fp32 Bf);
//   * sum_t q_a q_b = sum_{p, r} 256^(p + r) sum_t d_p^a d_r^b, and every digit-pair sum is an int8 GEMM with EXACT int32 accumulation
//     (v_mfma_i32_16x16x64_i8: 64 frames per instruction, twice the bf16 rate, 32x the fp64 rate).  Pairs of equal weight share an accumulator; the classes
//     p + r = 0, 1 are dropped (relative weight 2^-40: nothing), p + r = 2 .. 6 are kept -- 13 MFMAs per 16 x 16 tile and 64 frames into 5 accumulators
//     (dropping p + r = 2 as well costs 4.7e-7 of W: measured on the CPU emulation, scripts/probes/gram_i8_emulation.py);
//   * cov(a, b) = s_a s_b sum_c 256^c acc_c in fp64 at the end.
// The only error is the 32-bit quantisation of the data: W differs from the fp64 table's by 1.3e-8 (the fp64 table itself is 2e-7 from the float64 oracle through
// the fp32 storage of W).  int32 range: 4 pairs x 64 frames x 2^14 per step -> recordings up to 24576 used frames; longer ones take the fp64 kernel.
// Data path = the split-bf16 mode's (LDS-DMA of 16-frame stages, one ds_read_b128 per operand fragment), with steps of FOUR stages (K = 64), two steps
// resident (128 KB of LDS, one 8-wave workgroup per CU).  Work items, tile lists and the table layout are k_gram4's.
#pragma once

namespace cnmfe {

typedef int int4v_t __attribute__((ext_vector_type(4)));

// 16 consecutive used frames tp .. tp + 15 of Bf = Yc - A (C - mean C) for block-region pixel q (arow == nullptr: the centred video itself); zero outside the block
// region and behind the last used frame.  Same fp32 arithmetic as k_build_bf (bg.hip).
struct DigA { const int *arow, *acol; const float *aval, *Cc; int64_t ldc; };
__device__ __forceinline__ void dig_frames16(const float4 *__restrict__ Y4, int64_t Tc, const BgGeom &g, bool in, int64_t q, int64_t tp, int e0, int e1, const DigA &A, float (&x)[16]) {
    if (g.kstride == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = (tp >> 2) + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in && c < Tc) {
                v = Y4[c * g.d_b + q];
                for (int e = e0; e < e1; ++e) {
                    const float av = A.aval[e];
                    const float4 c4 = *reinterpret_cast<const float4 *>(A.Cc + (int64_t)A.acol[e] * A.ldc + 4 * c);
                    v.x -= av * c4.x; v.y -= av * c4.y; v.z -= av * c4.z; v.w -= av * c4.w;
                }
            }
            const int64_t t0 = tp + 4 * j;                         // (frames behind the last used one count as 0 whatever the chunk's padding holds)
            x[4 * j] = t0 < g.Tp ? v.x : 0.f; x[4 * j + 1] = t0 + 1 < g.Tp ? v.y : 0.f; x[4 * j + 2] = t0 + 2 < g.Tp ? v.z : 0.f; x[4 * j + 3] = t0 + 3 < g.Tp ? v.w : 0.f;
        }
    } else {
        const float *Ys = reinterpret_cast<const float *>(Y4);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = 0.f;
            if (in && tp + j < g.Tp) {
                const int64_t t = (tp + j) * g.kstride;            // frame subsampling Bf(:, 1:k:end)  (fit_ring_model.m:87)
                v = Ys[((t >> 2) * g.d_b + q) * 4 + (t & 3)];
                for (int e = e0; e < e1; ++e) v -= A.aval[e] * A.Cc[(int64_t)A.acol[e] * A.ldc + t];
            }
            x[j] = v;
        }
    }
}

// smax[blk * 256 + lp] = max over the used frames of |Bf| (as the bits of a non-negative float: atomicMax over the frame chunks of the grid; zeroed by the caller)
__global__ void __launch_bounds__(256) k_dig_scale(const float4 *__restrict__ Y4, int64_t Tc, BgGeom g, DigA A, int tchunk, unsigned *__restrict__ smax) {
    const int blk = blockIdx.x, bi = blk % g.nbr, bj = blk / g.nbr, lp = threadIdx.x;
    const int lr = ((lp >> 4) & 3) * 4 + (lp & 3), lc = (lp >> 6) * 4 + ((lp >> 2) & 3);
    const int rb = bi * BLK + lr, cb = bj * BLK + lc;
    const bool in = rb < g.nr_b && cb < g.nc_b;
    if (!in) return;
    const int64_t q = (int64_t)cb * g.nr_b + rb;
    int e0 = 0, e1 = 0;
    if (A.arow) { e0 = A.arow[q]; e1 = A.arow[q + 1]; }
    const int64_t tp0 = (int64_t)blockIdx.y * tchunk, tp1 = tp0 + tchunk < g.Tpad ? tp0 + tchunk : g.Tpad;
    float m = 0.f;
    for (int64_t tp = tp0; tp < tp1; tp += 16) {
        float x[16];
        dig_frames16(Y4, Tc, g, in, q, tp, e0, e1, A, x);
#pragma unroll
        for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(x[j]));
    }
    atomicMax(&smax[(int64_t)blk * BLKPX + lp], __float_as_uint(m));
}

// digit planes dig[((blk * T16 + s) * 4 + plane) * 256 + lp] (16 bytes = the digits of frames 16 s .. 16 s + 15, zero behind the used frames); scale = the pixel's
// smax / (2^31 - 2^24) (1 for an all-zero or out-of-region pixel), written by the first frame chunk; rs[chunk][blk][lp] = this frame chunk's row sums of the fp32
// values (the ones row of the regression: not quantised)
__global__ void __launch_bounds__(256) k_build_dig(const float4 *__restrict__ Y4, int64_t Tc, BgGeom g, DigA A, const unsigned *__restrict__ smax, double *__restrict__ scale,
                                                   uint4 *__restrict__ dig, int tchunk, double *__restrict__ rs) {
    const int blk = blockIdx.x, bi = blk % g.nbr, bj = blk / g.nbr, lp = threadIdx.x;
    const int lr = ((lp >> 4) & 3) * 4 + (lp & 3), lc = (lp >> 6) * 4 + ((lp >> 2) & 3);
    const int rb = bi * BLK + lr, cb = bj * BLK + lc;
    const bool in = rb < g.nr_b && cb < g.nc_b;
    const int64_t q = in ? (int64_t)cb * g.nr_b + rb : 0;
    int e0 = 0, e1 = 0;
    if (in && A.arow) { e0 = A.arow[q]; e1 = A.arow[q + 1]; }
    const float mx = __uint_as_float(smax[(int64_t)blk * BLKPX + lp]);
    const double sc = mx > 0.f ? (double)mx / 2130706432.0 : 1.0;          // 2^31 - 2^24
    if (blockIdx.y == 0) scale[(int64_t)blk * BLKPX + lp] = sc;
    const double inv = 1.0 / sc;
    const int64_t tp0 = (int64_t)blockIdx.y * tchunk;          // tchunk is a multiple of 16
    const int64_t tp1 = tp0 + tchunk < g.Tpad ? tp0 + tchunk : g.Tpad;
    const int64_t T16 = g.Tpad >> 4;
    double rsum = 0.0;
    for (int64_t tp = tp0; tp < tp1; tp += 16) {
        float x[16];
        dig_frames16(Y4, Tc, g, in, q, tp, e0, e1, A, x);
        unsigned pl[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) pl[p][k] = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            rsum += (double)x[j];
            int qv = __double2int_rn((double)x[j] * inv);                   // |qv| <= 2^31 - 2^24
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int d = (int)(signed char)(qv & 0xff);               // balanced digit in [-128, 127]
                pl[p][j >> 2] |= (unsigned)(d & 0xff) << (8 * (j & 3));
                qv = (qv - d) >> 8;
            }
        }
        uint4 *o = dig + (((int64_t)blk * T16 + (tp >> 4)) * 4) * BLKPX + lp;
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p * BLKPX] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    }
    if (rs) rs[((int64_t)blockIdx.y * gridDim.x + blk) * BLKPX + lp] = rsum;       // per frame chunk: k_rs_reduce adds the chunks in a fixed order (an fp64 atomicAdd here made
}                                                                                   // the row sums -- and through them W -- differ in the last bits from run to run)

__global__ void __launch_bounds__(256) k_rs_reduce(const double *__restrict__ part, int nchunk, int64_t n, double *__restrict__ rs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[(int64_t)c * n + i];
    rs[i] = s;
}

constexpr int GI_STAGE_B = 16384;            // a 16-frame stage in LDS: [A half: plane(4) x 128 px x 16 B][B half: the same]
constexpr int GI_NBUF = 8;                   // two steps of four stages: 128 KB

struct GiWave {
    const uint4 *gA, *gB;                    // digit planes of the two blocks (pixel-half offset included)
    unsigned vo, ldsA, ldsB;                 // this wave's DMA piece: byte offset inside a stage of a block / LDS destinations inside a stage buffer
    int ns, lane, nstep;
    int64_t stage_stride;                    // uint4 per 16-frame stage of a block = 4 * 256
    double *out; const double *sA, *sB;
};

template <int NS>
__device__ __forceinline__ void gram_i8_run(const GiWave &w, const char *smem, unsigned smem_base, const int *__restrict__ tlw) {
    int ti[NS], tj[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int code = __builtin_amdgcn_readfirstlane(tlw[sl < w.ns ? sl : 0]);     // a contiguous run of the (row-major) tile list; slots past ns redo tile 0, never stored
        ti[sl] = code & 7; tj[sl] = code >> 4;
    }
    int4v_t acc[NS][5];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[sl][c] = (int4v_t){0, 0, 0, 0};
    auto issue = [&](int step) {             // the four stages of a step -> buffers (step & 1) * 4 ..
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int64_t st = (int64_t)step * 4 + h;
            const unsigned d = smem_base + (unsigned)(((step & 1) * 4 + h) * GI_STAGE_B);
            glds16(w.gA + st * w.stage_stride, w.vo, d + w.ldsA);
            glds16(w.gB + st * w.stage_stride, w.vo, d + w.ldsB);
        }
    };
    issue(0);
    const int g4 = w.lane >> 4, fl = w.lane & 15;
    for (int step = 0; step < w.nstep; ++step) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of the step have landed
        __builtin_amdgcn_s_barrier();                               // ... everybody's; and everybody is done reading the other buffer set
        asm volatile("" ::: "memory");
        if (step + 1 < w.nstep) issue(step + 1);
        const char *sb = smem + ((step & 1) * 4 + g4) * GI_STAGE_B;  // lane group g4 contracts the 16 frames of stage g4
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            int4v_t a[4], b[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                a[p] = *reinterpret_cast<const int4v_t *>(sb + p * 2048 + (ti[sl] * 16 + fl) * 16);
                b[p] = *reinterpret_cast<const int4v_t *>(sb + 8192 + p * 2048 + (tj[sl] * 16 + fl) * 16);
            }
            // class c = p + r - 2; consecutive MFMAs go to different accumulators
            acc[sl][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[3], b[0], acc[sl][1], 0, 0, 0);
            acc[sl][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[3], b[1], acc[sl][2], 0, 0, 0);
            acc[sl][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[3], b[2], acc[sl][3], 0, 0, 0);
            acc[sl][4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[3], b[3], acc[sl][4], 0, 0, 0);
            acc[sl][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], b[0], acc[sl][0], 0, 0, 0);
            acc[sl][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], b[1], acc[sl][1], 0, 0, 0);
            acc[sl][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], b[2], acc[sl][2], 0, 0, 0);
            acc[sl][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], b[3], acc[sl][3], 0, 0, 0);
            acc[sl][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], b[1], acc[sl][0], 0, 0, 0);
            acc[sl][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], b[2], acc[sl][1], 0, 0, 0);
            acc[sl][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], b[3], acc[sl][2], 0, 0, 0);
            acc[sl][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], b[2], acc[sl][0], 0, 0, 0);
            acc[sl][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], b[3], acc[sl][1], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
        if (sl < w.ns) {
            const double sb_ = w.sB[tj[sl] * 16 + fl];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = (w.lane >> 4) * 4 + r;               // D layout of the 16x16 int32 result: row = (lane >> 4) * 4 + r, column = lane & 15
                const double v = 65536.0 * (double)acc[sl][0][r] + 16777216.0 * (double)acc[sl][1][r] + 4294967296.0 * (double)acc[sl][2][r] +
                                 1099511627776.0 * (double)acc[sl][3][r] + 281474976710656.0 * (double)acc[sl][4][r];
                w.out[(int64_t)(ti[sl] * 16 + rr) * BLKPX + tj[sl] * 16 + fl] = v * w.sA[ti[sl] * 16 + rr] * sb_;
            }
        }
}

__global__ void __launch_bounds__(512, 2) k_gram_i8(const uint4 *__restrict__ dig, int64_t T16, const int4 *__restrict__ pairs, const int *__restrict__ work, int nwork,
                                                    const int *__restrict__ tl_cnt, const int *__restrict__ tl, const double *__restrict__ scale, double *__restrict__ cov) {
    extern __shared__ __attribute__((aligned(16))) char smem_i8[];
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
    GiWave w;
    w.stage_stride = 4 * BLKPX;
    w.gA = dig + (int64_t)pr.x * T16 * w.stage_stride + ih * 128;
    w.gB = dig + (int64_t)pr.y * T16 * w.stage_stride + jh * 128;
    // DMA: a stage of a half is plane(4) x 128 px x 16 B = 8 pieces of 1 KB; wave v moves piece v (plane v >> 1, pixels (v & 1) * 64 ..) of both halves
    w.vo = (unsigned)((((wave >> 1) * BLKPX) + (wave & 1) * 64 + lane) * 16);
    w.ldsA = (unsigned)((wave >> 1) * 2048 + (wave & 1) * 1024);
    w.ldsB = w.ldsA + 8192u;
    w.lane = lane; w.nstep = (int)(T16 >> 2);
    const int nsmax = (cnt + 7) >> 3;
    const int lo_ = wave * nsmax;
    w.ns = cnt > lo_ ? (cnt - lo_ < nsmax ? cnt - lo_ : nsmax) : 0;
    w.out = cov + (int64_t)pair * BLKPX * BLKPX + (int64_t)(ih * 128) * BLKPX + jh * 128;
    w.sA = scale + (int64_t)pr.x * BLKPX + ih * 128;
    w.sB = scale + (int64_t)pr.y * BLKPX + jh * 128;
    const int *tlw = tl + lidx * 64 + (w.ns ? lo_ : 0);
    const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem_i8;
    switch (nsmax) {
#define GI_CASE(N) case N: gram_i8_run<N>(w, smem_i8, sbase, tlw); break;
        GI_CASE(1) GI_CASE(2) GI_CASE(3) GI_CASE(4) GI_CASE(5) GI_CASE(6) GI_CASE(7) GI_CASE(8)
#undef GI_CASE
        default: break;
    }
}

}  // namespace cnmfe
