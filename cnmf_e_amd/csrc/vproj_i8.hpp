// T1', round 5 (late): the temporal projection  part(b, k)(t) = sum_{j in block b} B(j, k) Yc(j, t)  on the int8 matrix pipe, out of the video's resident digit planes.
//
// k_vp_proj_b (vproj.hip) contracts the block's 256 pixels on the fp64 matrix pipe: 16 MFMAs of 64 clocks per 64 pixels x 16 frames x 16 list slots -- 1.07 ms of matrix
// time at the headline size under a 10.5 GB read, and the two do not overlap perfectly (2.0-2.2 ms = 0.59-0.66 of HBM).  The fit's window projection had the same
// problem and left it on the int8 pipe (win_proj_i8.hpp); here the contraction runs over PIXELS, so
//   * the video's digit planes are needed pixel-major: digp[(((blk * T16 + fg) * 4 + pg) * 4 + plane) * 64 + lane], lane = (frame f = lane & 15, kg = lane >> 4), 16 bytes =
//     digit `plane` of the pixels lp = 64 pg + 16 kg + j, j = 0 .. 15, at frame 16 fg + f (lp: the 4 x 4-patch pixel order of the planes) -- one B operand fragment of
//     v_mfma_i32_16x16x64_i8 (K = 64 pixels, N = 16 frames).  k_dig_pixmajor transposes P->dig once per upload (16 x 16 byte tiles through LDS); it replaces the
//     read-order fp32 copy (same size);
//   * the pixel scales s_j of the planes sit INSIDE the contraction, so they are folded into the panel: B'(j, k) = B(j, k) s_j, quantised per (block, slot) column to
//     32-bit fixed point (scale t) and cut into four balanced base-256 digits in A-operand order (k_vp_bdig: rows = 16 list slots, K = 64 pixels);
//   * 13 MFMAs (digit-pair classes p + r = 2 .. 6, exact int32 sums: 256 pixels x 4 pairs x 2^14 fits easily) per 64 pixels x 16 frames x 16 slots, and
//     part = t sum_c 256^(c + 2) acc_c in fp64.  The only error is the 32-bit quantisation of the two operands (1e-9 of a column's largest term).
#pragma once

namespace cnmfe {

typedef int int4v_t __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ int lp_inv_px(int lp) {       // local pixel (4 x 4-patch order) -> r + 16 c inside the block
    const int r = ((lp >> 4) & 3) * 4 + (lp & 3), c = (lp >> 6) * 4 + ((lp >> 2) & 3);
    return r + 16 * c;
}

// one workgroup per (block, 16-frame group): the four planes' 256 x 16-byte fragments transposed from frame-major (16 frames of a pixel) to pixel-major (16 pixels of a frame)
__global__ void __launch_bounds__(256) k_dig_pixmajor(const uint4 *__restrict__ dig, int64_t T16, uint4 *__restrict__ digp) {
    __shared__ unsigned char tile[4][256][16 + 4];                  // [plane][pixel][frame] (+4: the gathers below walk pixels at a fixed frame)
    const int64_t blk = blockIdx.x, fg = blockIdx.y;
    const int lp = threadIdx.x;
    const uint4 *src = dig + ((blk * T16 + fg) * 4) * BLKPX + lp;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint4 v = src[p * BLKPX];
        unsigned *o = reinterpret_cast<unsigned *>(&tile[p][lp][0]);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    __syncthreads();
    const int pg = threadIdx.x >> 6, lane = threadIdx.x & 63, f = lane & 15, kg = lane >> 4;
    uint4 *dst = digp + (((blk * T16 + fg) * 4 + pg) * 4) * 64 + lane;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j >> 2] |= (unsigned)tile[p][pg * 64 + kg * 16 + j][f] << (8 * (j & 3));
        dst[p * 64] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// B' digits of one (block, slot group): 256 threads = the block's pixels in lp order.  bdig[(((grp * 4 + pg) * 4 + plane) * 64 + lane)], lane = (slot s = lane & 15, kg):
// 16 bytes = digit `plane` of B'(lp = 64 pg + 16 kg + j, s); bscale[grp * 16 + s] = the column's scale (1 for an all-zero column: its digits are 0)
__global__ void __launch_bounds__(256) k_vp_bdig(const double *__restrict__ Bt, const int *__restrict__ grp_blk, const double *__restrict__ dig_sc, uint4 *__restrict__ bdig,
                                                 double *__restrict__ bscale) {
    __shared__ double red[4][16];
    __shared__ double tsc[16];
    __shared__ unsigned char q8[4][16][256];                        // [plane][slot][lp]
    const int64_t grp = blockIdx.x;
    const int blk = grp_blk[grp], lp = threadIdx.x, px = lp_inv_px(lp);
    const double s_px = dig_sc[(int64_t)blk * BLKPX + lp];
    double v[16];
    const double *src = Bt + (grp * BLKPX + px) * 16;
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = src[s] * s_px;
    const int lane = lp & 63, wave = lp >> 6;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        double m = fabs(v[s]);
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        if (lane == 0) red[wave][s] = m;
    }
    __syncthreads();
    if (lp < 16) {
        const double m = fmax(fmax(red[0][lp], red[1][lp]), fmax(red[2][lp], red[3][lp]));
        const double t = m > 0.0 ? m / 2130706432.0 : 1.0;           // 2^31 - 2^24
        tsc[lp] = t; bscale[grp * 16 + lp] = t;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        int qv = __double2int_rn(v[s] / tsc[s]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int d = (int)(signed char)(qv & 0xff);
            q8[p][s][lp] = (unsigned char)(d & 0xff);
            qv = (qv - d) >> 8;
        }
    }
    __syncthreads();
    // 4 pixel groups x 4 planes x 64 lanes fragments: four per thread
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int ln = i & 63, p = (i >> 6) & 3, pg = i >> 8, s = ln & 15, kg = ln >> 4;
        const unsigned *w = reinterpret_cast<const unsigned *>(&q8[p][s][pg * 64 + kg * 16]);
        bdig[((grp * 4 + pg) * 4 + p) * 64 + ln] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// The projection.  Workgroup = (block, frame segment), 4 waves; a wave takes 16-frame groups.  Per group: 16 fragment loads of the video (4 pixel groups x 4 planes), the
// panel's fragments from LDS (one ds_read_b128 each), 52 MFMAs per slot group into 5 class accumulators, and the D tile (row = list slot (lane >> 4) * 4 + r, column =
// frame lane & 15) scaled into the partial rows part[(l0 + slot) * ldp + 16 fg + f] (a wave instruction writes 16 consecutive frames of four slots)
// V0 = 1 (round 6, option proj_i8_planes = 3, the default): the video's lowest digit plane is neither loaded nor multiplied -- 24-bit samples (the rounding of a
// pixel's centred trace to 2^-23 of its largest value, the precision its fp32 samples have around a mean of that size anyway), 3/4 of the bytes, 11 of the 13 MFMAs.
// The regression's window projection (win_proj_i8.hpp) reads three planes too since the end of round 6 (win_i8_planes): W stays within the tests' 2e-6 of the oracle.
template <int NT, int V0>
__global__ void __launch_bounds__(256) k_vp_proj_i8(const uint4 *__restrict__ digp, int64_t T16, const int *__restrict__ blk_list, const int *__restrict__ lst_ptr,
                                                    const int *__restrict__ g16, const uint4 *__restrict__ bdig, const double *__restrict__ bscale, int nseg,
                                                    double *__restrict__ part, int64_t ldp) {
    extern __shared__ __attribute__((aligned(16))) uint4 pan[];   // [NT][4 pg][4 planes][64 lanes]
    const int blk = blk_list[blockIdx.x / nseg], seg = blockIdx.x % nseg;
    {
        const uint4 *src = bdig + (int64_t)g16[blk] * 1024;
        for (int i = threadIdx.x; i < NT * 1024; i += 256) pan[i] = src[i];
    }
    __syncthreads();
    const int l0 = lst_ptr[blk], nl = lst_ptr[blk + 1] - l0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, f = lane & 15, kq = lane >> 4;
    double ts[NT][4];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) ts[b][r] = bscale[((int64_t)g16[blk] + b) * 16 + kq * 4 + r];
    const int64_t gseg = (T16 + nseg - 1) / nseg, fg0 = seg * gseg, fg1 = fg0 + gseg < T16 ? fg0 + gseg : T16;
    const uint4 *vb = digp + ((int64_t)blk * T16 * 16) * 64 + lane;
    auto load = [&](int64_t fg, int4v_t (&x)[4][4 - V0]) {
        const uint4 *p = vb + (fg < fg1 ? fg : fg1 - 1) * 16 * 64;  // (the group behind the segment re-reads its last one: no branch, never used)
#pragma unroll
        for (int pg = 0; pg < 4; ++pg)
#pragma unroll
            for (int pl = V0; pl < 4; ++pl) { const uint4 u = ld_stream(p + (pg * 4 + pl) * 64); x[pg][pl - V0] = (int4v_t){(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
    };
    auto compute = [&](int64_t fg, const int4v_t (&x)[4][4 - V0]) {
        const int64_t t = fg * 16 + f;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            // (one slot group at a time: 5 accumulators live.  With more than one group the panel's fragments are re-read from LDS per frame group -- hoisted out of
            //  the frame loop they cost 64 registers per group and the kernel its second wave per SIMD)
            if (NT > 1) asm volatile("" ::: "memory");
            int4v_t c[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) c[q] = (int4v_t){0, 0, 0, 0};
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                int4v_t y[4];                                       // the panel's four planes of this pixel group (A operand: rows = slots)
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) { const uint4 u = pan[((b * 4 + pg) * 4 + pl) * 64 + lane]; y[pl] = (int4v_t){(int)u.x, (int)u.y, (int)u.z, (int)u.w}; }
                // class = (plane of B') + (plane of the video) - 2; consecutive MFMAs on different accumulators
                if constexpr (V0 == 0) c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[3], x[pg][0], c[1], 0, 0, 0);
                c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[3], x[pg][1 - V0], c[2], 0, 0, 0);
                c[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[3], x[pg][2 - V0], c[3], 0, 0, 0);
                c[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[3], x[pg][3 - V0], c[4], 0, 0, 0);
                if constexpr (V0 == 0) c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[2], x[pg][0], c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[2], x[pg][1 - V0], c[1], 0, 0, 0);
                c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[2], x[pg][2 - V0], c[2], 0, 0, 0);
                c[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[2], x[pg][3 - V0], c[3], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[1], x[pg][1 - V0], c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[1], x[pg][2 - V0], c[1], 0, 0, 0);
                c[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[1], x[pg][3 - V0], c[2], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[0], x[pg][2 - V0], c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(y[0], x[pg][3 - V0], c[1], 0, 0, 0);
            }
            if (t < ldp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int slot = b * 16 + kq * 4 + r;
                    if (slot < nl) {
                        const double v = 65536.0 * (double)c[0][r] + 16777216.0 * (double)c[1][r] + 4294967296.0 * (double)c[2][r] +
                                         1099511627776.0 * (double)c[3][r] + 281474976710656.0 * (double)c[4][r];
                        part[(int64_t)(l0 + slot) * ldp + t] = v * ts[b][r];
                    }
                }
            }
        }
    };
    int4v_t x0[4][4 - V0], x1[4][4 - V0];                           // two named fragment sets: the next group's 16 loads go out before this group's MFMAs
    if (fg0 + wave >= fg1) return;
    load(fg0 + wave, x0);
    for (int64_t fg = fg0 + wave; fg < fg1; fg += 8) {
        load(fg + 4, x1);
        compute(fg, x0);
        load(fg + 8, x0);
        if (fg + 4 < fg1) compute(fg + 4, x1);
    }
}

}  // namespace cnmfe
