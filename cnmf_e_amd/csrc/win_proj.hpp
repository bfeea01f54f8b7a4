// The window projection of the ring regression, shared by bg.hip (U~ of the incremental covariance table) and vproj.hip (the P = Yc Cc' table the
// sweep-free spatial update reads): per 16x16 pixel block and list of traces, U(i,k) = sum_t Yc_i(t) Cc_k(t) on the fp64 matrix pipe.
#pragma once
#include "common.hpp"
#include "ring_solve_core.hpp"

namespace cnmfe {

// local pixel index inside a 16x16 block: 4x4-pixel patches (patch = (r>>2) + 4*(c>>2)), 16 pixels per patch.
// One MFMA 16-row/col fragment is then one 4x4 patch, so the set of pixel displacements a 16x16 sub-tile of a
// block-pair covariance covers is a 7x7 window and sub-tiles no ring can ever touch are skipped.
__host__ __device__ __forceinline__ int lp_of(int r, int c) { return (((r >> 2) + ((c >> 2) << 2)) << 4) + (r & 3) + ((c & 3) << 2); }

struct BgGeom {
    int nr, nc, nr_b, nc_b, roff, coff;    // patch / block sizes, patch origin in block
    int r0_abs, c0_abs;                    // absolute 1-based row/col of block pixel (0,0)
    int d1, d2;
    int nbr, nbc;                          // 16x16 blocks tiling the block region
    int64_t d, d_b, T, Tp, Tpad;           // Tp frames used (stride kstride), padded to a multiple of 16
    int kstride;
    int p;
    int p_radius, nbw;                     // largest |offset| of the ring; blocks per side of the (2*radius+1)-pixel window
    int bf4;                               // Bf layout: 0 = [blk][frame][256], 1 = [blk][frame/4][256][4] (k_gram4)
};

constexpr int WIN_NLB = 64;
#ifndef WIN_AHEAD_N
#define WIN_AHEAD_N 2
#endif
constexpr int WIN_AHEAD = WIN_AHEAD_N;

// The same GEMMs with 16-BYTE loads (round 3; frame strides 1, 2, 4 -- every first fit, and every fit at the headline size).  Probes showed the
// scalar-load version above bound by its loads, not by the matrix pipe: 5.0 ms with the MFMAs removed against 5.4 ms with them
// (profiles/r03/win_probe.txt) -- 256 bytes per wave instruction.  Here lane (fi, kq) loads the float4 of ITS pixel for chunk s + kq (and the float4
// of its trace for the same four frames): a wave instruction moves 1 KB, and the four components are the k = kq slices of FOUR MFMAs -- MFMA m
// contracts over the frames 4 (s + kq) + m, kq = 0..3, on both operands alike, so no value ever changes lanes.  16 frames per step.
template <int NT>
__device__ __forceinline__ void win_body4(const float4 *__restrict__ Y4, const BgGeom &g, const float *__restrict__ Cc, int64_t ldc, int blk, int l0, int nl,
                                          const int *__restrict__ lst_k, int64_t c0, int64_t c1, double *__restrict__ Ut, double *__restrict__ Gb) {
    constexpr int AHEAD = NT <= 2 ? 2 : 1;
    const int bi = blk % g.nbr, bj = blk / g.nbr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, kq = lane >> 4;
    const float4 *ya[4]; const float4 *tb[NT];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int lp = (wave * 4 + a) * 16 + fi;
        const int lr = ((lp >> 4) & 3) * 4 + (lp & 3), lc = (lp >> 6) * 4 + ((lp >> 2) & 3);
        const int rb = bi * BLK + lr, cb = bj * BLK + lc;
        ya[a] = (rb < g.nr_b && cb < g.nc_b) ? Y4 + ((int64_t)cb * g.nr_b + rb) : nullptr;
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) { const int sl = b * 16 + fi; tb[b] = sl < nl ? reinterpret_cast<const float4 *>(Cc + (int64_t)lst_k[l0 + sl] * ldc) : nullptr; }
    double4_t acc[4][NT], accg[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        accg[b] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    struct Frag { float4 y[4]; float4 t[NT]; };
    auto load = [&](int64_t s) {
        Frag f;
        const int64_t c = s + kq;
        const bool on = c < c1;
        // (`cond ? *p : z4` on two lvalues becomes a select of ADDRESSES: z4 then lives in scratch memory and the kernel's first launch makes the runtime
        //  set its scratch arena up -- an intermittent 0.5-0.8 s stall of the first fit, profiles/r03/README.md)
#pragma unroll
        for (int a = 0; a < 4; ++a) { f.y[a] = make_float4(0.f, 0.f, 0.f, 0.f); if (on && ya[a]) f.y[a] = ya[a][c * g.d_b]; }
#pragma unroll
        for (int b = 0; b < NT; ++b) { f.t[b] = make_float4(0.f, 0.f, 0.f, 0.f); if (on && tb[b]) f.t[b] = tb[b][c]; }
        return f;
    };
    const bool gw = wave < NT && Gb != nullptr;             // wave w also owns row-group w of G (Gb == nullptr: the caller only wants U)
    const int ks = g.kstride;
    auto comp = [](float4 v, int m) -> float { return m == 0 ? v.x : m == 1 ? v.y : m == 2 ? v.z : v.w; };
    auto mm = [&](const Frag &f) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m & (ks - 1)) continue;                     // frame stride 2: components 0, 2; stride 4: component 0 (fit_ring_model.m:84-87)
            double bv[NT];
#pragma unroll
            for (int b = 0; b < NT; ++b) bv[b] = (double)comp(f.t[b], m);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const double av = (double)comp(f.y[a], m);
#pragma unroll
#ifdef CNMFE_PROBE_NOMFMA                                       // timing probe (scripts/build_variant.py): the loads stay, the matrix work goes -- results are garbage
                for (int b = 0; b < NT; ++b) acc[a][b][0] = fma(av, bv[b], acc[a][b][0]);
#else
                for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[b], acc[a][b], 0, 0, 0);
#endif
            }
            if (gw) {
                double gv = bv[0];
#pragma unroll
                for (int b = 1; b < NT; ++b) gv = wave == b ? bv[b] : gv;
#pragma unroll
                for (int b = 0; b < NT; ++b) accg[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(gv, bv[b], accg[b], 0, 0, 0);
            }
        }
    };
    if constexpr (NT >= 3) {
        // 33..64 traces: 160 accumulator registers leave no room for a second fragment set -- no software prefetch (the other wave of the SIMD covers the
        // latency; with it the body spilled to scratch memory)
        for (int64_t s = c0; s < c1; s += 4) { const Frag f = load(s); mm(f); }
    } else {
        Frag f[AHEAD];
#pragma unroll
        for (int d = 0; d < AHEAD; ++d) f[d] = load(c0 + 4 * d);
        for (int64_t s = c0; s < c1; s += 4 * AHEAD) {
#pragma unroll
            for (int d = 0; d < AHEAD; ++d) {
                const Frag nx = load(s + 4 * (AHEAD + d));
                mm(f[d]);
                f[d] = nx;
            }
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

// frame strides 1, 2, 4: 16-byte loads over the video's chunks (the used frames are components of them).  BIG = the blocks with 49..64 traces, a kernel of
// their own: together with the other bodies the 4-group body did not fit 256 registers, and a kernel that spills needs scratch memory, whose arena the
// runtime sets up at the kernel's FIRST launch -- measured as an intermittent 0.5-0.8 s inside the first fit of a process (profiles/r03/README.md)
template <bool BIG>
__global__ void __launch_bounds__(256, (BIG ? 1 : 2)) k_win_proj4(const float4 *__restrict__ Y4, BgGeom g, const float *__restrict__ Cc, int64_t ldc, const int *__restrict__ lst_ptr,
                                                      const int *__restrict__ lst_k, const int *__restrict__ blk_list, int nseg, double *__restrict__ Ut, int64_t ut_stride,
                                                      double *__restrict__ Gb, int64_t gb_stride) {
    const int blk = blk_list[blockIdx.x / nseg], seg = blockIdx.x % nseg;
    const int l0 = lst_ptr[blk], nl = lst_ptr[blk + 1] - l0;
    double *ut = Ut + seg * ut_stride, *gb = Gb ? Gb + seg * gb_stride : nullptr;
    const int64_t nchunk = (g.T + 3) >> 2;
    const int64_t cseg = ((nchunk + nseg - 1) / nseg + 7) & ~int64_t(7);
    const int64_t c0 = seg * cseg, c1 = c0 + cseg < nchunk ? c0 + cseg : nchunk;
    if constexpr (BIG) win_body4<4>(Y4, g, Cc, ldc, blk, l0, nl, lst_k, c0, c1, ut, gb);
    else switch ((nl + 15) >> 4) {
        case 1: win_body4<1>(Y4, g, Cc, ldc, blk, l0, nl, lst_k, c0, c1, ut, gb); break;
        case 2: win_body4<2>(Y4, g, Cc, ldc, blk, l0, nl, lst_k, c0, c1, ut, gb); break;
        case 3: win_body4<3>(Y4, g, Cc, ldc, blk, l0, nl, lst_k, c0, c1, ut, gb); break;
        default: break;
    }
}


}  // namespace cnmfe
