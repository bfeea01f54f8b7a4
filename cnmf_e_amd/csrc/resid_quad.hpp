// R1 "quad roles": the four arc roles of resid_arc.hpp inside ONE wave.   (included by resid.hip; gfx950 only)
//
// What bounds k_residual_arc_dma is not a pipe but its phases (option r1_probe, 512 x 512 x 10000: ring product alone 5.3 ms, LDS-DMA ingest
// alone 4.4 ms, partial-sum exchange 1.1 ms, stores 1.3 ms, loop skeleton 1.6 ms -- 11.5 ms together): one 512-thread workgroup per CU with two
// barriers per chunk cannot overlap them, and a second workgroup does not fit beside 91 KB of halo buffers + 33 KB of partial sums.
// Here a wave is self-contained:
//   * the ring is ONE canonical arc (the left one) and its three 90-degree rotations; every lane runs the SAME read / FMA program with its
//     own base address and strides (s_fix, s_mov) in the halo image, so the four roles sit side by side in a wave;
//   * a role occupies exactly one of the four 16-lane groups a ds_read_b128 is served in ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), with 16
//     distinct strips: every read is conflict-free for any offset (halo row stride == 1 mod 16);
//   * the partners of a strip (left/right arc, top/bottom arc) are lanes l and l ^ 7: one DPP row_half_mirror add combines them; the
//     vertical and horizontal sums of a wave's 16 x 4 centres meet in 2 KB of wave-private LDS -- no workgroup barrier;
//   * every lane finishes ONE centre (lane l <-> row l & 15, column l >> 4 of the wave's block) and the store is four 256-byte segments.
// A workgroup is 4 waves on a 16 x 16 tile with two halo buffers (2 x 36 KB) + 8 KB of exchange space = 80 KB: TWO workgroups per CU, one
// barrier per chunk each, so one workgroup's ingest wait, exchange and stores run under the other's ring product.
// Index algebra checked lane by lane in scripts/r1_quad_model.py.
//
// MEASURED (512 x 512 x 10000, scripts/r1_probe.py, scripts/pmc_r1_probe.sh): 13.3 ms against 11.5 ms for k_residual_arc_dma -- NOT the default
// (cnmfe_set_option("r1_variant", 12) selects it).  The 16 x 16 tiles stage 8.3 halo pixels per centre instead of 5.6, and what bounds the
// sweep turned out to be the fabric, not the phases: the XCD L2s only catch part of the halo overlap between tiles (TCC hit rate 56 % for the
// 16 x 32 tiles, 27 % here), so the arc kernel moves 41 GB (read) + 10.5 GB (write) = 6.2 TB/s -- the rate a plain copy reaches on this
// chip -- and this one 90 + 10.5 GB; with the ring product switched off they take 8.3 / 15.7 ms.  Two workgroups per CU do overlap: the
// compute-only time is the same 9.1 ms with twice the instruction overhead per centre.  DESIGN.md section 3, R1.
#pragma once

namespace cnmfe {

constexpr int QD_T = 16, QD_P = 4;

template <int R> struct QuadTab {
    int na;                   // offsets of the canonical (left) arc
    int fix[40], mov[40];     // canonical offsets sorted by (fix, mov): (dr, dc) = (mov, fix)
    int ring[4][40];          // ring index (W row) of canonical offset a under the role's rotation
    int nl;                   // LDS reads of the program
    int rfix[64], rmov[64];   // position of read li: fix, and the moving coordinate (first offset of the run + x)
    int nf[64], fj[64][4], fa[64][4];
};
template <int R> constexpr QuadTab<R> make_quad() {
    QuadTab<R> t{};
    constexpr RingTab<R> ring = make_ring<R>();
    int n = 0;
    for (int i = 0; i < ring.n; ++i) {
        const int dr = ring.dr[i], dc = ring.dc[i];
        const int adr = dr < 0 ? -dr : dr, adc = dc < 0 ? -dc : dc;
        if (dc < 0 && (adc > adr || (adc == adr && dr < 0))) { t.fix[n] = dc; t.mov[n] = dr; ++n; }
    }
    t.na = n;
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && (t.fix[j] < t.fix[j - 1] || (t.fix[j] == t.fix[j - 1] && t.mov[j] < t.mov[j - 1])); --j) {
            int x = t.fix[j]; t.fix[j] = t.fix[j - 1]; t.fix[j - 1] = x;
            x = t.mov[j]; t.mov[j] = t.mov[j - 1]; t.mov[j - 1] = x;
        }
    // rotations: role 0 (dr, dc) = (mov, fix); 1: (-mov, -fix); 2: (fix, -mov); 3: (-fix, mov)
    for (int role = 0; role < 4; ++role)
        for (int a = 0; a < n; ++a) {
            const int m = t.mov[a], f = t.fix[a];
            const int dr = role == 0 ? m : role == 1 ? -m : role == 2 ? f : -f;
            const int dc = role == 0 ? f : role == 1 ? -f : role == 2 ? -m : m;
            int idx = -1;
            for (int i = 0; i < ring.n; ++i) if (ring.dr[i] == dr && ring.dc[i] == dc) idx = i;
            t.ring[role][a] = idx;
        }
    // runs of consecutive moving offsets at one fixed offset; a run of length L costs L + P - 1 reads for P * L products
    int li = 0;
    for (int a0 = 0; a0 < n;) {
        int rl = 1;
        while (a0 + rl < n && t.fix[a0 + rl] == t.fix[a0] && t.mov[a0 + rl] == t.mov[a0] + rl) ++rl;
        for (int x = 0; x < rl + QD_P - 1; ++x) {
            t.rfix[li] = t.fix[a0]; t.rmov[li] = t.mov[a0] + x;
            int nf = 0;
            for (int j = 0; j < QD_P; ++j) { const int u = x - j; if (u >= 0 && u < rl) { t.fj[li][nf] = j; t.fa[li][nf] = a0 + u; ++nf; } }
            t.nf[li] = nf;
            ++li;
        }
        a0 += rl;
    }
    t.nl = li;
    return t;
}
template <int R> struct QuadConst { static constexpr QuadTab<R> tab = make_quad<R>(); };
// the per-lane table lookups (role is a lane property) go through global memory once per workgroup
template <int R> struct QuadRingIdx { int v[4][40]; };
template <int R> constexpr QuadRingIdx<R> make_quad_ringidx() { QuadRingIdx<R> o{}; constexpr QuadTab<R> t = make_quad<R>(); for (int r = 0; r < 4; ++r) for (int a = 0; a < 40; ++a) o.v[r][a] = t.ring[r][a]; return o; }
template <int R> __device__ const QuadRingIdx<R> g_quad_ringidx = make_quad_ringidx<R>();

__device__ __forceinline__ float qd_half_mirror(float v) {                    // lane l <- lane l ^ 7 (DPP row_half_mirror)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
}

template <int R, int BUF, int NHs, int HRp, int D = 4>
__device__ __forceinline__ void quad_product(const float4 *halo, const unsigned (&addr)[QuadConst<R>::tab.nl], const f2 (&wp)[QD_P][QuadConst<R>::tab.na / 2],
                                             f2 (&acc)[QD_P][2]) {
    constexpr int NL = QuadConst<R>::tab.nl;
    const char *hb = reinterpret_cast<const char *>(halo) + (size_t)BUF * NHs * 16;
    float4 r[D + 1];
#pragma unroll
    for (int li = 0; li < D; ++li) if (li < NL) r[li] = *reinterpret_cast<const float4 *>(hb + addr[li]);
#pragma unroll
    for (int li = 0; li < NL; ++li) {
        if (li + D < NL) r[(li + D) % (D + 1)] = *reinterpret_cast<const float4 *>(hb + addr[li + D]);
        const float4 rv = r[li % (D + 1)];
        const f2 r01 = {rv.x, rv.y}, r23 = {rv.z, rv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < QuadConst<R>::tab.nf[li]) {
                const int j = QuadConst<R>::tab.fj[li][q], a = QuadConst<R>::tab.fa[li][q];
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
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]) : : "memory");      // keep later reads from being hoisted above these FMAs (register pressure)
    }
}

template <int R>
__global__ void __launch_bounds__(256, 2) k_residual_quad(R1Args a) {
    constexpr int P = QD_P, TR = QD_T, TC = QD_T, NT = 256, NWV = 4;
    constexpr int HR = TR + 2 * R, HC = TC + 2 * R;
    constexpr int HRp = ((HR + 14) / 16) * 16 + 1;                    // == 1 (mod 16)
    constexpr int NHp = HRp * HC;
    constexpr int NIT = (NHp + NT - 1) / NT, NHs = NIT * NT;          // DMA slots per buffer (lane-linear image of [HC][HRp])
    constexpr int NA = QuadConst<R>::tab.na, NW = NA / 2, NL = QuadConst<R>::tab.nl;
    static_assert(NA % 2 == 0 && 4 * NA == RingConst<R>::tab.n, "the four rotated arcs must tile the ring");
    extern __shared__ __attribute__((aligned(16))) float4 lds[];      // halo[2][NHs] | xch[NWV][2][64]
    float4 *halo = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float4 *xch = lds + 2 * NHs + wave * 128;
    const int tmap = a.tile_map[blockIdx.x];
    const int tile_r = tmap & 0xffff, tile_c = tmap >> 16;
    const int hr0 = tile_r * TR + a.roff - R, hc0 = tile_c * TC + a.coff - R;
    // ---- lane -> (role, strip): a role per ds_read_b128 lane group; partner lanes l, l ^ 7 share a strip ----
    const int q5 = lane & 31, quad = q5 >> 2;
    const int par = (quad == 0 || quad == 3 || quad == 5 || quad == 6) ? 0 : 1;
    const int role = 2 * (lane >> 5) + par;
    const int s = par ? ((lane ^ 7) & 15) : (lane & 15);
    int rb, cb, sM, sF;                                               // strip base (tile-local), strides in halo slots
    if (role == 0) { rb = 4 * (s >> 2); cb = 4 * wave + (s & 3); sM = 1; sF = HRp; }
    else if (role == 1) { rb = 4 * (s >> 2) + 3; cb = 4 * wave + (s & 3); sM = -1; sF = -HRp; }
    else if (role == 2) { rb = s; cb = 4 * wave + 3; sM = -HRp; sF = 1; }
    else { rb = s; cb = 4 * wave; sM = HRp; sF = -1; }
    const int base = (cb + R) * HRp + (rb + R);
    // ---- weights of the P centres for the canonical offsets under this lane's rotation, as pairs ----
    f2 wp[P][NW];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int cr = role == 0 ? rb + j : role == 1 ? rb - j : rb, cc = role == 2 ? cb - j : role == 3 ? cb + j : cb;
        const int pr = tile_r * TR + cr, pc = tile_c * TC + cc;
        const int64_t m = (pr < a.nr && pc < a.nc) ? (int64_t)pc * a.nr + pr : 0;      // off-patch centres read pixel 0; never stored
        const uint32_t mb = (uint32_t)m * 4u;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            wp[j][k].x = ld_off(a.W + (int64_t)g_quad_ringidx<R>.v[role][2 * k] * a.d, mb);
            wp[j][k].y = ld_off(a.W + (int64_t)g_quad_ringidx<R>.v[role][2 * k + 1] * a.d, mb);
        }
    }
    // ---- byte offsets of the program's reads inside a halo buffer (frame-invariant) ----
    unsigned addr[NL];
#pragma unroll
    for (int li = 0; li < NL; ++li) addr[li] = (unsigned)((base + QuadConst<R>::tab.rfix[li] * sF + QuadConst<R>::tab.rmov[li] * sM) * 16);
    // ---- DMA plan: instruction j of wave w fills slots (j*NWV + w)*64 + lane of the [HC][HRp] image ----
    uint32_t qoff[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int idx = (j * NWV + wave) * 64 + lane;
        const int hr = idx % HRp, hc = idx / HRp;
        int rbb = hr0 + hr, cbb = hc0 + (hc < HC ? hc : HC - 1);
        rbb = rbb < 0 ? 0 : (rbb >= a.nr_b ? a.nr_b - 1 : rbb);           // halo slots outside the block fetch a clamped address: their weights are exactly 0
        cbb = cbb < 0 ? 0 : (cbb >= a.nc_b ? a.nc_b - 1 : cbb);
        qoff[j] = (uint32_t)(cbb * a.nr_b + rbb) * 16u;
    }
    const unsigned ldsA = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float4 *)lds;
    const unsigned lds0 = ldsA + (unsigned)wave * 1024u;
    // ---- the centre this lane finishes: row lane & 15, column lane >> 4 of the wave's 16 x 4 block ----
    const int fr = lane & 15, fc = 4 * wave + (lane >> 4);
    const int fpr = tile_r * TR + fr, fpc = tile_c * TC + fc;
    const bool fvalid = fpr < a.nr && fpc < a.nc;
    const int64_t fm = fvalid ? (int64_t)fpc * a.nr + fpr : 0;
    const uint32_t fmb = (uint32_t)fm * 4u;
    const float dl = ld_off(a.dlt, fmb);
    const int fslot = (fc + R) * HRp + (fr + R);
    // exchange slots of this lane's four sums (written by the even-role lanes only)
    int xs[P];
#pragma unroll
    for (int j = 0; j < P; ++j) xs[j] = role == 0 ? (s & 3) * 16 + 4 * (s >> 2) + j : 64 + (3 - j) * 16 + s;
    const int probe = __builtin_amdgcn_readfirstlane(a.probe);

    const int64_t cbeg = ((int64_t)blockIdx.y * a.tseg) >> 2;
    const int64_t tend = (int64_t)blockIdx.y * a.tseg + a.tseg < a.T ? (int64_t)blockIdx.y * a.tseg + a.tseg : a.T;
    const int64_t cend = (tend + 3) >> 2;
    auto issue = [&](int64_t c) {
        if ((probe & 1) && c > cbeg + 1) return;
        const int64_t cx = c < cend ? c : cend - 1;
        const float4 *y4 = a.Y4 + cx * a.d_b;
        const unsigned dst = lds0 + (unsigned)((c - cbeg) & 1) * (unsigned)(NHs * 16);
#pragma unroll
        for (int j = 0; j < NIT; ++j) glds16(y4, qoff[j], dst + (unsigned)(j * NWV) * 1024u);
    };
    auto chunk = [&](auto bufc, int64_t c) {
        constexpr int BUF = decltype(bufc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's part of chunk c (and its stores of chunk c-1)
        __builtin_amdgcn_s_barrier();                           // halo(c) complete; everybody is done reading the other buffer
        asm volatile("" ::: "memory");
        issue(c + 1);
        f2 acc[P][2];
#pragma unroll
        for (int j = 0; j < P; ++j) { acc[j][0] = (f2){0.f, 0.f}; acc[j][1] = (f2){0.f, 0.f}; }
        if (!(probe & 2)) quad_product<R, BUF, NHs, HRp>(halo, addr, wp, acc);
        // pair combine: the partner's centres run the other way along the strip
        float4 tot[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            tot[j].x = acc[j][0].x + qd_half_mirror(acc[P - 1 - j][0].x);
            tot[j].y = acc[j][0].y + qd_half_mirror(acc[P - 1 - j][0].y);
            tot[j].z = acc[j][1].x + qd_half_mirror(acc[P - 1 - j][1].x);
            tot[j].w = acc[j][1].y + qd_half_mirror(acc[P - 1 - j][1].y);
        }
        if (par == 0) {
#pragma unroll
            for (int j = 0; j < P; ++j) xch[xs[j]] = tot[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private exchange: the LDS serves a wave's accesses in order
        const float4 pv = xch[lane], ph = xch[64 + lane];
        const float4 cv = halo[BUF * NHs + fslot];
        if (fvalid && !((probe & 8) && cv.x != 12345.f))
        {
            const float4 yo = make_float4(cv.x + dl - (pv.x + ph.x), cv.y + dl - (pv.y + ph.y), cv.z + dl - (pv.z + ph.z), cv.w + dl - (pv.w + ph.w));
            if (probe & 16) st4_off(a.Ysig4 + c * a.d, fmb * 4u, yo);
            else st4_off_wt(a.Ysig4 + c * a.d, fmb * 4u, yo);
        }
        asm volatile("" ::: "memory");
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue(cbeg);
    for (int64_t c = cbeg; c < cend; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < cend) chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int R>
static int launch_r1_quad(cnmfe_ctx *ctx, const R1Args &a, int ntile_c, int64_t nseg) {
    constexpr int HR = QD_T + 2 * R, HC = QD_T + 2 * R, HRp = ((HR + 14) / 16) * 16 + 1, NT = 256;
    constexpr int NIT = (HRp * HC + NT - 1) / NT;
    constexpr size_t shmem = (2 * (size_t)NIT * NT + 4 * 128) * sizeof(float4);
    static_assert(shmem <= 80 * 1024, "two quad-role workgroups must fit one CU's LDS");
    dim3 grid((unsigned)((int64_t)a.ntile_r * ntile_c), (unsigned)nseg);
    CK(hipFuncSetAttribute((const void *)k_residual_quad<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    LAUNCH(ctx, "residual_r1", (k_residual_quad<R>), grid, dim3(NT), shmem, a);
    return 0;
}

}  // namespace cnmfe
