// Sustained matrix-pipe rate on gfx950 under load (the chip clocks to its power budget): bf16 MFMA shapes x waves per SIMD x operand data,
// with s_memtime (shader cycles) against s_memrealtime (100 MHz).  hipcc --offload-arch=gfx950 -O3 -o scripts/probes/clock_probe scripts/probes/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int SHAPE>
__global__ void __launch_bounds__(256) k_load(long iters, int rnd, unsigned long long *out, float *sink) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    union { uint4 u; float4 v; bf16x8_t h; } a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        const unsigned s = rnd ? hash(threadIdx.x * 8 + i + blockIdx.x * 4096) : 0x3f803f80u;   // random bf16 pairs in [0.5, 2) or all ones
        a[i].u = make_uint4((s & 0x007f007fu) | 0x3f003f00u, (hash(s) & 0x007f007fu) | 0x3f803f80u, (hash(s + 1) & 0x807f807fu) | 0x3f003f00u, (hash(s + 2) & 0x807f807fu) | 0x3f003f00u);
        b[i].u = make_uint4((hash(s + 3) & 0x807f807fu) | 0x3f003f00u, (hash(s + 4) & 0x007f007fu) | 0x3f003f00u, (hash(s + 5) & 0x807f807fu) | 0x3f003f00u, (hash(s + 6) & 0x007f007fu) | 0x3f003f00u);
        if (!rnd) { a[i].u = make_uint4(s, s, s, s); b[i].u = a[i].u; }
    }
    float s = 0.f;
    if (SHAPE == 0) {
        float4_t acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
        for (long it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3].h, b[(i >> 1) & 3].h, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        float16_t acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (long it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].h, b[(i + 1) & 3].h, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (s == 123.456f) sink[0] = s;
}
int main() {
    unsigned long long *d; float *sink; hipMalloc(&d, 1024 * 16); hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(2 * 1024);
    for (int shape = 0; shape < 2; ++shape)
        for (int wps = 1; wps <= 2; ++wps)
            for (int rnd = 0; rnd < 2; ++rnd) {
                const int nwg = 256 * wps;
                const long iters = 600000;
                const double flop_per_iter = shape == 0 ? 8 * 16384.0 : 4 * 32768.0;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (shape == 0) k_load<0><<<nwg, 256>>>(iters, rnd, d, sink); else k_load<1><<<nwg, 256>>>(iters, rnd, d, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
                double cs = 0, rs = 0; for (int i = 0; i < nwg; ++i) { cs += h[2 * i]; rs += h[2 * i + 1]; }
                const double n_mfma_simd = (double)iters * (shape == 0 ? 8 : 4) * wps;
                printf("%s  waves/SIMD %d  data %-6s: %7.2f ms  %6.0f TFLOP/s  %.1f s_memtime cycles per MFMA per SIMD, s_memtime/s_memrealtime x100 = %.0f MHz\n", shape == 0 ? "16x16x32" : "32x32x16", wps,
                       rnd ? "random" : "ones", ms, (double)nwg * 4 * iters * flop_per_iter / (ms * 1e-3) / 1e12, cs / nwg / n_mfma_simd, cs / rs * 100.0);
            }
    return 0;
}
