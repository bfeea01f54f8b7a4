// Stand-alone test of the ticket-ordered level scheme (round 6): items of level l may start when every item of the levels before (this sweep's and all earlier
// sweeps') is counted done.  hipcc --offload-arch=gfx950 -O3 ticket_test.hip -o ticket_test && ./ticket_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(128) k_levels(const int *__restrict__ off, int nlev, int iters, int *sync, float *val, int variant) {
    __shared__ int s_item;
    const int per_it = off[nlev], total = per_it * iters;
    if (variant & 16) { if (threadIdx.x == 0) atomicAdd(&sync[0], 1); return; }
    if (variant & 32) {                                       // the ticket loop alone
        bool have = false;
        for (;;) {
            if (threadIdx.x == 0) { if (have) atomicAdd(&sync[1], 1); s_item = atomicAdd(&sync[0], 1); }
            __syncthreads();
            const int item = __builtin_amdgcn_readfirstlane(s_item);
            if (item >= total) return;
            have = true;
            __syncthreads();
        }
    }
    bool have = false;
    for (;;) {
        // ONE divergent block per round, at its top (count the item just finished, take the next ticket), and a barrier right behind it: with the counting at the
        // bottom of the loop the structurizer split the loop by lanes and the barriers ran under partial masks, a different number of times per wave -- a hang
        if (threadIdx.x == 0) {
            if (have) { if (variant & 8) atomicAdd(&sync[1], 1); else __hip_atomic_fetch_add(&sync[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
            s_item = atomicAdd(&sync[0], 1);
        }
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(s_item);      // (uniform BY CONSTRUCTION and the compiler must know it: read as a per-lane value the loop becomes a divergent one and its barriers are executed under partial masks -- a hang)
        if (item >= total) return;
        const int it = item / per_it, r = item - it * per_it;
        int l = 0;
        while (off[l + 1] <= r) ++l;
        const int need = it * per_it + off[l];
        if (threadIdx.x == 0 && !(variant & 4)) {
            int spins = 0;
            if (!(variant & 1)) while (__hip_atomic_load(&sync[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            else while (atomicAdd(&sync[1], 0) < need && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 20)) atomicAdd(&sync[2], 1);                                  // gave up: counted, the item runs anyway
        }
        __syncthreads();
        if (!(variant & 2)) __threadfence();
        // work: item r of level l reads the values of level l - 1 (all of them) and writes its own
        float s = 1.f;
        if (l > 0) for (int j = off[l - 1] + threadIdx.x; j < off[l]; j += 128) s += val[j] * 1e-3f;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (threadIdx.x == 0) val[r] = s + (float)it;
        if (!(variant & 2)) __threadfence();
        __syncthreads();
        have = true;
    }
}
int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0, grid = argc > 2 ? atoi(argv[2]) : 340;
    std::vector<int> off = {0, 170, 290, 380, 440, 470, 490};
    const int nlev = (int)off.size() - 1, iters = 3;
    int *dOff, *dSync; float *dVal;
    hipMalloc(&dOff, off.size() * 4); hipMalloc(&dSync, 16); hipMalloc(&dVal, off.back() * 4);
    hipMemcpy(dOff, off.data(), off.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dSync, 0, 16); hipMemset(dVal, 0, off.back() * 4);
    fprintf(stderr, "set up\n"); fflush(stderr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_levels, dim3(grid), dim3(128), 0, 0, dOff, nlev, iters, dSync, dVal, variant);
    hipEventRecord(e1);
    fprintf(stderr, "launched\n"); fflush(stderr);
    hipError_t er = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    int h[4]; hipMemcpy(h, dSync, 16, hipMemcpyDeviceToHost);
    printf("variant %d grid %d: %s, %.3f ms, tickets %d done %d (expected done %d), items that gave up waiting %d\n", variant, grid, hipGetErrorString(er), ms, h[0], h[1], off.back() * iters, h[2]);
    return 0;
}
