// lds_cas_bench.hip -- ground truth for frag_count_kernel's set (rsqc_k4.h): what an LDS compare-and-swap costs on MI355X as a function of
// its width (ds_cmpst_rtn_b32 / _b64), of the table's load and of the workgroups per CU.  Every thread inserts KPT pseudo-random keys into a
// table of SLOTS entries by linear probing, the table is cleared between rounds (as the kernel does per partition).
// hipcc --offload-arch=gfx950 -O3 tools/lds_cas_bench.hip -o tools/lds_cas_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <class T, int SLOTS, int KPT>
__global__ void __launch_bounds__(256) k_cas(uint32_t rounds, uint32_t *out) {
    __shared__ T tab[SLOTS];
    uint32_t fresh = 0;
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t r = 0; r < rounds; ++r) {
        __syncthreads();
        for (int i = threadIdx.x; i < SLOTS; i += 256) tab[i] = (T)0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            x = x * 1664525u + 1013904223u;
            const uint32_t lo = x | 1u;
            const T k = sizeof(T) == 8 ? (T)(((unsigned long long)(x ^ 0x9E3779B9u) << 32) | lo) : (T)lo;
            uint32_t slot = ((lo * 0x9E3779B1u) >> 16) & (SLOTS - 1);
            for (int probe = 0; probe < SLOTS; ++probe) {
                const T old = atomicCAS(&tab[slot], (T)0, k);
                if (old == (T)0) { ++fresh; break; }
                if (old == k) break;
                slot = (slot + 1) & (SLOTS - 1);
            }
        }
    }
    if (fresh == 0xFFFFFFFFu) out[0] = fresh;
    atomicAdd(&out[1], fresh);
}

template <class T, int SLOTS, int KPT>
void run(const char *name, int blocks) {
    uint32_t *out; hipMalloc(&out, 16); hipMemset(out, 0, 16);
    const uint32_t rounds = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_cas<T, SLOTS, KPT>), dim3(blocks), dim3(256), 0, 0, rounds, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_cas<T, SLOTS, KPT>), dim3(blocks), dim3(256), 0, 0, rounds, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double keys = (double)blocks * 256 * KPT * rounds;
    // cycles per key and CU at 2.4 GHz, 256 CUs
    printf("%-4s slots %5d keys/partition %5d (load %.2f) blocks %5d: %7.3f ms  %6.2f keys/ns  %5.2f cycles per key and CU\n", name, SLOTS, 256 * KPT,
           256.0 * KPT / SLOTS, blocks, ms, keys / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / keys);
    hipFree(out);
}

int main() {
    for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
        run<uint32_t, 2048, 2>("u32", blocks); run<unsigned long long, 2048, 2>("u64", blocks);
        run<uint32_t, 2048, 4>("u32", blocks); run<unsigned long long, 2048, 4>("u64", blocks);
        run<uint32_t, 4096, 4>("u32", blocks); run<unsigned long long, 4096, 4>("u64", blocks);
        run<uint32_t, 4096, 8>("u32", blocks); run<unsigned long long, 4096, 8>("u64", blocks);
    }
    return 0;
}
