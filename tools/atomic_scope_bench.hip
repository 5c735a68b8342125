// atomic_scope_bench.hip -- device-scope atomics on one shared array (executed memory-side on MI355X) versus
// workgroup-scope atomics on XCD-private copies (executed in the XCD's own L2), keyed by HW_REG_XCC_ID.
// Checks that the 8 private copies sum to the expected total, and measures (a) throughput and
// (b) the stall a following dependent load sees (vmcnt is in-order on gfx9: a load issued after
// an atomic cannot be waited for without waiting for the atomic's acknowledgement too).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>

__device__ __forceinline__ uint32_t xcc_id() {
    return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7u;   // HW_REG_XCC_ID, bits 3:0
}

template <int SCOPE_WG, int CHASE>
__global__ void k(uint32_t *buf, const uint32_t *chase, uint32_t n_addr, uint32_t per_thread, uint32_t *sink) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t *base = SCOPE_WG ? buf + (size_t)xcc_id() * n_addr : buf;
    uint32_t c = tid & 1023u, acc = 0;
    for (uint32_t it = 0; it < per_thread; ++it) {
        const uint32_t a = (tid * 2654435761u + it * 40503u + (CHASE ? c : 0u)) % n_addr;
        if (SCOPE_WG) __hip_atomic_fetch_add(&base[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else          __hip_atomic_fetch_add(&base[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (CHASE) { c = chase[c]; acc += c; }           // dependent load right after the atomic
    }
    if (CHASE && acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int SCOPE_WG, int CHASE>
void run(uint32_t n_addr, int blocks, uint32_t per_thread) {
    const size_t copies = SCOPE_WG ? 8 : 1;
    uint32_t *buf, *chase, *sink;
    hipMalloc(&buf, copies * n_addr * 4); hipMalloc(&chase, 1024 * 4); hipMalloc(&sink, 4);
    std::vector<uint32_t> hc(1024); for (int i = 0; i < 1024; ++i) hc[i] = (i * 37 + 11) & 1023;
    hipMemcpy(chase, hc.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(buf, 0, copies * n_addr * 4);
    hipLaunchKernelGGL((k<SCOPE_WG, CHASE>), dim3(blocks), dim3(256), 0, 0, buf, chase, n_addr, per_thread, sink);
    hipDeviceSynchronize();
    hipMemset(buf, 0, copies * n_addr * 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SCOPE_WG, CHASE>), dim3(blocks), dim3(256), 0, 0, buf, chase, n_addr, per_thread, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> h(copies * n_addr);
    hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost);
    uint64_t sum = 0; for (uint32_t v : h) sum += v;
    uint64_t per_copy[8] = {0}; if (SCOPE_WG) for (size_t x = 0; x < 8; ++x) for (uint32_t i = 0; i < n_addr; ++i) per_copy[x] += h[x * n_addr + i];
    const uint64_t expect = (uint64_t)blocks * 256 * per_thread;
    printf("%s %s addr=%-9u %8.3f ms %8.2f G atomics/s  sum %s", SCOPE_WG ? "wg-scope/xcd-private" : "agent-scope/shared  ",
           CHASE ? "+dep-load" : "         ", n_addr, ms, expect / ms / 1e6, sum == expect ? "OK" : "MISMATCH");
    if (SCOPE_WG) { printf("  per-xcd:"); for (int x = 0; x < 8; ++x) printf(" %llu", (unsigned long long)per_copy[x]); }
    printf("\n");
    hipFree(buf); hipFree(chase); hipFree(sink);
}

int main() {
    for (uint32_t n_addr : {16u, 1024u, 65536u, 4194304u}) {
        run<0, 0>(n_addr, 2048, 32); run<1, 0>(n_addr, 2048, 32);
        run<0, 1>(n_addr, 2048, 32); run<1, 1>(n_addr, 2048, 32);
    }
    return 0;
}
