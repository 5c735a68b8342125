// atomic_bench.hip -- ground truth for the scatter design: cost of global atomics on MI355X as a
// function of how many lanes / waves hit the same address.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <class T>
__global__ void k_atomic(T *buf, uint32_t n_addr, uint32_t per_thread, int lane_mode) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = tid >> 6;
    for (uint32_t it = 0; it < per_thread; ++it) {
        uint32_t a;
        if (lane_mode == 0) a = (wave * 7919u + it) % n_addr;              // wave-uniform address, one lane issues
        else if (lane_mode == 1) a = (wave * 7919u + it) % n_addr;         // all 64 lanes, same address
        else a = (tid * 2654435761u + it * 40503u) % n_addr;               // per-lane addresses
        if (lane_mode == 0) { if (lane == 0) atomicAdd(&buf[a], (T)1); }
        else atomicAdd(&buf[a], (T)1);
    }
}

template <class T>
void run(const char *name, uint32_t n_addr, int lane_mode, int blocks, uint32_t per_thread) {
    T *buf; hipMalloc(&buf, (size_t)n_addr * sizeof(T)); hipMemset(buf, 0, (size_t)n_addr * sizeof(T));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_atomic<T>, dim3(blocks), dim3(256), 0, 0, buf, n_addr, per_thread, lane_mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_atomic<T>, dim3(blocks), dim3(256), 0, 0, buf, n_addr, per_thread, lane_mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = (double)blocks * 4 * per_thread;                 // wave-level atomic instructions
    const double n_lane = lane_mode == 0 ? n_inst : n_inst * 64;
    printf("%-10s addr=%-9u mode=%d  %8.3f ms  %8.2f G lane-atomics/s  %8.2f ns per wave-instr per address\n", name, n_addr, lane_mode, ms,
           n_lane / ms / 1e6, ms * 1e6 / (n_inst / n_addr));
    hipFree(buf);
}

int main() {
    for (uint32_t n_addr : {1u, 4u, 64u, 1024u, 65536u, 16777216u}) {
        run<unsigned int>("u32", n_addr, 0, 2048, 32);
        run<unsigned int>("u32", n_addr, 1, 2048, 32);
        run<unsigned int>("u32", n_addr, 2, 2048, 32);
        run<unsigned long long>("u64", n_addr, 0, 2048, 32);
        run<double>("f64", n_addr, 0, 2048, 32);
        run<double>("f64", n_addr, 2, 2048, 32);
    }
    return 0;
}
