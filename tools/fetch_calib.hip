// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the per-record kernel
// (VERDICT r5 item 2a).  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (it reports HALF of the
// bytes); the x2 was applied to K1's whole FETCH_SIZE in rounds 3-5, gathers included.  Every kernel below moves a KNOWN number of
// bytes in one pattern; run once per counter:
//   rocprofv3 --pmc FETCH_SIZE  -d out/f -o c -- tools/fetch_calib
//   rocprofv3 --pmc WRITE_SIZE  -d out/w -o c -- tools/fetch_calib
// and compare the counters per kernel name with the `expected` lines this program prints (tools/r6_calib.sh does both and prints the table).
// hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// A: 16 bytes per lane, coalesced, streamed once (the record arrays)
__global__ void stream16(const uint4 *src, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// B: 4 bytes per lane, coalesced (the qhash2 column, tile_span)
__global__ void stream4(const uint32_t *src, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 0x12345678u) *sink = acc;
}
// C: the rank-word pattern: consecutive lanes read the 16-byte word of positions `step` apart (sorted records ~30 positions apart on a
//    table of one word per 64 positions): ascending, duplicate-heavy, every word of the table touched about twice
__global__ void rankwalk16(const uint4 *tab, size_t n_lanes, uint32_t step, size_t n_words, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lanes; i += (size_t)gridDim.x * blockDim.x) {
        size_t w = (i * step) >> 6; if (w >= n_words) w = n_words - 1;
        const uint4 v = tab[w]; acc += v.x ^ v.z;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// C2: the same with each wave owning a contiguous run of lanes (what K1's waves do: a wave's 64 records are neighbours, waves far apart)
__global__ void rankwalk16_tiles(const uint4 *tab, size_t n_tiles, uint32_t step, size_t n_words, uint32_t *sink) {
    uint32_t acc = 0;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t per = (n_tiles + n_waves - 1) / n_waves;
    for (size_t t = wave * per; t < (wave + 1) * per && t < n_tiles; ++t) {
        size_t w = ((t * 64 + (threadIdx.x & 63)) * step) >> 6; if (w >= n_words) w = n_words - 1;
        const uint4 v = tab[w]; acc += v.x ^ v.z;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// D: 16 bytes per lane at random 16-byte-aligned offsets of a 1 GiB buffer (a gather that shares nothing)
__global__ void gather16_random(const uint4 *src, size_t n_lanes, size_t n_words, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lanes; i += (size_t)gridDim.x * blockDim.x) {
        const size_t w = ((size_t)mix((uint32_t)i) * 2654435761ull + mix((uint32_t)(i >> 7))) % n_words;
        const uint4 v = src[w]; acc += v.x ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// E: 32 bytes per lane from a 10 MB table (the interval entries: cache-resident)
__global__ void gather32_small(const uint4 *src, size_t n_lanes, size_t n_entries, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lanes; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = mix((uint32_t)i) % n_entries;
        const uint4 v = src[2 * e], u = src[2 * e + 1]; acc += v.x ^ u.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// F: stores -- 16 bytes per lane coalesced (the pairs), 12 bytes per lane at random 12-byte offsets (frag_local's key scatter),
//    4 bytes per lane coalesced
__global__ void store16(uint4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
struct K12 { uint32_t a, b, c; };
__global__ void store12_random(K12 *dst, size_t n_lanes, size_t n_slots) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lanes; i += (size_t)gridDim.x * blockDim.x) {
        const size_t w = ((size_t)mix((uint32_t)i) * 2654435761ull + mix((uint32_t)(i >> 7))) % n_slots;
        dst[w] = K12{(uint32_t)i, 7u, 9u};
    }
}
__global__ void store4(uint32_t *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint32_t)i;
}
// G: memory atomics without return -- the coverage difference array: ascending with neighbours `step` words apart per lane (a sorted
//    stream over a 0.5 GB array) and at random
__global__ void atomic_walk(uint32_t *cov, size_t n_lanes, uint32_t step, size_t n_words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lanes; i += (size_t)gridDim.x * blockDim.x) {
        size_t w = i * step + (mix((uint32_t)i) & 7u); if (w >= n_words) w = n_words - 1;
        atomicAdd(&cov[w], 1u);
    }
}
__global__ void atomic_random(uint32_t *cov, size_t n_lanes, size_t n_words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lanes; i += (size_t)gridDim.x * blockDim.x) {
        const size_t w = ((size_t)mix((uint32_t)i) * 2654435761ull + mix((uint32_t)(i >> 7))) % n_words;
        atomicAdd(&cov[w], 1u);
    }
}

int main() {
    const size_t GiB = 1ull << 30;
    void *buf = nullptr; uint32_t *sink = nullptr;
    CK(hipMalloc(&buf, 2 * GiB)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, 2 * GiB));
    const dim3 g(256 * 16), b(256);
    auto flush = [&]() { (void)hipDeviceSynchronize(); };
    // (every kernel runs twice: rocprofv3 reports each dispatch; the second finds the Infinity Cache as the first left it -- buffers are
    //  >= 512 MB, beyond its 256 MB, except E)
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream16, g, b, 0, 0, (const uint4 *)buf, 2 * GiB / 16, sink); flush();
        hipLaunchKernelGGL(stream4, g, b, 0, 0, (const uint32_t *)buf, GiB / 4, sink); flush();
        hipLaunchKernelGGL(rankwalk16, g, b, 0, 0, (const uint4 *)buf, (size_t)100 << 20, 30u, (size_t)775 * 1000000 / 16, sink); flush();
        hipLaunchKernelGGL(rankwalk16_tiles, g, b, 0, 0, (const uint4 *)buf, (size_t)(100 << 20) / 64, 30u, (size_t)775 * 1000000 / 16, sink); flush();
        hipLaunchKernelGGL(gather16_random, g, b, 0, 0, (const uint4 *)buf, (size_t)32 << 20, GiB / 16, sink); flush();
        hipLaunchKernelGGL(gather32_small, g, b, 0, 0, (const uint4 *)buf, (size_t)64 << 20, (size_t)10 * 1000000 / 32, sink); flush();
        hipLaunchKernelGGL(store16, g, b, 0, 0, (uint4 *)buf, GiB / 16); flush();
        hipLaunchKernelGGL(store12_random, g, b, 0, 0, (K12 *)buf, (size_t)32 << 20, GiB / 12); flush();
        hipLaunchKernelGGL(store4, g, b, 0, 0, (uint32_t *)buf, GiB / 8); flush();
        hipLaunchKernelGGL(atomic_walk, g, b, 0, 0, (uint32_t *)buf, (size_t)64 << 20, 2u, (size_t)512 * 1000000 / 4); flush();
        hipLaunchKernelGGL(atomic_random, g, b, 0, 0, (uint32_t *)buf, (size_t)32 << 20, (size_t)512 * 1000000 / 4); flush();
    }
    CK(hipDeviceSynchronize());
    // expected useful bytes per dispatch (what an ideal memory system moves), as `kernel read_bytes write_bytes note`
    const double rw_words = (double)(100 << 20) * 30 / 64;
    printf("expected stream16 %.0f 0 coalesced-16B\n", (double)(2 * GiB));
    printf("expected stream4 %.0f 0 coalesced-4B\n", (double)GiB);
    printf("expected rankwalk16 %.0f 0 distinct-16B-words(%.0f)x16;sectors64=%.0f\n", rw_words * 16, rw_words, rw_words * 16);
    printf("expected rankwalk16_tiles %.0f 0 same-words-as-rankwalk16\n", rw_words * 16);
    printf("expected gather16_random %.0f 0 lanes-x16;x64-per-sector=%.0f\n", (double)(32 << 20) * 16, (double)(32 << 20) * 64);
    printf("expected gather32_small 0 0 table-10MB-cache-resident\n");
    printf("expected store16 0 %.0f coalesced-16B\n", (double)GiB);
    printf("expected store12_random 0 %.0f lanes-x12;x64-per-sector=%.0f\n", (double)(32 << 20) * 12, (double)(32 << 20) * 64);
    printf("expected store4 0 %.0f coalesced-4B\n", (double)GiB / 2);
    printf("expected atomic_walk %.0f %.0f lanes-x4-touching-%.0f-bytes-once\n", (double)(64 << 20) * 2 * 4, (double)(64 << 20) * 2 * 4, (double)(64 << 20) * 2 * 4);
    printf("expected atomic_random %.0f %.0f lanes-x64-per-sector\n", (double)(32 << 20) * 64, (double)(32 << 20) * 64);
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}
