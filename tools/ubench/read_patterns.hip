// Micro-benchmark: GB/s of the ways a kernel of this library can read 2-byte records from HBM (event-timed, 640 MB = config 2's records).
//   stream      16 B per lane, consecutive lanes consecutive addresses
//   lane-chunk  a lane per 2 KiB chunk of its own, LINE bytes a trip with the next trip's loads in flight (k_k1p_local, k_k1p_replay)
//   tile        64 chunks interleaved in 16-byte groups: a wave's load is one contiguous KiB (the layout of the one-lane-per-slice kernels)
// hipcc --offload-arch=gfx950 -O3 -o read_patterns read_patterns.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr uint64_t kBytes = 640ull << 20;
constexpr uint32_t kChunk = 2048;

__global__ __launch_bounds__(256) void k_stream(const uint4 *in, uint32_t *out, uint64_t n16) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int LINE, int WORK>
__global__ __launch_bounds__(1024) void k_lane_chunk(const uint4 *in, uint32_t *out, uint64_t n_chunks) {
    const uint64_t c = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
    if (c >= n_chunks) return;
    const uint4 *q = in + c * (kChunk / 16);
    constexpr int N = LINE / 16;
    uint4 cur[N], nxt[N];
#pragma unroll
    for (int j = 0; j < N; j++) cur[j] = q[j];
    uint32_t acc = 0;
#pragma unroll 1
    for (uint32_t t = 0; t < kChunk / LINE; t++) {
#pragma unroll
        for (int j = 0; j < N; j++) nxt[j] = t + 1 < kChunk / LINE ? q[(t + 1) * N + j] : cur[j];
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint32_t x = cur[j].x ^ cur[j].y ^ cur[j].z ^ cur[j].w;
#pragma unroll
            for (int k = 0; k < WORK; k++) x = __builtin_amdgcn_perm(x, acc, x) + k;      // WORK dependent instructions per 16 bytes (8 records)
            acc += x;
        }
#pragma unroll
        for (int j = 0; j < N; j++) cur[j] = nxt[j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int WORK>
__global__ __launch_bounds__(1024) void k_tile(const uint4 *in, uint32_t *out, uint64_t n_chunks) {
    extern __shared__ uint32_t dyn[];
    const uint64_t c = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
    if (c >= n_chunks) return;
    const uint4 *q = in + (c >> 6) * 64 * (kChunk / 16) + (c & 63);       // group g of the tile's lane l at (64 g + l) * 16 bytes
    uint4 cur = q[0];
    uint32_t acc = 0;
#pragma unroll 1
    for (uint32_t t = 0; t < kChunk / 16; t++) {
        const uint4 nxt = t + 1 < kChunk / 16 ? q[(t + 1) * 64] : cur;
        uint32_t x = cur.x ^ cur.y ^ cur.z ^ cur.w;
#pragma unroll
        for (int k = 0; k < WORK; k++) x = __builtin_amdgcn_perm(x, acc, x) + k;
        acc += x;
        cur = nxt;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// D cache lines (64 B each) of the lane's chunk in flight, rolled loop: the ring of registers is rotated by hand
template <int D, int N = 4>      // N x 16 bytes a trip
__global__ __launch_bounds__(1024) void k_lane_deep(const uint4 *in, uint32_t *out, uint64_t n_chunks) {
    extern __shared__ uint32_t dyn[];
    const uint64_t c = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
    if (c >= n_chunks) return;
    const uint4 *q = in + c * (kChunk / 16);
    uint4 ring[D][N];
#pragma unroll
    for (int d = 0; d < D; d++)
#pragma unroll
        for (int j = 0; j < N; j++) ring[d][j] = q[d * N + j];
    uint32_t acc = 0;
    constexpr uint32_t trips = kChunk / (16 * N);
#pragma unroll 1
    for (uint32_t t = 0; t < trips; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            uint4 cur[N];
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = ring[d][j];
            const uint32_t nt = t + d + D < trips ? t + d + D : trips - 1;          // never a load inside a branch: the index is clamped instead
#pragma unroll
            for (int j = 0; j < N; j++) ring[d][j] = q[nt * N + j];
#pragma unroll
            for (int j = 0; j < N; j++) {
                uint32_t x = cur[j].x ^ cur[j].y ^ cur[j].z ^ cur[j].w;
#pragma unroll
                for (int k = 0; k < 8; k++) x = __builtin_amdgcn_perm(x, acc, x) + k;
                acc += x;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc + dyn[0];
}

// the tile pattern with D trips (16 B a lane each) in flight, 1 or 2 bytes a record: what a pass over wave-interleaved records costs at low occupancy
template <int D>
__global__ __launch_bounds__(1024) void k_tile_deep(const uint4 *in, uint32_t *out, uint64_t n_chunks, uint32_t chunk16) {
    extern __shared__ uint32_t dyn[];
    const uint64_t c = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
    if (c >= n_chunks) return;
    const uint4 *q = in + (c >> 6) * 64 * chunk16 + (c & 63);
    uint4 ring[D];
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = q[d * 64];
    uint32_t acc = 0;
#pragma unroll 1
    for (uint32_t t = 0; t < chunk16; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint4 cur = ring[d];
            const uint32_t nt = t + d + D < chunk16 ? t + d + D : chunk16 - 1;
            ring[d] = q[nt * 64];
            uint32_t x = cur.x ^ cur.y ^ cur.z ^ cur.w;
#pragma unroll
            for (int k = 0; k < 8; k++) x = __builtin_amdgcn_perm(x, acc, x) + k;
            acc += x;
        }
    }
    if (acc == 0x12345678u) out[0] = acc + dyn[0];
}

template <class F>
static int timeit(const char *name, F &&launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; i++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %7.3f ms  %7.1f GB/s\n", name, ms / 5, kBytes / (ms / 5 * 1e-3) / 1e9);
    return 0;
}

int main() {
    uint4 *buf; uint32_t *out;
    CK(hipMalloc(&buf, kBytes + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, kBytes + 4096));
    const uint64_t nch = kBytes / kChunk;
    timeit("stream, 16 B a lane", [&] { hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, buf, out, kBytes / 16); });
#define LC(LINE, WORK, BLK) timeit("lane-chunk, " #LINE " B a trip, the next trip's loads inside a branch, work " #WORK ", workgroups of " #BLK, [&] { hipLaunchKernelGGL((k_lane_chunk<LINE, WORK>), dim3((nch + BLK - 1) / BLK), dim3(BLK), 0, 0, buf, out, nch); })
#define TL(WORK, BLK) timeit("tile, 16 B a lane a trip, work " #WORK ", workgroups of " #BLK, [&] { hipLaunchKernelGGL((k_tile<WORK>), dim3((nch + BLK - 1) / BLK), dim3(BLK), 0, 0, buf, out, nch); })
    LC(64, 0, 256); LC(64, 8, 256); LC(64, 64, 256); LC(128, 8, 256); LC(32, 8, 256); LC(16, 8, 256);
    TL(0, 256); TL(8, 256); TL(64, 256);
    // by waves a CU (dynamic LDS holds the occupancy down): tiles with D trips in flight, 2 B records (128 trips a chunk) and 1 B records (64)
    for (uint32_t waves : {8u, 10u, 12u, 20u}) {
        const uint32_t lds = (160u * 1024 / waves / 1024) * 1024 * 4 - 1024;      // per workgroup of 4 waves: waves / 4 workgroups fit
        char name[128];
#define TDEEP(D, C16) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tile_deep<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        snprintf(name, sizeof name, "tile, %d trips in flight, %u B a chunk, %u waves a CU (ms for 310 M records)", D, C16 * 16, waves); \
        if (timeit(name, [&] { hipLaunchKernelGGL((k_tile_deep<D>), dim3((nch + 255) / 256), dim3(256), lds, 0, buf, out, nch, C16); })) return 1; }
        TDEEP(1, 128) TDEEP(2, 128) TDEEP(4, 128) TDEEP(8, 128) TDEEP(1, 64) TDEEP(2, 64) TDEEP(4, 64) TDEEP(8, 64)
#define DEEP(D) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lane_deep<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        snprintf(name, sizeof name, "lane-chunk, %d lines of 64 B in flight, loads unconditional (clamped), %u waves a CU", D, waves); \
        if (timeit(name, [&] { hipLaunchKernelGGL((k_lane_deep<D>), dim3((nch + 255) / 256), dim3(256), lds, 0, buf, out, nch); })) return 1; }
        DEEP(1) DEEP(2) DEEP(4)
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lane_deep<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        snprintf(name, sizeof name, "lane-chunk, one line of 128 B in flight, loads unconditional, %u waves a CU", waves);
        if (timeit(name, [&] { hipLaunchKernelGGL((k_lane_deep<1, 8>), dim3((nch + 255) / 256), dim3(256), lds, 0, buf, out, nch); })) return 1;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lane_deep<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        snprintf(name, sizeof name, "lane-chunk, one half line of 32 B in flight, loads unconditional, %u waves a CU", waves);
        if (timeit(name, [&] { hipLaunchKernelGGL((k_lane_deep<1, 2>), dim3((nch + 255) / 256), dim3(256), lds, 0, buf, out, nch); })) return 1;
    }
    CK(hipFree(buf)); CK(hipFree(out));
    return 0;
}
