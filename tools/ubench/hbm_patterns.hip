// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library's kernels
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel moves a KNOWN number of bytes of a 1 GiB buffer (four times the Infinity Cache) in one pattern;
// run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and again with WRITE_SIZE  and divide.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_patterns hbm_patterns.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr uint64_t kBytes = 1ull << 30;
constexpr uint32_t kChunkBytes = 2048;                 // a K1p chunk: 1024 records of 2 bytes

// A: wide coalesced stream, 16 B per lane, consecutive lanes consecutive addresses (the guide's calibrated pattern: reads x2)
__global__ __launch_bounds__(256) void k_read_stream16(const uint4 *in, uint32_t *out, uint64_t n16) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// B: a lane per 2 KB chunk, a 64-byte cache line (four 16 B loads) per trip -- k_k1p_local, k_k1p_replay, k_k1p_c, k_k2p_code
template <int PASSES>
__global__ __launch_bounds__(256) void k_read_lane_chunks(const uint4 *in, uint32_t *out, uint64_t n_chunks) {
    const uint64_t c = blockIdx.x * 256ull + threadIdx.x;
    if (c >= n_chunks) return;
    const uint4 *q = in + c * (kChunkBytes / 16);
    uint32_t acc = 0;
    for (int p = 0; p < PASSES; p++)
        for (uint32_t t = 0; t < kChunkBytes / 64; t++) {
            const uint4 a = q[4 * t], b = q[4 * t + 1], d = q[4 * t + 2], e = q[4 * t + 3];
            acc += (a.x ^ b.y ^ d.z ^ e.w) + p;
        }
    if (acc == 0x12345678u) out[0] = acc;
}
// C: eight lanes per 128-byte line, one line in `stride` -- k_k1p_census
__global__ __launch_bounds__(256) void k_read_line8(const uint4 *in, uint32_t *out, uint64_t n_lines, uint32_t stride) {
    const uint64_t l = (blockIdx.x * 256ull + threadIdx.x) >> 3;
    if (l * stride >= n_lines) return;
    const uint4 v = in[l * stride * 8 + (threadIdx.x & 7)];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = v.x;
}
// D: wide coalesced 16 B stores (the guide's calibrated pattern: exact)
__global__ __launch_bounds__(256) void k_write_stream16(uint4 *o, uint64_t n16) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) o[i] = uint4{uint32_t(i), 1, 2, 3};
}
// E: a lane per chunk of its own, 16 B stores one after the other -- k_k1p_replay's codes, k_k1p_c's digit sums
__global__ __launch_bounds__(256) void k_write_lane_chunks(uint4 *o, uint64_t n_chunks, uint32_t chunk16) {
    const uint64_t c = blockIdx.x * 256ull + threadIdx.x;
    if (c >= n_chunks) return;
    for (uint32_t t = 0; t < chunk16; t++) o[c * chunk16 + t] = uint4{uint32_t(c), t, 2, 3};
}
// F: a lane per chunk, dword stores
__global__ __launch_bounds__(256) void k_write_lane_dwords(uint32_t *o, uint64_t n_chunks, uint32_t chunk4) {
    const uint64_t c = blockIdx.x * 256ull + threadIdx.x;
    if (c >= n_chunks) return;
    for (uint32_t t = 0; t < chunk4; t++) o[c * chunk4 + t] = uint32_t(c) + t;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    uint4 *buf; uint32_t *out;
    CHECK(hipMalloc(&buf, kBytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 1, kBytes));
    const uint64_t n16 = kBytes / 16, n_chunks = kBytes / kChunkBytes, n_lines = kBytes / 128;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_read_stream16, dim3(4096), dim3(256), 0, 0, buf, out, n16);
        hipLaunchKernelGGL(k_read_lane_chunks<1>, dim3((n_chunks + 255) / 256), dim3(256), 0, 0, buf, out, n_chunks);
        hipLaunchKernelGGL(k_read_lane_chunks<2>, dim3((n_chunks + 255) / 256), dim3(256), 0, 0, buf, out, n_chunks);
        hipLaunchKernelGGL(k_read_line8, dim3((n_lines / 16 * 8 + 255) / 256), dim3(256), 0, 0, buf, out, n_lines, 16u);
        hipLaunchKernelGGL(k_write_stream16, dim3(4096), dim3(256), 0, 0, buf, n16);
        hipLaunchKernelGGL(k_write_lane_chunks, dim3((n_chunks + 255) / 256), dim3(256), 0, 0, buf, n_chunks, kChunkBytes / 16);
        hipLaunchKernelGGL(k_write_lane_dwords, dim3((n_chunks + 255) / 256), dim3(256), 0, 0, reinterpret_cast<uint32_t *>(buf), n_chunks, kChunkBytes / 4);
        CHECK(hipDeviceSynchronize());
    }
    printf("bytes moved per launch: k_read_stream16 %llu, k_read_lane_chunks<1> %llu, <2> %llu, k_read_line8 %llu, k_write_* %llu\n",
           (unsigned long long)kBytes, (unsigned long long)kBytes, (unsigned long long)(2 * kBytes), (unsigned long long)(kBytes / 16),
           (unsigned long long)kBytes);
    return 0;
}
