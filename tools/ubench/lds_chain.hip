// Micro-benchmark: the state chain of the K1 kernels as the LDS sees it -- per lane and bin: read the context's state from the lane's own
// column, look the (state, bin) entry up in a shared table, write the successor state back -- a chain of two dependent LDS round trips
// a bin.  Whole chip, W waves a SIMD, event-timed: ns per bin per wave (its latency) and ns per wave-bin per CU (the throughput).
// Variants: the state as a byte or as a dword; the table entry 2, 4 or 16 bytes; without the write; two independent chains interleaved.
// hipcc --offload-arch=gfx950 -O3 -o lds_chain lds_chain.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kBins = 4096;
constexpr int kRows = 24;   // state dwords per lane (96 contexts)

// MODE 0: u8 state, u16 table   1: u8 state, u32 table   2: u8 state, uint4 table   3: u32 state (one context per dword), u32 table
// 4: mode 0 without the write-back   5: mode 0, two lanes' worth of chains in one lane (two independent columns, interleaved)
template <int MODE>
__global__ __launch_bounds__(256) void k_chain(uint32_t *out, uint32_t seed) {
    extern __shared__ uint32_t lds[];
    __shared__ uint4 tab16[272];
    __shared__ uint32_t tab4[272];
    __shared__ uint16_t tab2[272];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 272; i += 256) {
        const uint32_t next = (i * 7 + 3) % 126;
        tab16[i] = make_uint4(i * seed, next, i, i + 1); tab4[i] = next | i << 8; tab2[i] = uint16_t(next | i << 8);
    }
    constexpr uint32_t cols = MODE == 5 ? 2 : 1, rows = MODE == 3 ? 96 : kRows;
    uint32_t *col = lds + wv * rows * 64 * cols;
    for (uint32_t i = lane; i < rows * 64 * cols; i += 64) col[i] = (i * 2654435761u >> 8) % 126 * (MODE == 3 ? 1u : 0x01010101u);
    __syncthreads();
    uint8_t *stb = reinterpret_cast<uint8_t *>(col) + lane * 4;
    uint32_t x = seed + threadIdx.x * 977 + blockIdx.x, acc = 0;
    for (int i = 0; i < kBins; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t k = (x >> 10) % 96, bin = (x >> 9) & 1u;
        if (MODE == 3) {
            uint32_t *sp = reinterpret_cast<uint32_t *>(stb + k * 256);
            const uint32_t st = *sp;
            const uint32_t e = tab4[((st & 127u) << 1) | bin];
            *sp = e & 0xffu;
            acc += e;
        } else {
            uint8_t *sp = stb + (k >> 2) * 256 + (k & 3);
            const uint32_t st = *sp;
            uint32_t e;
            if (MODE == 1) e = tab4[(st << 1) | bin];
            else if (MODE == 2) { const uint4 t = tab16[(st << 1) | bin]; e = t.y; acc += t.x ^ t.z ^ t.w; }
            else e = tab2[(st << 1) | bin];
            if (MODE != 4) *sp = uint8_t(e);
            acc += e;
            if (MODE == 5) {
                uint8_t *sq = stb + kRows * 256 + (((k + 17) % 96) >> 2) * 256 + (k & 3);
                const uint32_t s2 = *sq;
                const uint32_t e2 = tab2[(s2 << 1) | (bin ^ 1u)];
                *sq = uint8_t(e2);
                acc += e2;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static int run(const char *name, uint32_t *d) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr uint32_t cols = MODE == 5 ? 2 : 1, rows = MODE == 3 ? 96 : kRows;
    const uint32_t lds = 4 * rows * 64 * cols * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    printf("%-44s", name);
    for (int w : {1, 2, 4, 6, 8}) {
        if ((lds + 8192) * w > 160 * 1024) { printf("  W=%d   -          ", w); continue; }
        hipLaunchKernelGGL(k_chain<MODE>, dim3(256 * w), dim3(256), lds, 0, d, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_chain<MODE>, dim3(256 * w), dim3(256), lds, 0, d, 1u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_bin_wave = ms * 1e6 / (5.0 * kBins * (MODE == 5 ? 2 : 1));   // ns per bin for one wave (its chain's pace)
        printf("  W=%d %6.1f /%5.2f", w, per_bin_wave, per_bin_wave / (4.0 * w));      // ... and ns per wave-bin per CU
    }
    printf("\n");
    return 0;
}

int main() {
    uint32_t *d;
    CK(hipMalloc(&d, 4 * 256 * 8 * 256));
    printf("ns per bin for a wave / ns per wave-bin per CU, by waves a SIMD (x 2.4 = cycles)\n");
    if (run<0>("u8 state, u16 entry, write-back", d)) return 1;
    if (run<1>("u8 state, u32 entry, write-back", d)) return 1;
    if (run<2>("u8 state, 16-byte entry, write-back", d)) return 1;
    if (run<3>("u32 state, u32 entry, write-back", d)) return 1;
    if (run<4>("u8 state, u16 entry, no write-back", d)) return 1;
    if (run<5>("two chains a lane (u8, u16), per chain", d)) return 1;
    CK(hipFree(d));
    return 0;
}
