// Micro-benchmark: dependent-chain latency (cycles per instruction, one wave alone on its SIMD) of the instructions the
// K2 range recurrence is made of (DESIGN.md section 4, K2p pass 1).  hipcc --offload-arch=gfx950 -O3 -o chain_latency chain_latency.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 64

#define CHAIN_KERNEL(name, decl, body, sink)                                      \
    __global__ void name(uint64_t *out, uint64_t seed) {                          \
        decl;                                                                     \
        uint64_t t0 = 0, t1 = 0;                                                  \
        for (int w = 0; w < 2; w++) {                                             \
            t0 = __builtin_readcyclecounter();                                    \
            for (int i = 0; i < ITER; i++) { REP16(body) }                        \
            t1 = __builtin_readcyclecounter();                                    \
        }                                                                         \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                   \
        out[1 + threadIdx.x] = (uint64_t)(sink);                                  \
    }

CHAIN_KERNEL(k_fma_f64, double x = (double)seed; double a = 1.0000001; double b = 0.5,
             asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));, x)
CHAIN_KERNEL(k_mul_f64, double x = (double)seed; double a = 1.0000001,
             asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));, x)
CHAIN_KERNEL(k_add_f64, double x = (double)seed; double a = 1.0000001,
             asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(a));, x)
CHAIN_KERNEL(k_trunc_f64, double x = (double)seed,
             asm volatile("v_trunc_f64 %0, %0" : "+v"(x));, x)
CHAIN_KERNEL(k_floor_f64, double x = (double)seed,
             asm volatile("v_floor_f64 %0, %0" : "+v"(x));, x)
CHAIN_KERNEL(k_cvt_u32_f64_rt, double x = (double)seed; uint32_t u = 0,
             asm volatile("v_cvt_u32_f64 %1, %0\n v_cvt_f64_u32 %0, %1" : "+v"(x), "+v"(u));, x)
CHAIN_KERNEL(k_ldexp_f64, double x = (double)seed; int e = 0,
             asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(e));, x)
CHAIN_KERNEL(k_frexp_exp_rt, double x = (double)seed; int e = 0,
             asm volatile("v_frexp_exp_i32_f64 %1, %0\n v_cvt_f64_i32 %0, %1" : "+v"(x), "+v"(e));, x)
CHAIN_KERNEL(k_fma_f32, float x = (float)seed; float a = 1.0001f; float b = 0.5f,
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));, x)
CHAIN_KERNEL(k_add_u32, uint32_t x = (uint32_t)seed; uint32_t a = 3,
             asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));, x)
CHAIN_KERNEL(k_mul_lo_u32, uint32_t x = (uint32_t)seed; uint32_t a = 3,
             asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));, x)
CHAIN_KERNEL(k_mul_hi_u32, uint32_t x = (uint32_t)seed; uint32_t a = 0xfffffff3u,
             asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a));, x)
CHAIN_KERNEL(k_mad_u64_u32, uint64_t x = seed; uint32_t a = 3,
             asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x) : "v"(a) : "vcc");, x)
CHAIN_KERNEL(k_lshl_b64, uint64_t x = seed; uint32_t a = 1,
             asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(x) : "v"(a));, x)
CHAIN_KERNEL(k_ffbh, uint32_t x = (uint32_t)seed,
             asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));, x)
CHAIN_KERNEL(k_cndmask, uint32_t x = (uint32_t)seed; uint32_t a = 5,
             asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");, x)
CHAIN_KERNEL(k_mul_u32_u24, uint32_t x = (uint32_t)seed; uint32_t a = 3,
             asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));, x)
CHAIN_KERNEL(k_mad_u32_u24, uint32_t x = (uint32_t)seed; uint32_t a = 3,
             asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(a));, x)
// scalar unit
CHAIN_KERNEL(k_s_add_u32, uint32_t x = __builtin_amdgcn_readfirstlane((uint32_t)seed),
             asm volatile("s_add_u32 %0, %0, 3" : "+s"(x) : : "scc");, x)
CHAIN_KERNEL(k_s_mul_i32, uint32_t x = __builtin_amdgcn_readfirstlane((uint32_t)seed),
             asm volatile("s_mul_i32 %0, %0, 3" : "+s"(x));, x)
CHAIN_KERNEL(k_s_mul_hi_u32, uint32_t x = __builtin_amdgcn_readfirstlane((uint32_t)seed); uint32_t a = __builtin_amdgcn_readfirstlane(0xfffffff3u),
             asm volatile("s_mul_hi_u32 %0, %0, %1" : "+s"(x) : "s"(a));, x)
CHAIN_KERNEL(k_s_lshl_b64, uint64_t x = __builtin_amdgcn_readfirstlane((uint32_t)seed),
             asm volatile("s_lshl_b64 %0, %0, 1" : "+s"(x) : : "scc");, x)
CHAIN_KERNEL(k_s_flbit_b64, uint64_t x = __builtin_amdgcn_readfirstlane((uint32_t)seed); uint32_t r = 0,
             asm volatile("s_flbit_i32_b64 %1, %0\n s_lshr_b64 %0, %0, %1" : "+s"(x), "+s"(r) : : "scc");, x)
CHAIN_KERNEL(k_s_cselect, uint32_t x = __builtin_amdgcn_readfirstlane((uint32_t)seed); uint32_t a = __builtin_amdgcn_readfirstlane(7),
             asm volatile("s_cmp_lt_u32 %0, %1\n s_cselect_b32 %0, %0, %1" : "+s"(x) : "s"(a) : "scc");, x)
CHAIN_KERNEL(k_s_addc_pair, uint32_t x = __builtin_amdgcn_readfirstlane((uint32_t)seed); uint32_t y = x,
             asm volatile("s_add_u32 %0, %0, %1\n s_addc_u32 %1, %1, %0" : "+s"(x), "+s"(y) : : "scc");, x + y)

// two independent chains in one wave's stream (ILP 2): does a second chain ride for free?
CHAIN_KERNEL(k_fma_f64_x2, double x = (double)seed; double y = x + 1; double a = 1.0000001; double b = 0.5,
             asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(a), "v"(b));, x + y)
CHAIN_KERNEL(k_fma_f64_x4, double x = (double)seed; double y = x + 1; double z = x + 2; double u = x + 3; double a = 1.0000001; double b = 0.5,
             asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                          : "+v"(x), "+v"(y), "+v"(z), "+v"(u) : "v"(a), "v"(b));, x + y + z + u)
CHAIN_KERNEL(k_mad_u64_u32_x2, uint64_t x = seed; uint64_t y = seed + 1; uint32_t a = 3,
             asm volatile("v_mad_u64_u32 %0, vcc, %2, %2, %0\n v_mad_u64_u32 %1, vcc, %2, %2, %1" : "+v"(x), "+v"(y) : "v"(a) : "vcc");, x + y)

// The K2 range recurrence itself (range_step_fp of csrc/avr_k2p.h) with operands that are already in registers: what one bin
// costs when nothing but the chain is there (no record unpacking, no table look-ups, no loads).
#include "../../avrecode-ms_amd/csrc/avr_k2p.h"
__global__ void k_range_step_fp(uint64_t *out, uint64_t seed) {
    using namespace avr::k2p;
    RangeFP r = fp_from_u64(kOne - seed);
    FpConsts K = fp_consts();
    asm volatile("" : "+s"(K.two32), "+s"(K.inv_two32), "+s"(K.split32), "+s"(K.two51), "+s"(K.two47));
    BinFP o[4] = {fp_operands(1 | (3 << 1) | (5 << 8)), fp_operands(0 | (9 << 1) | (2 << 8)), fp_operands(1 | (40 << 1) | (41 << 8)), fp_operands(0 | (1 << 1) | (90 << 8))};
    for (int k = 0; k < 4; k++) asm volatile("" : "+v"(o[k].inv), "+v"(o[k].h), "+v"(o[k].d), "+v"(o[k].ps), "+v"(o[k].nb), "+v"(o[k].inv32));
    uint32_t vmin = 0xffffffffu, bits = 0;
    uint64_t t0 = 0, t1 = 0;
    for (int w = 0; w < 2; w++) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < ITER * 4; i++) {
            bits += range_step_fp(r, vmin, o[0], K); bits += range_step_fp(r, vmin, o[1], K);
            bits += range_step_fp(r, vmin, o[2], K); bits += range_step_fp(r, vmin, o[3], K);
        }
        t1 = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) out[0] = t1 - t0;
    out[1 + threadIdx.x] = fp_to_u64(r) + bits + vmin;
}

struct Test { const char *name; void (*fn)(uint64_t *, uint64_t); int per_body; };
#define T(n, k) {#n, n, k}

int main() {
    uint64_t *d;
    hipMalloc(&d, 8 * 80);
    const Test tests[] = {
        T(k_fma_f64, 1), T(k_mul_f64, 1), T(k_add_f64, 1), T(k_trunc_f64, 1), T(k_floor_f64, 1), T(k_cvt_u32_f64_rt, 2), T(k_ldexp_f64, 1),
        T(k_frexp_exp_rt, 2), T(k_fma_f32, 1), T(k_add_u32, 1), T(k_mul_lo_u32, 1), T(k_mul_hi_u32, 1), T(k_mad_u64_u32, 1),
        T(k_lshl_b64, 1), T(k_ffbh, 1), T(k_cndmask, 2), T(k_mul_u32_u24, 1), T(k_mad_u32_u24, 1),
        T(k_s_add_u32, 1), T(k_s_mul_i32, 1), T(k_s_mul_hi_u32, 1), T(k_s_lshl_b64, 1), T(k_s_flbit_b64, 2), T(k_s_cselect, 2), T(k_s_addc_pair, 2),
        T(k_fma_f64_x2, 2), T(k_fma_f64_x4, 4), T(k_mad_u64_u32_x2, 2),
    };
    for (int lanes : {64, 1}) {                              // cycles per BIN of the K2 range recurrence, operands in registers
        hipLaunchKernelGGL(k_range_step_fp, dim3(1), dim3(lanes), 0, 0, d, 12345ull);
        uint64_t h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-20s lanes %2d: %7.2f cycles per bin (%llu cycles / %d bins)\n", "range_step_fp", lanes, double(h) / (ITER * 16), (unsigned long long)h, ITER * 16);
    }
    for (const Test &t : tests) {
        for (int lanes : {64, 32, 1}) {
            hipLaunchKernelGGL(t.fn, dim3(1), dim3(lanes), 0, 0, d, 12345ull);
            uint64_t h = 0;
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("%-20s lanes %2d: %7.2f cycles per instruction (%llu cycles / %d)\n", t.name, lanes,
                   double(h) / (ITER * 16 * t.per_body), (unsigned long long)h, ITER * 16 * t.per_body);
        }
    }
    hipFree(d);
    return 0;
}
