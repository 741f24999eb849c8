// Micro-benchmark: THROUGHPUT of independent integer VALU instructions on one SIMD as a function of the waves resident on it
// (is a kernel with 3-5 waves a SIMD at 0.2 instructions a cycle latency-bound or issue-bound?).
// One workgroup of 256 x W threads on one CU = W waves on each of its 4 SIMDs; every wave runs the same body of eight
// INDEPENDENT instructions (eight registers), so nothing waits for a result; wave 0 times the loop with s_memtime.
//   cycles per wave-instruction per SIMD = cycles / (W x instructions per wave)
// hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define ITER 256
#define REP4(x) x x x x

// body: eight independent instructions on r0..r7 (operands a, b are loop invariants)
#define RATE_KERNEL(name, type, body)                                                              \
    __global__ __launch_bounds__(1024) void name(uint64_t *out, uint32_t seed) {                   \
        type r0 = (type)(seed + threadIdx.x), r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3,              \
             r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;                                   \
        type a = (type)(seed | 3), b = (type)(seed >> 3 | 1);                                      \
        uint32_t s0 = __builtin_amdgcn_readfirstlane(seed), s1 = s0 + 1;                           \
        (void)s0; (void)s1; uint64_t m = __builtin_amdgcn_read_exec() >> (seed & 1); (void)m;                                                                        \
        uint64_t t0 = 0, t1 = 0;                                                                   \
        for (int w = 0; w < 2; w++) {                                                              \
            __syncthreads();                                                                       \
            t0 = __builtin_readcyclecounter();                                                     \
            for (int i = 0; i < ITER; i++) { REP4(body) }                                          \
            t1 = __builtin_readcyclecounter();                                                     \
        }                                                                                          \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                    \
        out[1 + threadIdx.x] = (uint64_t)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) + s0 + s1;       \
    }

#define EIGHT(op) op(r0) op(r1) op(r2) op(r3) op(r4) op(r5) op(r6) op(r7)

#define OP_ADD(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_AND(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_SHL(r) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r));
#define OP_SHLV(r) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r) : "v"(b));
#define OP_PERM(r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define OP_FFBH(r) asm volatile("v_ffbh_u32 %0, %0" : "+v"(r));
#define OP_BFI(r) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(a), "v"(b));
#define OP_BFE(r) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r));
#define OP_MUL24(r) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_MAD24(r) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define OP_MULLO(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_CNDMASK(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(a));
#define OP_CMP(r) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r), "v"(a) : "vcc");
#define OP_LSHLADD(r) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r) : "v"(a));
#define OP_ADD3(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define OP_ANDOR(r) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define OP_ALIGNBIT(r) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define OP_PKADD(r) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_PKSHL(r) asm volatile("v_pk_lshlrev_b16 %0, %1, %0" : "+v"(r) : "v"(b));
#define OP_SDWA(r) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(r) : "v"(b));
#define OP_FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define OP_ADDF(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_SHL64(r) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(r) : "v"((uint32_t)b));
#define OP_ADD64(r) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"(a));
#define OP_MAD64(r) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(r) : "v"((uint32_t)a) : "vcc");
#define OP_READLANE(r) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(r));
#define OP_SUB(r) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_OR(r) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_XOR(r) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_MINU(r) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(a));
#define OP_LSHR(r) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r));
#define OP_ADDCO(r) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r) : "v"(a) : "vcc");
#define OP_CNDS(r) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "s"(m));
#define OP_CMPCND(r) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(a) : "vcc");
#define OP_ANDSDWA(r) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r) : "v"(a));
#define OP_ADDSDWA(r) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r) : "v"(a));
#define OP_ADDLIT(r) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(r));
#define OP_ANDS(r) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "s"(s1));
#define OP_MOV(r) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(a));

RATE_KERNEL(k_add_u32, uint32_t, EIGHT(OP_ADD))
RATE_KERNEL(k_and_b32, uint32_t, EIGHT(OP_AND))
RATE_KERNEL(k_lshl_b32_imm, uint32_t, EIGHT(OP_SHL))
RATE_KERNEL(k_lshl_b32_v, uint32_t, EIGHT(OP_SHLV))
RATE_KERNEL(k_perm_b32, uint32_t, EIGHT(OP_PERM))
RATE_KERNEL(k_ffbh_u32, uint32_t, EIGHT(OP_FFBH))
RATE_KERNEL(k_bfi_b32, uint32_t, EIGHT(OP_BFI))
RATE_KERNEL(k_bfe_u32, uint32_t, EIGHT(OP_BFE))
RATE_KERNEL(k_mul_u32_u24, uint32_t, EIGHT(OP_MUL24))
RATE_KERNEL(k_mad_u32_u24, uint32_t, EIGHT(OP_MAD24))
RATE_KERNEL(k_mul_lo_u32, uint32_t, EIGHT(OP_MULLO))
RATE_KERNEL(k_cndmask_b32, uint32_t, EIGHT(OP_CNDMASK))
RATE_KERNEL(k_cmp_lt_u32, uint32_t, EIGHT(OP_CMP))
RATE_KERNEL(k_lshl_add_u32, uint32_t, EIGHT(OP_LSHLADD))
RATE_KERNEL(k_add3_u32, uint32_t, EIGHT(OP_ADD3))
RATE_KERNEL(k_and_or_b32, uint32_t, EIGHT(OP_ANDOR))
RATE_KERNEL(k_alignbit_b32, uint32_t, EIGHT(OP_ALIGNBIT))
RATE_KERNEL(k_pk_add_u16, uint32_t, EIGHT(OP_PKADD))
RATE_KERNEL(k_pk_lshl_b16, uint32_t, EIGHT(OP_PKSHL))
RATE_KERNEL(k_lshl_sdwa, uint32_t, EIGHT(OP_SDWA))
RATE_KERNEL(k_mov_b32, uint32_t, EIGHT(OP_MOV))
RATE_KERNEL(k_sub_u32, uint32_t, EIGHT(OP_SUB))
RATE_KERNEL(k_or_b32, uint32_t, EIGHT(OP_OR))
RATE_KERNEL(k_xor_b32, uint32_t, EIGHT(OP_XOR))
RATE_KERNEL(k_min_u32, uint32_t, EIGHT(OP_MINU))
RATE_KERNEL(k_lshr_b32_imm, uint32_t, EIGHT(OP_LSHR))
RATE_KERNEL(k_add_co_u32, uint32_t, EIGHT(OP_ADDCO))
RATE_KERNEL(k_cndmask_sgpr, uint32_t, EIGHT(OP_CNDS))
RATE_KERNEL(k_cmp_cndmask_vcc, uint32_t, EIGHT(OP_CMPCND))
RATE_KERNEL(k_and_sdwa, uint32_t, EIGHT(OP_ANDSDWA))
RATE_KERNEL(k_add_sdwa, uint32_t, EIGHT(OP_ADDSDWA))
RATE_KERNEL(k_add_literal, uint32_t, EIGHT(OP_ADDLIT))
RATE_KERNEL(k_and_sgpr, uint32_t, EIGHT(OP_ANDS))
RATE_KERNEL(k_fma_f32, float, EIGHT(OP_FMA))
RATE_KERNEL(k_add_f32, float, EIGHT(OP_ADDF))
RATE_KERNEL(k_lshl_b64, uint64_t, EIGHT(OP_SHL64))
RATE_KERNEL(k_lshl_add_u64, uint64_t, EIGHT(OP_ADD64))
RATE_KERNEL(k_mad_u64_u32, uint64_t, EIGHT(OP_MAD64))
RATE_KERNEL(k_readfirstlane, uint32_t, EIGHT(OP_READLANE))

// mixes: does a scalar instruction or an LDS read between two vector instructions take a vector issue slot?
#define OP_ADD_S(r) asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 3" : "+v"(r), "+s"(s0) : "v"(a) : "scc");
RATE_KERNEL(k_add_u32_plus_salu, uint32_t, EIGHT(OP_ADD_S))   // counts as 8 VALU: the SALU rides or it does not

__global__ __launch_bounds__(1024) void k_add_u32_plus_ds_read(uint64_t *out, uint32_t seed) {
    __shared__ uint32_t lds[4096];
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * seed;
    uint32_t r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    uint32_t a = seed | 3, acc = 0;
    const uint32_t *p = lds + (threadIdx.x & 1023);
    uint64_t t0 = 0, t1 = 0;
    for (int w = 0; w < 2; w++) {
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < ITER; i++) {
            REP4({ uint32_t x = p[(i & 3) * 1024]; asm volatile("" : "+v"(x)); EIGHT(OP_ADD) acc ^= x; })   // one ds_read_b32 per eight adds (+ the xor)
        }
        t1 = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) out[0] = t1 - t0;
    out[1 + threadIdx.x] = uint64_t(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) + acc;
}

// LDS instructions: eight independent ones per body, lane-private addresses (row r of a [rows][64] dword array: conflict-free)
#define LDS_KERNEL(name, body)                                                                      \
    __global__ __launch_bounds__(256) void name(uint64_t *out, uint32_t seed) {                    \
        __shared__ uint32_t lds[8 * 256 * 4];                                                      \
        for (uint32_t i = threadIdx.x; i < 8 * 256 * 4; i += blockDim.x) lds[i] = i * seed;        \
        uint32_t r0 = 0, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, one = seed | 1;   \
        uint32_t ad = (threadIdx.x * 4) + (seed & 1) * 1024;  /* byte address of the lane's column */ \
        uint64_t t0 = 0, t1 = 0;                                                                   \
        for (int w = 0; w < 2; w++) {                                                              \
            __syncthreads();                                                                       \
            t0 = __builtin_readcyclecounter();                                                     \
            for (int i = 0; i < ITER; i++) { REP4(body asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");) }   \
            t1 = __builtin_readcyclecounter();                                                     \
        }                                                                                          \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                    \
        out[1 + threadIdx.x] = (uint64_t)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) + lds[threadIdx.x];  \
    }
#define L8(op) op(r0, 0) op(r1, 1024) op(r2, 2048) op(r3, 3072) op(r4, 4096) op(r5, 5120) op(r6, 6144) op(r7, 7168)
#define DS_RD32(r, o) asm volatile("ds_read_b32 %0, %1 offset:" #o : "=v"(r) : "v"(ad) : "memory");
#define DS_RDU8(r, o) asm volatile("ds_read_u8 %0, %1 offset:" #o : "=v"(r) : "v"(ad) : "memory");
#define DS_RDU16(r, o) asm volatile("ds_read_u16 %0, %1 offset:" #o : "=v"(r) : "v"(ad) : "memory");
#define DS_WR32(r, o) asm volatile("ds_write_b32 %1, %0 offset:" #o : : "v"(r), "v"(ad) : "memory");
#define DS_WR8(r, o) asm volatile("ds_write_b8 %1, %0 offset:" #o : : "v"(r), "v"(ad) : "memory");
#define DS_ADD(r, o) asm volatile("ds_add_u32 %1, %0 offset:" #o : : "v"(one), "v"(ad) : "memory");
#define DS_ADDR(r, o) asm volatile("ds_add_rtn_u32 %0, %1, %2 offset:" #o : "=v"(r) : "v"(ad), "v"(one) : "memory");
#define DS_OR(r, o) asm volatile("ds_or_b32 %1, %0 offset:" #o : : "v"(one), "v"(ad) : "memory");
LDS_KERNEL(k_ds_read_b32, L8(DS_RD32))
LDS_KERNEL(k_ds_read_u8, L8(DS_RDU8))
LDS_KERNEL(k_ds_read_u16, L8(DS_RDU16))
LDS_KERNEL(k_ds_write_b32, L8(DS_WR32))
LDS_KERNEL(k_ds_write_b8, L8(DS_WR8))
LDS_KERNEL(k_ds_add_u32, L8(DS_ADD))
LDS_KERNEL(k_ds_add_rtn_u32, L8(DS_ADDR))
LDS_KERNEL(k_ds_or_b32, L8(DS_OR))
// ds_read_b128 of a 16-byte table entry at a lane-dependent (pseudo-random) index: the look-up of the K1 / K1p steps
__global__ __launch_bounds__(256) void k_ds_read_b128_table(uint64_t *out, uint32_t seed) {
    __shared__ uint4 tab[272];
    for (uint32_t i = threadIdx.x; i < 272; i += blockDim.x) tab[i] = make_uint4(i * seed, i, i + 1, i + 2);
    uint32_t idx = (threadIdx.x * 37 + seed) % 272, acc = 0;
    uint64_t t0 = 0, t1 = 0;
    for (int w = 0; w < 2; w++) {
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < ITER * 4; i++) {
            uint4 e[8];
#pragma unroll
            for (int j = 0; j < 8; j++) e[j] = tab[(idx + 31 * j) % 272];
#pragma unroll
            for (int j = 0; j < 8; j++) acc += e[j].x ^ e[j].y ^ e[j].z ^ e[j].w;
            idx = (idx + (acc & 1) + 7) % 272;
        }
        t1 = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) out[0] = t1 - t0;
    out[1 + threadIdx.x] = acc;
}

__global__ void k_tick(uint64_t *out, uint32_t seed) {
    uint32_t r0 = seed, a = 3;
    const uint64_t c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < 200000; i++) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(r0) : "v"(a)); }
    const uint64_t c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[0] = c1 - c0; out[1] = w1 - w0; out[2] = r0;
}

struct Test { const char *name; void (*fn)(uint64_t *, uint32_t); int per_body; };
#define T(n, k) {#n, n, k}

int main() {
    uint64_t *d;
    CK(hipMalloc(&d, 8 * 1100));
    const Test tests[] = {
        T(k_add_u32, 8), T(k_and_b32, 8), T(k_lshl_b32_imm, 8), T(k_lshl_b32_v, 8), T(k_perm_b32, 8), T(k_ffbh_u32, 8), T(k_bfi_b32, 8), T(k_bfe_u32, 8),
        T(k_mul_u32_u24, 8), T(k_mad_u32_u24, 8), T(k_mul_lo_u32, 8), T(k_cmp_lt_u32, 8), T(k_lshl_add_u32, 8), T(k_add3_u32, 8),
        T(k_and_or_b32, 8), T(k_alignbit_b32, 8), T(k_pk_add_u16, 8), T(k_pk_lshl_b16, 8), T(k_lshl_sdwa, 8), T(k_mov_b32, 8), T(k_sub_u32, 8), T(k_or_b32, 8), T(k_xor_b32, 8), T(k_min_u32, 8), T(k_lshr_b32_imm, 8), T(k_add_co_u32, 8), T(k_cndmask_sgpr, 8), T(k_cmp_cndmask_vcc, 16), T(k_and_sdwa, 8), T(k_add_sdwa, 8), T(k_add_literal, 8), T(k_and_sgpr, 8), T(k_fma_f32, 8), T(k_add_f32, 8),
        T(k_lshl_b64, 8), T(k_lshl_add_u64, 8), T(k_mad_u64_u32, 8), T(k_readfirstlane, 8), T(k_add_u32_plus_salu, 8), T(k_add_u32_plus_ds_read, 9), T(k_ds_read_b32, 8), T(k_ds_read_u8, 8), T(k_ds_read_u16, 8), T(k_ds_write_b32, 8), T(k_ds_write_b8, 8), T(k_ds_add_u32, 8), T(k_ds_add_rtn_u32, 8), T(k_ds_or_b32, 8), T(k_ds_read_b128_table, 8),
    };
    // In real time, chip-wide: 256 x W workgroups of 256 threads (the dispatcher deals them evenly: W waves on every SIMD), timed by events;
    // and what s_memtime counts in, against the 100 MHz s_memrealtime.
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        printf("whole chip, ns per wave-instruction per SIMD (x 2.4 = cycles at 2.4 GHz)\n");
        for (const Test &c : tests) {
            printf("%-26s", c.name);
            for (int w : {1, 2, 4, 8}) {
                hipLaunchKernelGGL(c.fn, dim3(256 * w), dim3(256), 0, 0, d, 1u);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < 20; i++) hipLaunchKernelGGL(c.fn, dim3(256 * w), dim3(256), 0, 0, d, 1u);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double inst = 20.0 * w * 2 * ITER * 4 * c.per_body;   // per SIMD: w waves x 2 passes x ITER x 4 bodies, 20 launches
                printf("  W=%d %.3f", w, ms * 1e6 / inst);
            }
            printf("\n");
        }
    }
    {
        hipLaunchKernelGGL(k_tick, dim3(1), dim3(64), 0, 0, d, 1u);
        uint64_t h[2];
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("s_memtime: %llu ticks in %llu ticks of the 100 MHz s_memrealtime = %.1f MHz\n", (unsigned long long)h[0], (unsigned long long)h[1], double(h[0]) / double(h[1]) * 100.0);
    }
    CK(hipFree(d));
    return 0;
}
