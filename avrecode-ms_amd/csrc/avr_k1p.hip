// K1p kernels: intra-slice parallel CABAC encode for batches of few, long slices.
// The algorithm and its per-lane functions are in avr_k1p.h; this file maps them to lanes:
//
//   k_k1p_resolve   phase A   one lane per slice     records + initial states -> resolved codes
//   k_k1p_b1        phase B1  one lane per chunk     stretch summaries for the 4 entry quarters
//   k_k1p_b2        phase B2  one lane per slice     chain the summaries: entry range + bit position
//   k_k1p_zero                one workgroup per slice zero the digit sums that will be used
//   k_k1p_c         phase C   one lane per chunk     code each stretch, add its digits
//   k_k1p_d         phase D   one lane per slice     carries, finish(), bytes
//
// Results are byte-identical to k_cabac_encode (tests/test_gpu_k1p.py); a slice the scheme
// cannot take (a stretch with no LPS for 16 chunks) is handed to k_cabac_encode itself.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "avr_internal.h"
#include "avr_k1p.h"
#include "avr_tables.h"

namespace avr {

static __device__ const CabacTables d_tables = make_cabac_tables();

using namespace k1p;

// ------------------------------------------------------------------ phase A

// One lane per slice, one wave per workgroup; LDS layout as in k_cabac_encode (table, then
// state dwords (k, lane), then one scratch row).  Reads the tile layout, writes 8 resolved
// codes (8 bytes) per 8-record chunk.
__global__ __launch_bounds__(64) void k_k1p_resolve(
    const uint4 *tiles, const uint64_t *tile_off, const uint32_t *n_bins, const uint32_t *order,
    uint32_t n_slices, const uint8_t *init_states, uint32_t n_states,
    uint8_t *res, const uint64_t *res_off, int32_t *status, uint8_t *final_states) {
    extern __shared__ uint32_t lds[];
    uint32_t *next = lds;                                        // 136 entries: MPS | LPS << 8 successor
    uint32_t *st32 = lds + 136;
    uint8_t *st8 = reinterpret_cast<uint8_t *>(st32);
    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x * 64 + lane;
    for (uint32_t i = lane; i < 136; i += 64) next[i] = i < 128 ? d_tables.packed[i][1] : 0;

    const bool in_range = g < n_slices;
    const uint32_t slice = in_range ? (order ? order[g] : g) : 0;
    int32_t st = in_range ? status[slice] : AVR_SLICE_OK;
    const bool active = in_range && st == AVR_SLICE_OK;
    const uint32_t nb = active ? n_bins[slice] : 0;
    const uint32_t ns4 = (n_states + 3) >> 2;
    if (active) {
        const uint8_t *src = init_states + size_t(slice) * n_states;
        for (uint32_t k = 0; k < ns4; k++) {
            uint32_t v = 0;
            for (uint32_t b = 0; b < 4; b++)
                if (4 * k + b < n_states) v |= uint32_t(src[4 * k + b]) << (8 * b);
            st32[k * 64 + lane] = v;
        }
    }
    __syncthreads();

    const uint32_t lane4 = lane * 4, scratch = ns4 * 256 + lane * 4;
    const uint4 *src = tiles + tile_off[g >> 6] + (g & 63);
    uint2 *dst = reinterpret_cast<uint2 *>(res + (in_range ? res_off[slice] : 0));
    const uint32_t n_chunks = (nb + 7) >> 3;
    uint32_t term_at = 0xffffffffu;
    const uint4 nop4 = make_uint4(AVR_NOP_CABAC2, AVR_NOP_CABAC2, AVR_NOP_CABAC2, AVR_NOP_CABAC2);
    uint4 cur = n_chunks > 0 ? src[0] : nop4;
    for (uint32_t c = 0; c < n_chunks; c++) {
        const uint4 nxt = (c + 1 < n_chunks) ? src[size_t(c + 1) * 64] : nop4;
        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t codes[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t rec = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
            const uint32_t bin = rec & 1, sel = (rec >> 1) & 0x7ffu;
            const bool is_ctx = sel < n_states;
            const uint32_t saddr = is_ctx ? (((sel >> 2) << 8) + (sel & 3) + lane4) : scratch;
            uint32_t s_mem = st8[saddr];
            asm volatile("" : "+v"(s_mem));                     // keep the LDS read unconditional (see k_cabac_encode)
            const uint32_t s = is_ctx ? (s_mem & 127u) : 130u;   // 130: any entry >= 128 (successor 0, unused)
            uint32_t nx = next[s];
            asm volatile("" : "+v"(nx));
            const uint32_t sym = (bin ^ s) & 1;
            st8[saddr] = uint8_t(sym ? (nx >> 8) : nx);          // cabac_code.h:43-47
            const uint32_t code = is_ctx ? code_context(s, bin)
                                : sel == AVR_SEL_BYPASS ? (kCodeBypass | bin)
                                : sel == AVR_SEL_TERMINATE ? code_terminate(bin) : kCodePad;
            codes[j >> 2] |= code << ((j & 3) * 8);
            if (rec == ((AVR_SEL_TERMINATE << 1) | 1) && term_at == 0xffffffffu) term_at = c * 8 + j;
        }
        dst[c] = make_uint2(codes[0], codes[1]);
        cur = nxt;
    }
    if (active) {
        if (n_chunks & 1) dst[n_chunks] = make_uint2(0xfcfcfcfcu, 0xfcfcfcfcu);   // pad to 16 bytes with kCodePad
        if (term_at != 0xffffffffu && term_at + 1 < nb) status[slice] = AVR_SLICE_BAD_RECORD;   // a bin after finish()
        if (final_states) {
            uint8_t *fs = final_states + size_t(slice) * n_states;
            for (uint32_t k = 0; k < ns4; k++) {
                const uint32_t v = st32[k * 64 + lane];
                for (uint32_t b = 0; b < 4; b++)
                    if (4 * k + b < n_states) fs[4 * k + b] = uint8_t(v >> (8 * b));
            }
        }
    }
}

// ------------------------------------------------------------------ phases B1, B2, C, D

struct Plan {                       // device pointers of the per-slice / per-chunk plan (see the C ABI)
    const uint32_t *n_bins;
    const uint64_t *res_off;        // bytes, multiples of 16
    const uint32_t *chunk_base;     // first global chunk of each slice (n_slices + 1)
    const uint32_t *chunk_slice;    // slice of each global chunk
    const uint64_t *dig_off;        // first digit sum of each slice (n_slices + 1)
};

__global__ __launch_bounds__(256) void k_k1p_b1(Plan p, uint32_t total_chunks, const uint8_t *res,
                                                const int32_t *status, Stretch *st) {
    __shared__ uint32_t rows[64];
    if (threadIdx.x < 64) rows[threadIdx.x] = d_tables.packed[2 * threadIdx.x][0];
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const uint32_t slice = p.chunk_slice[gc];
    if (status[slice] != AVR_SLICE_OK) { st[gc].first = kNone; st[gc].too_long = 0; return; }
    Stretch o;
    b1_stretch(res + p.res_off[slice], p.n_bins[slice], gc - p.chunk_base[slice], rows, &o);
    st[gc] = o;
}

__global__ __launch_bounds__(64) void k_k1p_b2(Plan p, uint32_t n_slices, const int32_t *status,
                                               const Stretch *st, Entry *en, SliceTotals *tot) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_slices) return;
    if (status[s] != AVR_SLICE_OK) { tot[s].t_total = 0; tot[s].r_final = 510; tot[s].bad = 0; return; }
    const uint32_t c0 = p.chunk_base[s];
    b2_chain(st + c0, p.chunk_base[s + 1] - c0, en + c0, &tot[s]);
}

__global__ __launch_bounds__(256) void k_k1p_zero(Plan p, const SliceTotals *tot, uint32_t *S) {
    const uint32_t s = blockIdx.x;
    const uint32_t n = ref_digits(tot[s].t_total) + 2;
    uint32_t *d = S + p.dig_off[s];
    for (uint32_t i = threadIdx.x; i < n; i += 256) d[i] = 0;
}

struct DeviceAdder {
    uint32_t *S;
    __device__ void store(uint32_t i, uint32_t v) { S[i] = v; }
    __device__ void add(uint32_t i, uint32_t v) { atomicAdd(&S[i], v); }
};

__global__ __launch_bounds__(256) void k_k1p_c(Plan p, uint32_t total_chunks, const uint8_t *res,
                                               const Stretch *st, const Entry *en, const SliceTotals *tot,
                                               uint32_t *S) {
    __shared__ uint32_t rows[64];
    if (threadIdx.x < 64) rows[threadIdx.x] = d_tables.packed[2 * threadIdx.x][0];
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const Stretch o = st[gc];
    if (o.first == kNone) return;
    const uint32_t slice = p.chunk_slice[gc];
    if (tot[slice].bad) return;
    DeviceAdder add{S + p.dig_off[slice]};
    c_stretch(res + p.res_off[slice], o, en[gc], gc - p.chunk_base[slice], rows, add);
}

__global__ __launch_bounds__(64) void k_k1p_d(Plan p, uint32_t n_slices, const SliceTotals *tot, const uint32_t *S,
                                              uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                                              int32_t *status) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_slices) return;
    if (status[s] != AVR_SLICE_OK) { out_len[s] = 0; return; }
    if (tot[s].bad) { status[s] = AVR_SLICE_RETRY_SERIAL; return; }
    const uint32_t cap = uint32_t(out_off[s + 1] - out_off[s]);
    const uint32_t len = d_slice(S + p.dig_off[s], tot[s], out + out_off[s], cap);
    out_len[s] = len;
    if (len > cap) status[s] = AVR_SLICE_OVERFLOW;
}

// ------------------------------------------------------------------ launcher

size_t k1p_workspace_bytes(size_t n_slices, uint64_t res_total, uint32_t total_chunks, uint64_t dig_total) {
    auto up = [](uint64_t x) { return (x + 255) & ~uint64_t(255); };
    return size_t(up(res_total + 16) + up(uint64_t(total_chunks) * sizeof(Stretch)) +
                  up(uint64_t(total_chunks) * sizeof(Entry)) + up(n_slices * sizeof(SliceTotals)) +
                  up(dig_total * 4 + 16));
}

hipError_t launch_k1p(hipStream_t s, const void *tiles, const uint64_t *tile_off, const uint32_t *n_bins,
                      const uint32_t *order, uint32_t n_slices, const uint8_t *init_states, uint32_t n_states,
                      const uint64_t *res_off, uint64_t res_total, const uint32_t *chunk_base,
                      const uint32_t *chunk_slice, uint32_t total_chunks, const uint64_t *dig_off,
                      uint64_t dig_total, void *workspace, uint8_t *out, const uint64_t *out_off,
                      uint32_t *out_len, int32_t *status, uint8_t *final_states) {
    if (n_slices == 0) return hipSuccess;
    auto up = [](uint64_t x) { return (x + 255) & ~uint64_t(255); };
    uint8_t *w = static_cast<uint8_t *>(workspace);
    uint8_t *res = w;                   w += up(res_total + 16);
    Stretch *st = reinterpret_cast<Stretch *>(w);        w += up(uint64_t(total_chunks) * sizeof(Stretch));
    Entry *en = reinterpret_cast<Entry *>(w);            w += up(uint64_t(total_chunks) * sizeof(Entry));
    SliceTotals *tot = reinterpret_cast<SliceTotals *>(w); w += up(n_slices * sizeof(SliceTotals));
    uint32_t *S = reinterpret_cast<uint32_t *>(w);
    const Plan p{n_bins, res_off, chunk_base, chunk_slice, dig_off};

    const uint32_t lds = 136 * 4 + 64 * 4 * ((n_states + 3) / 4 + 1);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_k1p_resolve),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    const uint32_t slice_blocks = (n_slices + 63) / 64, chunk_blocks = (total_chunks + 255) / 256;
    hipLaunchKernelGGL(k_k1p_resolve, dim3(slice_blocks), dim3(64), lds, s, static_cast<const uint4 *>(tiles), tile_off,
                       n_bins, order, n_slices, init_states, n_states, res, res_off, status, final_states);
    hipLaunchKernelGGL(k_k1p_b1, dim3(chunk_blocks), dim3(256), 0, s, p, total_chunks, res, status, st);
    hipLaunchKernelGGL(k_k1p_b2, dim3(slice_blocks), dim3(64), 0, s, p, n_slices, status, st, en, tot);
    hipLaunchKernelGGL(k_k1p_zero, dim3(n_slices), dim3(256), 0, s, p, tot, S);
    hipLaunchKernelGGL(k_k1p_c, dim3(chunk_blocks), dim3(256), 0, s, p, total_chunks, res, st, en, tot, S);
    hipLaunchKernelGGL(k_k1p_d, dim3(slice_blocks), dim3(64), 0, s, p, n_slices, tot, S, out, out_off, out_len, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // slices the scheme declined (status AVR_SLICE_RETRY_SERIAL) are coded by the serial kernel
    return launch_cabac_encode(true, s, tiles, tile_off, n_bins, order, n_slices, init_states, n_states, out, out_off,
                               out_len, status, nullptr, AVR_SLICE_RETRY_SERIAL);
}

}  // namespace avr
