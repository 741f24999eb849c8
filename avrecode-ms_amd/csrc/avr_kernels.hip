// HIP kernels of the arithmetic re-encode path for gfx950 (MI355X).
//
//   K1  k_cabac_encode   cabac::encoder over recorded bins        (cabac_code.h:26-82
//                        on arithmetic_code<uint32_t,uint16_t,0x200>, decompress direction,
//                        recode.cpp:1442-1481)
//   K2  k_range_encode   recoded_code::encoder over recorded bins (arithmetic_code<uint64_t,
//                        uint8_t>, compress direction, recode.cpp:1075-1103, 823-827)
//   k_pack_tiles         slice-major records -> wave-interleaved tiles
//   k_synth_*            seeded synthetic bin streams (avr_synth.h)
//
// Mapping: one lane per slice (each slice has its own coder object in the reference,
// recode.cpp:1270, 1525, so slices are independent and a slice is strictly serial).
// One wave (64 slices) per workgroup.  Integer work only; no MFMA.
//
// LDS per workgroup (K1):
//   [0, 1 KiB)          packed CABAC table, 128 x 8 B (avr_tables.h)
//   [1 KiB, ...)        context states, dword (k, lane) at 4*(k*64 + lane) holds the four
//                       state bytes 4k..4k+3 of the lane's slice: the bank is lane % 32 for
//                       every context, so the data-dependent state read and write of a bin
//                       never conflict across lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "avr_coder.h"
#include "avr_internal.h"
#include "avr_synth.h"
#include "avr_tables.h"

namespace avr {

__device__ const CabacTables d_tables = make_cabac_tables();

// ------------------------------------------------------------------ record fetch

// Where lane `lane` of processing slot `g` finds chunk c (8 records = 16 bytes).
template <bool TILED>
struct ChunkSource {
    const uint4 *p;
    uint32_t stride;       // in uint4
    __device__ ChunkSource(const void *recs, const uint64_t *off, uint32_t g, uint32_t slice) {
        if (TILED) {       // off = tile_off (16-byte units), one entry per 64 slots
            p = reinterpret_cast<const uint4 *>(recs) + off[g >> 6] + (g & 63);
            stride = 64;
        } else {           // off = rec_off (records), indexed by slice
            p = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(recs) + off[slice]);
            stride = 1;
        }
    }
    __device__ __forceinline__ uint4 load(uint32_t c) const { return p[size_t(c) * stride]; }
};

__device__ __forceinline__ uint32_t chunk_rec(const uint4 &v, int j) {
    const uint32_t w = j < 2 ? v.x : j < 4 ? v.y : j < 6 ? v.z : v.w;
    return (j & 1) ? (w >> 16) : (w & 0xffffu);
}

// ------------------------------------------------------------------ K1

template <bool TILED>
__global__ __launch_bounds__(64) void k_cabac_encode(
    const void *recs, const uint64_t *off, const uint32_t *n_bins, const uint32_t *order,
    uint32_t n_slices, const uint8_t *init_states, uint32_t n_states,
    uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status,
    uint8_t *final_states) {
    extern __shared__ uint32_t lds[];
    uint2 *tab = reinterpret_cast<uint2 *>(lds);                 // 128 entries
    uint32_t *st32 = lds + 256;                                  // state dwords
    uint8_t *st8 = reinterpret_cast<uint8_t *>(st32);

    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x * 64 + lane;
    for (uint32_t i = lane; i < 128; i += 64)
        tab[i] = make_uint2(d_tables.packed[i][0], d_tables.packed[i][1]);

    const bool active = g < n_slices;
    const uint32_t slice = active ? (order ? order[g] : g) : 0;
    const uint32_t nb = active ? n_bins[slice] : 0;
    const uint32_t ns4 = (n_states + 3) >> 2;

    // states: global (n_states bytes per slice) -> LDS column of this lane
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

    RangeEncoder<uint32_t, 32, 16> e;
    const uint64_t o0 = active ? out_off[slice] : 0;
    const uint32_t cap = active ? uint32_t(out_off[slice + 1] - o0) : 0;
    e.init(0x7F800000u, out + o0, cap);                          // cabac_code.h:30
    int32_t st = AVR_SLICE_OK;

    const ChunkSource<TILED> src(recs, off, g, slice);
    const uint32_t n_chunks = (nb + 7) >> 3;
    uint4 cur = n_chunks ? src.load(0) : make_uint4(0, 0, 0, 0);
    for (uint32_t c = 0; c < n_chunks; c++) {
        const uint4 nxt = (c + 1 < n_chunks) ? src.load(c + 1) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (c * 8 + j >= nb) break;
            const uint32_t rec = chunk_rec(cur, j);
            const uint32_t bin = rec & 1, sel = (rec >> 1) & 0x7ff;
            if (e.range == 0 || sel > AVR_SEL_TERMINATE || (sel < 1024 && sel >= n_states)) {
                st = AVR_SLICE_BAD_RECORD;                       // bin after finish(), or bad selector
                c = n_chunks;
                break;
            }
            const bool is_ctx = sel < 1024;
            const uint32_t saddr = is_ctx ? (((sel >> 2) * 64 + lane) * 4 + (sel & 3)) : lane * 4;
            const uint32_t s = is_ctx ? (st8[saddr] & 127u) : 0;
            const uint2 ent = tab[s];
            // normalize = floor(log2(range / 0x100)) (cabac_code.h:37,59,70-79)
            const int norm = 23 - __clz(e.range);
            const uint32_t q = (e.range >> (norm + 6)) & 3;      // (range_approx & 0x180) >> 7, :39-40
            const uint32_t rlps = ((ent.x >> (q * 8)) & 0xff) << norm;           // :40-41
            const uint32_t r1 = is_ctx ? rlps                                    // :35
                              : (sel == AVR_SEL_BYPASS ? (e.range >> 1)          // :53
                                                       : (2u << norm));          // :60
            const uint32_t sym = is_ctx ? (bin ^ (s & 1)) : bin;                 // :34
            // arithmetic_code.h:107-114
            const uint32_t r0 = e.range - r1;
            e.low += sym ? r0 : 0;
            e.range = sym ? r1 : r0;
            if (is_ctx) st8[saddr] = uint8_t(sym ? (ent.y >> 8) : ent.y);        // cabac_code.h:43-47
            if (e.range < 0x200u) e.emit_digit();                // arithmetic_code.h:115-122 (one digit)
            if (sel == AVR_SEL_TERMINATE && bin) e.finish();     // cabac_code.h:63-65
        }
        cur = nxt;
    }
    if (active) {
        if (e.range != 0 && st == AVR_SLICE_OK) e.finish();      // ~encoder(), arithmetic_code.h:100
        e.w.flush();
        if (st == AVR_SLICE_OK && e.w.n > cap) st = AVR_SLICE_OVERFLOW;
        out_len[slice] = e.w.n;
        status[slice] = st;
        if (final_states) {
            uint8_t *dst = final_states + size_t(slice) * n_states;
            for (uint32_t k = 0; k < ns4; k++) {
                const uint32_t v = st32[k * 64 + lane];
                for (uint32_t b = 0; b < 4; b++)
                    if (4 * k + b < n_states) dst[4 * k + b] = uint8_t(v >> (8 * b));
            }
        }
    }
}

// ------------------------------------------------------------------ K2

template <bool TILED>
__global__ __launch_bounds__(64) void k_range_encode(
    const void *recs, const uint64_t *off, const uint32_t *n_bins, const uint32_t *order,
    uint32_t n_slices, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status) {
    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x * 64 + lane;
    if (g >= n_slices) return;
    const uint32_t slice = order ? order[g] : g;
    const uint32_t nb = n_bins[slice];

    RangeEncoder<uint64_t, 64, 8> e;
    const uint64_t o0 = out_off[slice];
    const uint32_t cap = uint32_t(out_off[slice + 1] - o0);
    e.init(uint64_t(1) << 63, out + o0, cap);                    // arithmetic_code.h:96-97
    int32_t st = AVR_SLICE_OK;

    const ChunkSource<TILED> src(recs, off, g, slice);
    const uint32_t n_chunks = (nb + 7) >> 3;
    uint4 cur = n_chunks ? src.load(0) : make_uint4(0, 0, 0, 0);
    for (uint32_t c = 0; c < n_chunks; c++) {
        const uint4 nxt = (c + 1 < n_chunks) ? src.load(c + 1) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (c * 8 + j >= nb) break;
            const uint32_t rec = chunk_rec(cur, j);
            const uint32_t bin = rec & 1, pos = (rec >> 1) & 0x7f, neg = (rec >> 8) & 0x7f;
            const uint32_t total = pos + neg;                    // recode.cpp:825
            if (total == 0) { st = AVR_SLICE_BAD_RECORD; c = n_chunks; break; }
            const uint64_t r1 = (e.range / total) * pos;         // recode.cpp:826
            const uint64_t r0 = e.range - r1;                    // arithmetic_code.h:108
            e.low += bin ? r0 : 0;
            e.range = bin ? r1 : r0;
            if (e.range < (uint64_t(1) << 51)) {                 // min_range, arithmetic_code.h:61-62,115
                if (e.range == 0) { st = AVR_SLICE_ZERO_PROB; c = n_chunks; break; }   // :116-118
                while (e.range < (uint64_t(1) << 55)) e.emit_digit();                  // :120-122
            }
        }
        cur = nxt;
    }
    if (st == AVR_SLICE_OK) e.finish();                          // recode.cpp:1100
    e.w.flush();
    if (st == AVR_SLICE_OK && e.w.n > cap) st = AVR_SLICE_OVERFLOW;
    out_len[slice] = e.w.n;
    status[slice] = st;
}

// ------------------------------------------------------------------ pack: slice-major -> tiles

// One workgroup (64 lanes) per tile.  Lane l copies the chunks of its slice; a wave-wide
// store instruction writes 1 KiB contiguous.  Chunks past a short slice's end are zeroed.
__global__ __launch_bounds__(64) void k_pack_tiles(
    const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins, const uint32_t *order,
    uint32_t n_slices, const uint64_t *tile_off, uint4 *tiles) {
    const uint32_t lane = threadIdx.x, t = blockIdx.x;
    const uint32_t g = t * 64 + lane;
    const bool active = g < n_slices;
    const uint32_t slice = active ? (order ? order[g] : g) : 0;
    const uint32_t nb = active ? n_bins[slice] : 0;
    const uint32_t my_chunks = (nb + 7) >> 3;
    const uint32_t tile_chunks = uint32_t((tile_off[t + 1] - tile_off[t]) >> 6);
    const uint4 *src = reinterpret_cast<const uint4 *>(recs + (active ? rec_off[slice] : 0));
    uint4 *dst = tiles + tile_off[t] + lane;
    for (uint32_t c = 0; c < tile_chunks; c++) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < my_chunks) {
            v = src[c];
            const uint32_t valid = nb - c * 8;                   // records valid in this chunk
            if (valid < 8) {                                     // zero the padding records
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (uint32_t k = 0; k < 4; k++) {
                    if (2 * k >= valid) w[k] = 0;
                    else if (2 * k + 1 >= valid) w[k] &= 0xffffu;
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        dst[size_t(c) * 64] = v;
    }
}

// ------------------------------------------------------------------ compact: per-slice regions -> dense

// One workgroup per slice; the copy is contiguous on both sides.  dense_off[i+1]-dense_off[i]
// is min(out_len[i], capacity) as computed by the host.
__global__ __launch_bounds__(64) void k_compact(const uint8_t *out, const uint64_t *out_off, const uint64_t *dense_off,
                                                uint32_t n_slices, uint8_t *dense) {
    const uint32_t i = blockIdx.x;
    if (i >= n_slices) return;
    const uint8_t *src = out + out_off[i];
    uint8_t *dst = dense + dense_off[i];
    const uint32_t len = uint32_t(dense_off[i + 1] - dense_off[i]);
    for (uint32_t k = threadIdx.x; k < len; k += 64) dst[k] = src[k];
}

// ------------------------------------------------------------------ synthetic streams

struct TileRecordSink {               // gathers 8 records, stores one 16-byte chunk
    uint4 *dst;                       // chunk 0 of this lane
    uint32_t n;
    uint32_t w[4];
    __device__ explicit TileRecordSink(uint4 *d) : dst(d), n(0) { w[0] = w[1] = w[2] = w[3] = 0; }
    __device__ void put_record(uint16_t rec) {
        const uint32_t j = n & 7;
        w[j >> 1] |= uint32_t(rec) << ((j & 1) * 16);
        n++;
        if ((n & 7) == 0) {
            dst[size_t((n >> 3) - 1) * 64] = make_uint4(w[0], w[1], w[2], w[3]);
            w[0] = w[1] = w[2] = w[3] = 0;
        }
    }
    __device__ void flush(uint32_t tile_chunks) {
        uint32_t c = n >> 3;
        if (n & 7) { dst[size_t(c) * 64] = make_uint4(w[0], w[1], w[2], w[3]); c++; }
        for (; c < tile_chunks; c++) dst[size_t(c) * 64] = make_uint4(0, 0, 0, 0);
    }
};

__global__ __launch_bounds__(64) void k_synth_count(
    int workload, uint32_t scale, uint64_t seed, uint64_t first_slice, int kind,
    uint32_t n_slices, uint32_t *n_bins) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_slices) return;
    CountSink cs;
    CabacSink<CountSink> sink(cs);       // K2 streams have the same length as K1 streams
    synth_slice(workload, scale, seed, first_slice + i, sink);
    (void)kind;
    n_bins[i] = cs.n;
}

__global__ __launch_bounds__(64) void k_synth_tiles(
    int workload, uint32_t scale, uint64_t seed, uint64_t first_slice, int kind,
    uint32_t n_slices, const uint32_t *order, const uint64_t *tile_off, uint4 *tiles,
    uint8_t *init_states, uint32_t n_states) {
    const uint32_t lane = threadIdx.x, t = blockIdx.x;
    const uint32_t g = t * 64 + lane;
    const uint32_t tile_chunks = uint32_t((tile_off[t + 1] - tile_off[t]) >> 6);
    TileRecordSink rs(tiles + tile_off[t] + lane);
    if (g < n_slices) {
        const uint32_t slice = order ? order[g] : g;
        if (kind == AVR_KIND_CABAC) {
            CabacSink<TileRecordSink> sink(rs);
            synth_slice(workload, scale, seed, first_slice + slice, sink);
            if (init_states)
                for (uint32_t c = 0; c < n_states; c++)
                    init_states[size_t(slice) * n_states + c] = synth_init_state(c, seed, first_slice + slice);
        } else {
            ModelSink<TileRecordSink> sink(rs);
            synth_slice(workload, scale, seed, first_slice + slice, sink);
        }
    }
    rs.flush(tile_chunks);
}

// ------------------------------------------------------------------ launchers

static inline uint32_t cabac_lds_bytes(uint32_t n_states) {
    const uint32_t rows = (n_states + 3) / 4;
    return 1024 + 64 * 4 * (rows ? rows : 1);          // row 0 is also the dummy read of non-context bins
}

hipError_t launch_cabac_encode(bool tiled, hipStream_t s, const void *recs, const uint64_t *off,
                               const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                               const uint8_t *init_states, uint32_t n_states, uint8_t *out,
                               const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                               uint8_t *final_states) {
    if (n_slices == 0) return hipSuccess;
    const uint32_t lds = cabac_lds_bytes(n_states);
    const dim3 grid((n_slices + 63) / 64), block(64);
    auto kern = tiled ? k_cabac_encode<true> : k_cabac_encode<false>;
    if (lds > 64 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (err != hipSuccess) return err;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, recs, off, n_bins, order, n_slices, init_states,
                       n_states, out, out_off, out_len, status, final_states);
    return hipGetLastError();
}

hipError_t launch_range_encode(bool tiled, hipStream_t s, const void *recs, const uint64_t *off,
                               const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                               uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                               int32_t *status) {
    if (n_slices == 0) return hipSuccess;
    const dim3 grid((n_slices + 63) / 64), block(64);
    if (tiled)
        hipLaunchKernelGGL(k_range_encode<true>, grid, block, 0, s, recs, off, n_bins, order, n_slices,
                           out, out_off, out_len, status);
    else
        hipLaunchKernelGGL(k_range_encode<false>, grid, block, 0, s, recs, off, n_bins, order, n_slices,
                           out, out_off, out_len, status);
    return hipGetLastError();
}

hipError_t launch_pack_tiles(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off,
                             const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                             const uint64_t *tile_off, void *tiles) {
    if (n_slices == 0) return hipSuccess;
    const dim3 grid((n_slices + 63) / 64), block(64);
    hipLaunchKernelGGL(k_pack_tiles, grid, block, 0, s, recs, rec_off, n_bins, order, n_slices, tile_off,
                       reinterpret_cast<uint4 *>(tiles));
    return hipGetLastError();
}

hipError_t launch_compact(hipStream_t s, const uint8_t *out, const uint64_t *out_off, const uint32_t *out_len,
                          const uint64_t *dense_off, uint32_t n_slices, uint8_t *dense) {
    (void)out_len;
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_compact, dim3(n_slices), dim3(64), 0, s, out, out_off, dense_off, n_slices, dense);
    return hipGetLastError();
}

hipError_t launch_synth_count(hipStream_t s, int workload, uint32_t scale, uint64_t seed,
                              uint64_t first_slice, int kind, uint32_t n_slices, uint32_t *n_bins) {
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_count, dim3((n_slices + 63) / 64), dim3(64), 0, s, workload, scale, seed,
                       first_slice, kind, n_slices, n_bins);
    return hipGetLastError();
}

hipError_t launch_synth_tiles(hipStream_t s, int workload, uint32_t scale, uint64_t seed,
                              uint64_t first_slice, int kind, uint32_t n_slices, const uint32_t *order,
                              const uint64_t *tile_off, void *tiles, uint8_t *init_states,
                              uint32_t n_states) {
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_tiles, dim3((n_slices + 63) / 64), dim3(64), 0, s, workload, scale, seed,
                       first_slice, kind, n_slices, order, tile_off, reinterpret_cast<uint4 *>(tiles),
                       init_states, n_states);
    return hipGetLastError();
}

}  // namespace avr
