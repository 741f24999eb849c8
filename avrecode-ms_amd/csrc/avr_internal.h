// Internal declarations shared by avr_kernels.hip (device code + launchers) and avr_api.cpp
// (the C ABI).  Not installed; the public surface is include/avrecode_ms_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/avrecode_ms_amd.h"

namespace avr {

hipError_t launch_cabac_encode(bool tiled, hipStream_t s, const void *recs, const uint64_t *off,
                               const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                               const uint8_t *init_states, uint32_t n_states, uint8_t *out,
                               const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                               uint8_t *final_states);
hipError_t launch_range_encode(bool tiled, hipStream_t s, const void *recs, const uint64_t *off,
                               const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                               uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                               int32_t *status);
hipError_t launch_pack_tiles(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off,
                             const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                             const uint64_t *tile_off, void *tiles);
hipError_t launch_synth_count(hipStream_t s, int workload, uint32_t scale, uint64_t seed,
                              uint64_t first_slice, int kind, uint32_t n_slices, uint32_t *n_bins);
hipError_t launch_synth_tiles(hipStream_t s, int workload, uint32_t scale, uint64_t seed,
                              uint64_t first_slice, int kind, uint32_t n_slices, const uint32_t *order,
                              const uint64_t *tile_off, void *tiles, uint8_t *init_states,
                              uint32_t n_states);

hipError_t launch_compact(hipStream_t s, const uint8_t *out, const uint64_t *out_off, const uint32_t *out_len,
                          const uint64_t *dense_off, uint32_t n_slices, uint8_t *dense);

}  // namespace avr
