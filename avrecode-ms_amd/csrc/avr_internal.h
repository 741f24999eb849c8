// Internal declarations shared by avr_kernels.hip (device code + launchers) and avr_api.cpp
// (the C ABI).  Not installed; the public surface is include/avrecode_ms_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/avrecode_ms_amd.h"

// No-op records: what the packer and the generators pad a slice's last chunk (and a tile's
// shorter lanes) with, so the encode kernels never test a record index against n_bins.
//   K1: selector 1026, bin 0 -> r1 = 0, symbol 0: low, range and states unchanged
//   K2: pos = neg = 0 (never valid in a real record) -> skipped
#define AVR_NOP_CABAC2  (AVR_NOP_CABAC | (AVR_NOP_CABAC << 16))
// transient per-slice status: K1p declined the slice, k_cabac_encode codes it in the same call
// (never leaves a call: k_cabac_encode / k_cabac_encode_codes replace it with the slice's real status)
#define AVR_SLICE_RETRY_SERIAL 100
// transient, inside the intra-slice parallel path only: the slice has a bin in a context the sampled census of the
// batch missed (it takes the second, fully counted pass); the slice is finished and must be left alone by that pass
#define AVR_SLICE_RETRY_CENSUS 101
#define AVR_SLICE_DONE         102

namespace avr {

// Run-time switches of the library.
//   * Environment, read ONCE per process (env()), documented in include/avrecode_ms_amd.h; none of them can change a
//     coded byte -- they choose between mappings that produce the same bytes:
//       AVR_K1_PATH=serial|chunked   force one lane per slice / the intra-slice parallel kernels in the batch API
//       AVR_NO_DENSE=1               one-lane-per-slice K1 without the renumbering onto the contexts the batch uses
//       AVR_BATCH_NO_HINT=1          avr_batch_submit always asks the device for the context count (and waits)
//   * Test hooks, compiled in only with -DAVR_TEST_HOOKS (libavrecode_hip_hooks.so, which tests/ load; the product
//     library has neither the setter nor any code that reads them): force the hand-over paths that real streams
//     take once in a blue moon, so that every run of the tests proves them.  All bit-exact as well.
struct Env { int k1_path; bool no_dense, no_hint; };             // k1_path: 0 auto, 1 serial, 2 chunked
const Env &env();
struct TestHooks {
    uint32_t k1p_force_retry_every;  // k_k1p_d hands every n-th slice to the serial kernel (0 = off)
    uint32_t census_stride;          // sampling stride of both census kernels (0 = the built-in 16)
    uint32_t chain_lanes;            // lanes per wave of k_k1p_ctxchain (0 = by batch shape)
    uint32_t k1_form_ref;           // one-lane-per-slice K1 as cabac_code.h writes the coder (CabacLane) instead of the shipped normalised form
    uint32_t k1_emit_lds;            // one-lane-per-slice K1 with its digits staged in LDS and stored in 16-byte rows: 1 = on the reference-style form (CabacLaneS), 2 = on the shipped one (CabacLaneNS)
    uint32_t k1_path, no_dense, no_hint;   // the environment switches above, settable per test (non-zero wins over env())
    uint32_t chain_segments;         // K1p: the context chains in segments whatever the batch's size
    uint32_t chain_whole;            // K1p: every context chain start to end (k_k1p_ctxchain alone), no segments
    uint32_t chain_force_redo;       // K1p: the segmented chains hand every n-th (slice, context) pair to the whole-slice walk
    uint32_t k2p_seg_len;            // chunks per segment of K2p's overlapped passes (0 = by batch shape): short slices through many segments
    uint32_t local_waves;            // K1p: waves to a workgroup of k_k1p_local (0 = as many waves to a CU as its LDS takes)
    uint32_t chain_nsegs;            // K1p: segments the context chains are cut in (0 = as many as keep the launch in one round of workgroups)
    uint32_t k1_fwd;                 // one-lane-per-slice K1: four state bytes read ahead with forwarding (CabacLaneN::bin4: a measured variant), not one per bin
    uint32_t k1_words8;              // one-lane-per-slice K1: the output in 8-byte stores (rounds 1-3), not 16-byte ones
    uint32_t k1_waves;               // one-lane-per-slice K1: waves to a workgroup (0 = the built-in count)
    uint32_t k2p_wave;               // K2p pass 1: 1 = a wave per slice, 2 = a lane per slice, 3 = a lane per slice and a wave each for the longest (0 = by slice count)
};
#ifdef AVR_TEST_HOOKS
TestHooks &test_hooks();
#else
inline const TestHooks &test_hooks() { static const TestHooks none{}; return none; }
#endif
inline int k1_path() { return test_hooks().k1_path ? int(test_hooks().k1_path) : env().k1_path; }
inline bool no_dense() { return test_hooks().no_dense || env().no_dense; }
inline bool no_hint() { return test_hooks().no_hint || env().no_hint; }

// How many LDS rows the renumbered contexts of a batch need is known on the device after the census; the launches
// that follow are sized by it, which costs the host one 4-byte round trip per call.  A caller that has a good guess
// (the batch API: the count of its previous batch) passes it here: the launches are sized by `rows`, nothing waits,
// and the true count is copied to *host_count (pinned) behind the kernels for the caller to check afterwards:
//   * the one-lane-per-slice kernel is exact whatever the guess: contexts beyond `rows` are handed back like
//     contexts the sampled census missed;
//   * the intra-slice parallel kernels are exact iff *host_count <= rows: otherwise the caller runs the call again
//     without a hint (they never fault: a dense id >= rows is treated as "no context").
//   * with a guess the intra-slice parallel path also leaves its second pass (slices with a context the sampled
//     census missed) to the caller: *host_retry (pinned) gets the number of such slices behind the kernels, and a
//     caller that finds it non-zero calls launch_k1p_retry with the arguments of launch_k1p.
//   * sharing: how many calls like this one are in flight on the device at once (the parts of avr_cabac_encode_chunked_device_parts):
//     the context chains are cut into as many segments as keep ALL of them in one round of workgroups (0 = 1).
struct DenseHint { uint32_t rows; uint32_t *host_count; uint32_t *host_retry; uint32_t sharing = 0; };

// what the library keeps per (device, stream) -- the renumbering's scratch, K2p's second stream and events -- released: call before the stream is destroyed
void forget_stream(hipStream_t s);
void forget_part_streams(hipStream_t s);                      // avr_api.cpp: the streams of avr_cabac_encode_chunked_device_parts kept for s

hipError_t launch_cabac_encode(bool tiled, hipStream_t s, const void *recs, const uint64_t *off,
                               const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                               const uint8_t *init_states, uint32_t n_states, uint8_t *out,
                               const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                               uint8_t *final_states, int32_t want_status = AVR_SLICE_OK, bool dense = true,
                               const DenseHint *hint = nullptr);
// used (1024 bits) -> table[caller's context number] = dense id (0xffff: unused), index[dense id] = caller's number, *n_dense
hipError_t launch_densemap(hipStream_t s, const uint32_t *used, uint16_t *table, uint16_t *index, uint32_t *n_dense);
hipError_t launch_range_encode(bool tiled, hipStream_t s, const void *recs, const uint64_t *off,
                               const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                               uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                               int32_t *status);
hipError_t launch_pack_tiles(hipStream_t s, int kind, uint32_t n_states, const uint16_t *recs,
                             const uint64_t *rec_off, const uint32_t *n_bins, const uint32_t *order,
                             uint32_t n_slices, const uint64_t *tile_off, void *tiles, int32_t *status);
hipError_t launch_synth_count(hipStream_t s, int workload, uint32_t scale, uint64_t seed,
                              uint64_t first_slice, int kind, uint32_t n_slices, uint32_t *n_bins);
hipError_t launch_synth_tiles(hipStream_t s, int workload, uint32_t scale, uint64_t seed,
                              uint64_t first_slice, int kind, uint32_t n_slices, const uint32_t *order,
                              const uint64_t *tile_off, void *tiles, uint8_t *init_states,
                              uint32_t n_states);

// one-byte K1 records (AVR_KIND_CABAC8) -> the two-byte records of the kernels; `total` = records of the batch (a multiple of 8)
hipError_t launch_expand_records8(hipStream_t s, const uint8_t *in, const uint64_t *rec_off, const uint32_t *n_bins, uint32_t n_slices,
                                  uint32_t n_states, uint64_t total, uint16_t *out);
hipError_t launch_context_census(hipStream_t s, const uint16_t *recs, uint64_t n, uint32_t *bitmap);
hipError_t launch_context_remap(hipStream_t s, uint16_t *recs, uint64_t n, const uint16_t *table);
hipError_t launch_states_permute(hipStream_t s, const uint8_t *src, uint32_t n_src, uint8_t *dst, uint32_t n_dst,
                                 const uint16_t *index, uint32_t n_index, uint64_t n_slices, int scatter);
hipError_t launch_compact(hipStream_t s, const uint8_t *out, const uint64_t *out_off, const uint32_t *out_len,
                          const uint64_t *dense_off, uint32_t n_slices, uint8_t *dense);

size_t k1p_workspace_bytes(size_t n_slices, uint32_t n_states, const avr_chunk_plan *pl);
hipError_t launch_k1p(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                      uint32_t n_slices, const uint8_t *init_states, uint32_t n_states, const avr_chunk_plan *pl,
                      void *workspace, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                      uint8_t *final_states, const DenseHint *hint = nullptr);
hipError_t launch_k1p_retry(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                            uint32_t n_slices, const uint8_t *init_states, uint32_t n_states, const avr_chunk_plan *pl,
                            void *workspace, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                            uint8_t *final_states);
size_t k1p_resolve_workspace_bytes(size_t n_slices, uint32_t n_states, const avr_chunk_plan *pl);
hipError_t launch_k1p_resolve(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                              uint32_t n_slices, const uint8_t *init_states, uint32_t n_states, const avr_chunk_plan *pl,
                              void *workspace, uint8_t *codes, int32_t *status, uint8_t *final_states);
size_t k1p_code_workspace_bytes(size_t n_slices, const avr_chunk_plan *pl);
hipError_t launch_k1p_code(hipStream_t s, const uint8_t *codes, const uint32_t *n_bins, uint32_t n_slices,
                           const avr_chunk_plan *pl, void *workspace, uint8_t *out, const uint64_t *out_off,
                           uint32_t *out_len, int32_t *status);
hipError_t launch_cabac_encode_codes(hipStream_t s, const uint8_t *codes, const uint64_t *res_off, const uint32_t *n_bins,
                                     const uint32_t *order, uint32_t n_slices, uint8_t *out, const uint64_t *out_off,
                                     uint32_t *out_len, int32_t *status, int32_t want_status = AVR_SLICE_OK);
// K2 for few, long slices (avr_k2p.hip): range recurrence per slice, coding per chunk into byte sums, carries + finish
size_t k2p_workspace_bytes(size_t n_slices, uint32_t total_chunks, uint64_t out_total);
hipError_t launch_k2p(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins, uint32_t n_slices,
                      const uint32_t *chunk_base, const uint32_t *chunk_slice, uint32_t total_chunks, uint64_t out_total,
                      void *workspace, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status);
hipError_t launch_synth_slices(hipStream_t s, int workload, uint32_t scale, uint64_t seed, uint64_t first_slice,
                               int kind, uint32_t n_slices, const uint64_t *rec_off, uint16_t *recs,
                               uint8_t *init_states, uint32_t n_states);

}  // namespace avr
