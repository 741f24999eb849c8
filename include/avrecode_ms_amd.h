/*
 * avrecode_ms_amd.h -- C ABI of the MI355X-native arithmetic re-encode path.
 *
 * Shared library: avrecode-ms_amd/libavrecode_hip.so (built by __graft_entry__.build()).
 * Plain C: pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * What this replaces in the reference (pbluc/avrecode-ms, paths relative to
 * /root/reference):
 *
 *   The reference codes every CABAC bin inline, inside the libavcodec hook
 *   callbacks (recode.cpp:149-171 -> Driver::cabac_decoder::get / get_bypass /
 *   get_terminate):
 *     compress    recode.cpp:1182-1199 -> h264_symbol::execute :1075-1103
 *                 -> recoded_code::encoder::put (arithmetic_code.h:106-126)
 *     decompress  recode.cpp:1442-1481
 *                 -> cabac::encoder::put / put_bypass / put_terminate (cabac_code.h:33-67)
 *   Nothing consumes the coded bytes before the end of the run
 *   (recode.cpp:1131 SerializeAsString, :1352-1363 final loop), so the build's
 *   hook adapter RECORDS one 16-bit record per bin and hands whole batches of
 *   independent slices to the device here.  The entry points are what a cgo /
 *   JNI / ctypes binding -- or the reference's own C++ -- would bind; the
 *   reference-side patch is shown in INTEGRATION.md.
 *
 * Record formats
 *   CABAC record (K1, decompress direction), uint16_t:
 *       bit 0      bin value
 *       bits 1..11 selector: 0..1023  context index = offset of the `uint8_t *state`
 *                                     handed to get() from the slice's first state
 *                                     (recode.cpp:156, context identity is the address, :325)
 *                            1024     bypass     (get_bypass,    recode.cpp:1458-1467)
 *                            1025     terminate  (get_terminate, recode.cpp:1469-1481)
 *   range record (K2, compress direction), uint16_t:
 *       bit 0      bin value
 *       bits 1..7  pos, bits 8..14 neg : the {pos,neg} estimator of the bin's model key at
 *                  the moment it is coded (recode.cpp:823-827, 1064); p(1) = (range/(pos+neg))*pos
 *
 * Environment.  The library reads three variables, once per process; NONE of them can change a coded byte -- they pick
 * between mappings that produce the same bytes (tests/ run all of them against the oracle):
 *   AVR_K1_PATH=serial|chunked   batch API: force one lane per slice / the intra-slice parallel kernels (default: by batch shape)
 *   AVR_NO_DENSE=1               one-lane-per-slice K1 keeps the caller's context numbering (no renumbering onto the contexts in use)
 *   AVR_BATCH_NO_HINT=1          avr_batch_submit always asks the device for the batch's context count (and waits for it)
 * There is no switch that alters output.  The switches tests use to force rare hand-over paths exist only in a separate
 * build of the same sources with -DAVR_TEST_HOOKS (libavrecode_hip_hooks.so, avr_test_hook_set); this library has neither
 * the setter nor the code that reads them.
 *
 * Threading: one avr_batch per host thread; calls on different batches are
 * independent.  Errors: functions return AVR_OK (0) or a negative AVR_ERR_*;
 * avr_last_error() gives the message for the calling thread.  The reference
 * throws C++ exceptions through its C callbacks instead (recode.cpp:43,71,99,169,
 * arithmetic_code.h:117); the host wrapper re-throws from these codes.
 */
#ifndef AVRECODE_MS_AMD_H
#define AVRECODE_MS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVR_OK              0
#define AVR_ERR_INVALID    -1   /* bad argument / call order */
#define AVR_ERR_NO_DEVICE  -2   /* no HIP device: there is NO CPU fallback */
#define AVR_ERR_HIP        -3   /* a HIP runtime call failed */
#define AVR_ERR_NOMEM      -4
#define AVR_ERR_CAPACITY   -5   /* more slices / bins than the batch was created for */

/* per-slice status written by the kernels */
#define AVR_SLICE_OK          0
#define AVR_SLICE_ZERO_PROB   1  /* arithmetic_code.h:116-118 "emitted a zero-probability symbol" */
#define AVR_SLICE_OVERFLOW    2  /* output region too small (never with the batch API's sizing) */
#define AVR_SLICE_BAD_RECORD  3  /* selector out of range, or a bin after put_terminate(1) */

#define AVR_SEL_BYPASS     1024
#define AVR_SEL_TERMINATE  1025
#define AVR_MAX_STATES     1024  /* size of libavcodec's per-slice cabac_state[] */

/* No-op records.  Device layouts keep every slice a whole number of 8-record (16-byte)
 * chunks; the records between a slice's last bin and the chunk end are no-ops, so the encode
 * kernels never compare a record index with n_bins.  K1: selector 1026 with bin 0 (probability
 * range 0, symbol 0: low, range and states unchanged).  K2: pos = neg = 0 (never a real
 * record).  The batch API and avr_pack_tiles_device write them; callers of the slice-major
 * device entry points must. */
#define AVR_NOP_CABAC      (1026 << 1)
#define AVR_NOP_RANGE      0

#define AVR_KIND_CABAC 0         /* K1: cabac::encoder           (cabac_code.h:26-82)   */
#define AVR_KIND_RANGE 1         /* K2: recoded_code::encoder    (recode.cpp:322-323)   */
#define AVR_KIND_CABAC_CODES 2   /* K1 from resolved codes: one byte per bin, see avr_batch_add_slice_codes */
#define AVR_KIND_CABAC8 3        /* K1 from ONE-BYTE records (bin, dense selector), see avr_batch_add_slice_cabac8 */

/* One-byte K1 records (AVR_KIND_CABAC8): bit 0 = the bin, bits 1..7 = a dense selector -- 0 .. 125 the slice's contexts in the
 * order the recorder met them (its ids are dense by first appearance already: INTEGRATION.md), AVR_SEL8_BYPASS, AVR_SEL8_TERMINATE.
 * What the record stands for is what the two-byte record stands for (recode.cpp:1442-1481); a stream with more than 126 contexts
 * keeps the two-byte form. */
#define AVR_SEL8_BYPASS     126
#define AVR_SEL8_TERMINATE  127
#define AVR_MAX_STATES8     126

const char *avr_last_error(void);
const char *avr_version(void);
int  avr_device_count(void);                 /* 0 when no GPU is visible */

/* The CABAC tables in the layout cabac_code.h:11-12 indexes (512 and 256 bytes). */
const uint8_t *avr_cabac_lps_range_table(void);
const uint8_t *avr_cabac_mlps_state_table(void);

/* ------------------------------------------------------------------ batch API (host memory)
 * Replaces the per-bin encoder calls of Driver::cabac_decoder (see above).  Usage:
 *   b = avr_batch_create(dev, max_slices, max_bins);
 *   per slice:  avr_batch_add_slice_cabac(...) | avr_batch_add_slice_range(...)
 *   avr_batch_run(b);                      // H2D, pack, encode kernel, D2H
 *   per slice:  avr_batch_get(b, i, &bytes, &len, &status);
 * A batch holds slices of one kind only.  Returned pointers stay valid until the next
 * avr_batch_reset / avr_batch_destroy.  K1 output is the raw encoder output: the caller
 * applies recode.cpp:1508-1512 (drop a trailing 0x80) and :1354-1360 (tail patch),
 * see avr_drop_stop_byte / avr_tail_patch. */
typedef struct avr_batch avr_batch;

avr_batch *avr_batch_create(int device, size_t max_slices, size_t max_bins);
void       avr_batch_destroy(avr_batch *b);
int        avr_batch_reset(avr_batch *b);

/* init_states: n_states bytes (2*pStateIdx+valMPS), the slice's cabac_state[] as it was
 * when the slice's first bin was requested; n_states <= AVR_MAX_STATES and all slices of a
 * batch use the same n_states.  Returns the slice index (>= 0) or an error (< 0). */
int avr_batch_add_slice_cabac(avr_batch *b, const uint16_t *recs, size_t n,
                              const uint8_t *init_states, size_t n_states);
int avr_batch_add_slice_range(avr_batch *b, const uint16_t *recs, size_t n);
/* K1 from RESOLVED CODES: one byte per bin made with AVR_CODE_CONTEXT(state, bin) -- `state` being *state
 * as it is when the bin is requested, which the adapter holds anyway (it updates it, cabac_code.h:43-47) --
 * AVR_CODE_BYPASS(bin) or AVR_CODE_TERMINATE(bin): the (symbol, *state) pairs cabac::encoder::put takes
 * (cabac_code.h:33).  Half the bytes of avr_batch_add_slice_cabac over PCIe, no state arrays, and the
 * context-state resolution on the GPU is skipped.  put_terminate(1), if present, must be the last bin.
 * A batch of few, long slices is coded by the intra-slice parallel kernels (K1p phases B-D), a batch of
 * many short ones by one lane per slice (k_cabac_encode_codes); either way every slice is coded. */
int avr_batch_add_slice_codes(avr_batch *b, const uint8_t *codes, size_t n);
/* K1 from ONE-BYTE records (AVR_KIND_CABAC8, above): the same slice as avr_batch_add_slice_cabac would take, in half the bytes
 * over PCIe -- which is what bounds the batch API end to end (2 B per bin against 0.1 B of output).  init_states: n_states <=
 * AVR_MAX_STATES8 bytes, indexed by the dense selector.  On the device the records are widened into the two-byte form
 * (k_expand_records8) and take the path of avr_batch_add_slice_cabac from there: same kernels, same bytes, same statuses -- a
 * selector >= n_states that is neither bypass nor terminate comes back as AVR_SLICE_BAD_RECORD. */
int avr_batch_add_slice_cabac8(avr_batch *b, const uint8_t *recs8, size_t n,
                               const uint8_t *init_states, size_t n_states);
/* Zero-copy form of the four calls above: room for a slice of n elements (uint16_t records for AVR_KIND_CABAC /
 * AVR_KIND_RANGE, uint8_t codes for AVR_KIND_CABAC_CODES, uint8_t records for AVR_KIND_CABAC8) in the batch's pinned staging buffer, which the H2D copy
 * reads directly; the caller writes exactly n elements to *buffer before avr_batch_submit / avr_batch_run (the padding
 * after them is already in place).  init_states / n_states as for avr_batch_add_slice_cabac (copied now), ignored
 * for the other kinds.  A recorder that appends here saves one pass over its records.  Returns the slice index. */
int avr_batch_reserve_slice(avr_batch *b, int kind, size_t n, const uint8_t *init_states, size_t n_states, void **buffer);

/* avr_batch_run = avr_batch_submit + avr_batch_wait.
 * avr_batch_submit enqueues the whole run on the batch's own stream -- H2D from pinned memory, the kernels, the
 * lengths on their way back -- and returns without waiting for the device (the first CABAC-record run of a batch
 * object waits once for a 4-byte context count; later runs are sized by that count and checked in avr_batch_wait).
 * avr_batch_wait blocks until the results are in host memory.  Between the two calls the batch must not be
 * touched (add / reset / get fail with AVR_ERR_INVALID).  Two or three batch objects used in turn keep the copy
 * engines and the kernels of consecutive batches overlapped:
 *     submit(b[0]);  for (i = 1; ; i++) { fill(b[i % 2]); submit(b[i % 2]); wait(b[(i - 1) % 2]); consume; reset; }
 * After avr_batch_wait a batch may be submitted again as it is (same slices, coded anew). */
int avr_batch_submit(avr_batch *b);
int avr_batch_wait(avr_batch *b);
int avr_batch_run(avr_batch *b);

int avr_batch_get(avr_batch *b, size_t slice, const uint8_t **bytes, size_t *len, int *status);
/* K1 only: the slice's state bytes after its last bin (what cabac_code.h:43-47 leaves in *state). */
int avr_batch_get_states(avr_batch *b, size_t slice, const uint8_t **states, size_t *n_states);
/* How the last run went: [0] 1 = intra-slice parallel kernels, 0 = one lane per slice; [1] context rows the kernels
 * were sized by from the previous run's count (0: the run asked the device and waited); [2] contexts the batch uses
 * (as the sampled census saw them); [3] bit 0 = avr_batch_wait found the guess too small and ran the batch
 * again, bit 1 = it ran the second pass of the intra-slice parallel path (slices with a bin in a context the sampled
 * census missed).  CABAC-record batches; zeros otherwise. */
int avr_batch_run_info(avr_batch *b, uint32_t info[4]);
/* milliseconds of the last run: [0] H2D, [1] pack kernel, [2] encode kernel, [3] D2H */
int avr_batch_timings(avr_batch *b, float ms[4]);

/* ------------------------------------------------------------------ one batch over several GPUs
 * Slices are independent (one coder object each in the reference, recode.cpp:1270, :1525), so a batch shards
 * with no exchange between devices: avr_multi_run sorts the slices by bin count, hands them out longest first to
 * the device with the least work so far (greedy LPT), runs one avr_batch per device from a host thread of its
 * own (own stream, own pinned staging) and the getters return results by the caller's slice index.  `devices`
 * may name a device more than once (it then gets that many independent sub-batches).  Same record formats,
 * same rules (one kind per batch, one n_states) and same errors as the single-device batch. */
typedef struct avr_multi avr_multi;
avr_multi *avr_multi_create(const int *devices, size_t n_devices, size_t max_slices, size_t max_bins);
void       avr_multi_destroy(avr_multi *m);
int avr_multi_add_slice_cabac(avr_multi *m, const uint16_t *recs, size_t n, const uint8_t *init_states, size_t n_states);
int avr_multi_add_slice_range(avr_multi *m, const uint16_t *recs, size_t n);
int avr_multi_add_slice_codes(avr_multi *m, const uint8_t *codes, size_t n);
int avr_multi_run(avr_multi *m);
int avr_multi_get(avr_multi *m, size_t slice, const uint8_t **bytes, size_t *len, int *status);
/* which entry of `devices` coded the slice, and the bins each entry was given (n_devices values): for tests and reports */
int avr_multi_placement(avr_multi *m, size_t slice);
int avr_multi_load(avr_multi *m, uint64_t *bins_per_device);

/* ------------------------------------------------------------------ device-resident API
 * All pointers below are DEVICE pointers on `device`; `stream` is a hipStream_t (NULL = the
 * null stream).  Calls enqueue work on `stream` and return, with ONE exception: the K1 entry points that take
 * (bin, selector) records -- avr_cabac_encode_tiles_device / _slices_device / _chunked_device and
 * avr_cabac_resolve_device -- size their launches by the number of contexts the batch uses, which the device counts:
 * they wait on `stream` once for that 4-byte count (the chunked forms a second time, for the 4-byte count of slices that
 * need their second pass), i.e. they block the calling thread until the stream has drained up to their census kernel.
 * A caller that must not block uses the batch API (avr_batch_submit sizes the launches by the previous batch's count and
 * checks afterwards), the same scheme on its own buffers (avr_cabac_encode_tiles_device_hinted / _chunked_device_hinted below: the
 * caller passes the count an earlier call reported and looks at what this one reports when it next synchronises), or resolved
 * codes (avr_cabac_encode_resolved_device / _codes_device and every K2 entry: no wait).
 * These are what the batch API is made of and what bench.py times with inputs already resident in HBM (its K1 steps: the hinted calls).
 * ONE THREAD PER STREAM: the library keeps a few kilobytes of scratch (and, for the chunked K2, a second stream with its
 * events) per (device, stream); a call's kernels find them there, so two host threads must not enqueue on the same stream
 * at the same time.  What is kept for a batch's own stream is released by avr_batch_destroy.
 *
 * Slice-major layout: slice i's records are recs[rec_off[i] .. rec_off[i] + n_bins[i]);
 * rec_off[] entries are multiples of 8 records (16 bytes).
 *
 * Wave-interleaved tile layout (what the encode kernels read): slices are taken 64 at a
 * time in `order` (order[g] = slice handled by lane g%64 of tile g/64).  A tile holds
 * max-over-its-lanes ceil(n_bins/8) chunks; chunk c of lane l is the 16 bytes at
 *   tiles + (tile_off[t] + c*64 + l) * 16
 * so one wave-wide load instruction reads 1 KiB contiguous.  tile_off has n_tiles+1 entries
 * in units of 16-byte chunks. */
/* Also validates every record (it is the one place each record is read exactly once): a K1
 * selector that is neither < n_states nor bypass/terminate, or a K2 record with pos+neg = 0 or
 * bit 15 set, sets status[slice] = AVR_SLICE_BAD_RECORD.  status must be zero-filled before. */
int avr_pack_tiles_device(int device, void *stream, int kind, size_t n_states,
                          const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                          const uint32_t *order, size_t n_slices,
                          const uint64_t *tile_off, void *tiles, int32_t *status);

/* K1.  init_states: n_slices*n_states bytes indexed by slice; out_off: n_slices+1 byte offsets
 * into out, each a multiple of 8; out_len/status indexed by slice; final_states may be NULL.
 * status is in/out: a slice whose status is already non-zero (set by the packer) is skipped
 * with out_len 0; otherwise the kernel writes AVR_SLICE_*.  Records are trusted to be valid
 * (packer / generator output); a selector that is not a context of the slice, bypass or
 * terminate is treated as a no-op. */
int avr_cabac_encode_tiles_device(int device, void *stream,
                                  const void *tiles, const uint64_t *tile_off,
                                  const uint32_t *n_bins, const uint32_t *order, size_t n_slices,
                                  const uint8_t *init_states, size_t n_states,
                                  uint8_t *out, const uint64_t *out_off,
                                  uint32_t *out_len, int32_t *status, uint8_t *final_states);

/* K2. */
int avr_range_encode_tiles_device(int device, void *stream,
                                  const void *tiles, const uint64_t *tile_off,
                                  const uint32_t *n_bins, const uint32_t *order, size_t n_slices,
                                  uint8_t *out, const uint64_t *out_off,
                                  uint32_t *out_len, int32_t *status);

/* K1, intra-slice parallel form ("K1p", avrecode-ms_amd/csrc/avr_k1p.h): the same bytes as
 * avr_cabac_encode_tiles_device, produced by many lanes per slice -- for batches of few, long
 * slices (a 1-slice-per-frame clip), where one lane per slice leaves the chip idle.  Input is the
 * slice-major layout (padding records must be no-ops).  Everything runs per chunk of AVR_CHUNK_BINS bins:
 * context states are resolved by a counting sort of each chunk's bins by context (one lane per chunk) and one
 * state chain per (slice, context) over the sorted chunks; the arithmetic coding per chunk follows.  The call
 * renumbers the batch onto the contexts it uses by itself (a census pass over blocks of AVR_SORT_BLOCK_BINS
 * bins, which also validates every record).  The caller supplies the plan (device arrays, n = n_slices):
 *   res_off[i]     byte offset of slice i in the per-bin work arrays, multiple of 16,
 *                  res_off[i+1] - res_off[i] >= roundup16(n_bins[i]) + 16;      res_total = res_off[n]
 *   chunk_base[i]  first global chunk of slice i; it has max(1, ceil(n_bins[i]/AVR_CHUNK_BINS))
 *                  chunks; total_chunks = chunk_base[n]; chunk_slice[c] = slice of global chunk c
 *   blk_base[i], blk_slice[b], total_blocks: the same for blocks of AVR_SORT_BLOCK_BINS bins (the census grid)
 *   dig_off[i]     first 32-bit digit sum of slice i, dig_off[i+1] - dig_off[i] >= n_bins[i]/2 + 8;
 *                  dig_total = dig_off[n]
 * workspace: avr_cabac_chunked_workspace_bytes(...) bytes of device memory, 256-byte aligned.
 * status is in/out as for the tile kernels; this path validates every record itself.  A slice the
 * scheme declines (no coded LPS for 16 consecutive chunks) is coded by the serial kernel in the
 * same call. */
#define AVR_CHUNK_BINS       1024
#define AVR_SORT_BLOCK_BINS  4096
typedef struct {
    const uint64_t *res_off;
    const uint32_t *chunk_base;
    const uint32_t *chunk_slice;
    const uint32_t *blk_base;
    const uint32_t *blk_slice;
    const uint64_t *dig_off;
    uint64_t res_total;
    uint64_t dig_total;
    uint32_t total_chunks;
    uint32_t total_blocks;
} avr_chunk_plan;

size_t avr_cabac_chunked_workspace_bytes(size_t n_slices, size_t n_states, const avr_chunk_plan *plan);
int avr_cabac_encode_chunked_device(int device, void *stream,
                                    const uint16_t *recs, const uint64_t *rec_off,
                                    const uint32_t *n_bins, size_t n_slices,
                                    const uint8_t *init_states, size_t n_states,
                                    const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes,
                                    uint8_t *out, const uint64_t *out_off,
                                    uint32_t *out_len, int32_t *status, uint8_t *final_states);

/* The K1 device calls SIZED BY THE CALLER'S GUESS of how many contexts the batch uses (r4) -- the count an earlier call reported,
 * which is how avr_batch runs from its second batch on -- so that the call enqueues everything and returns: the calls above read
 * that count back from the device before they can size their launches (avr_cabac_encode_chunked_device: two host round trips a
 * call, 70 us of config 2's 1.55 ms; the tiles call: one).
 *   rows_hint  0 = no guess: ask the device and wait, as the calls above do (the counts are reported all the same).
 *   counts     two words of PINNED host memory, valid once the caller has synchronised `stream`: counts[0] = context rows the
 *              batch needs, counts[1] = slices left for a second pass (chunked call only: slices with a bin in a context the
 *              sampled census missed).
 * What the caller does once it has synchronised.  Tiles call: nothing, it is exact whatever the guess (a slice the guess did not
 * fit was coded by the call's second launch).  Chunked call: counts[0] > rows_hint -- the outputs are NOT valid: restore `status`
 * to what it held before the call and call again with rows_hint = 0; else counts[1] > 0 --
 * avr_cabac_encode_chunked_second_pass_device with the same arguments codes the slices that were left. */
int avr_cabac_encode_tiles_device_hinted(int device, void *stream,
                                         const void *tiles, const uint64_t *tile_off,
                                         const uint32_t *n_bins, const uint32_t *order, size_t n_slices,
                                         const uint8_t *init_states, size_t n_states,
                                         uint8_t *out, const uint64_t *out_off,
                                         uint32_t *out_len, int32_t *status, uint8_t *final_states,
                                         uint32_t rows_hint, uint32_t *counts);
int avr_cabac_encode_chunked_device_hinted(int device, void *stream,
                                           const uint16_t *recs, const uint64_t *rec_off,
                                           const uint32_t *n_bins, size_t n_slices,
                                           const uint8_t *init_states, size_t n_states,
                                           const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes,
                                           uint8_t *out, const uint64_t *out_off,
                                           uint32_t *out_len, int32_t *status, uint8_t *final_states,
                                           uint32_t rows_hint, uint32_t *counts);
int avr_cabac_encode_chunked_second_pass_device(int device, void *stream,
                                                const uint16_t *recs, const uint64_t *rec_off,
                                                const uint32_t *n_bins, size_t n_slices,
                                                const uint8_t *init_states, size_t n_states,
                                                const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes,
                                                uint8_t *out, const uint64_t *out_off,
                                                uint32_t *out_len, int32_t *status, uint8_t *final_states);

/* A batch as several parts at once (r4).  The kernels of the intra-slice parallel path are launched over a part's chunks in rounds of
 * workgroups, and the last round of each is part empty -- for a batch the size of config 2 (4 736 waves against the 2 048 and 3 072 its
 * two longest kernels hold at a time) a third of one kernel and half of another.  Slices are independent, so a batch cut into parts of
 * consecutive slices, each part with a plan and a workspace of its own and run on a stream of its own, fills those rounds with the
 * other parts' kernels: config 2 in three parts 1.41 -> 1.32 ms.  This call is that: part i is avr_cabac_encode_chunked_device_hinted on
 * its own arguments (pointers of the whole batch's arrays moved to the part's first slice; rec_off / out_off values stay offsets into
 * the shared recs / out), on a stream the library keeps for (device, stream); the parts start when `stream` has reached the call and
 * `stream` continues when all are done.  What the caller does about each part's counts: as for the hinted call. */
typedef struct {
    const uint64_t *rec_off;  const uint32_t *n_bins;  size_t n_slices;
    const uint8_t *init_states;
    const avr_chunk_plan *plan;  void *workspace;  size_t workspace_bytes;
    const uint64_t *out_off;  uint32_t *out_len;  int32_t *status;  uint8_t *final_states;
    uint32_t rows_hint;  uint32_t *counts;
} avr_chunked_part;
#define AVR_MAX_PARTS 8
int avr_cabac_encode_chunked_device_parts(int device, void *stream, const uint16_t *recs, size_t n_states, uint8_t *out,
                                          const avr_chunked_part *parts, size_t n_parts);

/* K2, intra-slice parallel form ("K2p", avrecode-ms_amd/csrc/avr_k2p.h): the same bytes as avr_range_encode_slices_device for
 * batches of few, long slices.  The range recurrence of arithmetic_code<uint64_t, uint8_t> (recode.cpp:322-323, 823-827) is
 * walked by one lane per slice -- it is exact 63-bit arithmetic on its own previous value and does not decompose --
 * while low, the output bytes, the carries and finish() (arithmetic_code.h:128-144) are done per chunk of AVR_CHUNK_BINS
 * bins and per slice.  Of the plan only chunk_base, chunk_slice and total_chunks are used; out_total = out_off[n_slices];
 * out_off as for the other entry points (capacity n_bins + 16 per slice at least).  Input is the slice-major layout. */
size_t avr_range_chunked_workspace_bytes(size_t n_slices, const avr_chunk_plan *plan, uint64_t out_total);
int avr_range_encode_chunked_device(int device, void *stream,
                                    const uint16_t *recs, const uint64_t *rec_off,
                                    const uint32_t *n_bins, size_t n_slices,
                                    const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes,
                                    uint8_t *out, const uint64_t *out_off, uint64_t out_total,
                                    uint32_t *out_len, int32_t *status);

/* The two stages of K1p on their own.
 *
 * Stage 1, avr_cabac_resolve_device: context-state resolution (phase A).  Writes one RESOLVED CODE
 * per bin, slice i at codes + res_off[i] (codes must hold res_total + 32 bytes, 256-byte aligned):
 *     (state << 1) | bin   context bin met in state = 2*pStateIdx + valMPS <= 125
 *     252 | bin            bypass bin
 *     255 - bin            put_terminate(bin) (and a context bin at pStateIdx 63 with symbol bin)
 * i.e. exactly the inputs cabac::encoder::put takes -- (symbol, *state) -- per bin (cabac_code.h:33).
 *
 * Stage 2, avr_cabac_encode_resolved_device: the arithmetic coding (phases B-D) from resolved codes.
 * A hook adapter that tracks *state itself -- it has to keep libavcodec's state bytes current anyway,
 * cabac_code.h:43-47 -- can record resolved codes directly with the AVR_CODE_* macros and skip stage 1.
 * The plan is the one of avr_cabac_encode_chunked_device (blk_* unused by stage 2).  A slice whose digit
 * sums phase D does not resolve in parallel (a carry of two or more into a 33-digit segment of the form
 * ffff ... fffe) is coded by the serial kernel below in the same call: every slice comes back coded.
 *
 * avr_cabac_encode_codes_device: the serial form, one lane per slice straight from the codes (no plan, no
 * workspace): for batches of many short slices.  order may be NULL (else: slices taken longest first). */
#define AVR_CODE_CONTEXT(state, bin)  ((state) >= 126 ? 255 - (((bin) ^ (state)) & 1) : (((state) << 1) | (bin)))
#define AVR_CODE_BYPASS(bin)          (252 | (bin))
#define AVR_CODE_TERMINATE(bin)       (255 - (bin))
size_t avr_cabac_resolve_workspace_bytes(size_t n_slices, size_t n_states, const avr_chunk_plan *plan);
int avr_cabac_resolve_device(int device, void *stream,
                             const uint16_t *recs, const uint64_t *rec_off,
                             const uint32_t *n_bins, size_t n_slices,
                             const uint8_t *init_states, size_t n_states,
                             const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes,
                             uint8_t *codes, int32_t *status, uint8_t *final_states);
size_t avr_cabac_resolved_workspace_bytes(size_t n_slices, const avr_chunk_plan *plan);
int avr_cabac_encode_resolved_device(int device, void *stream,
                                     const uint8_t *codes, const uint32_t *n_bins, size_t n_slices,
                                     const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes,
                                     uint8_t *out, const uint64_t *out_off,
                                     uint32_t *out_len, int32_t *status);

int avr_cabac_encode_codes_device(int device, void *stream,
                                  const uint8_t *codes, const uint64_t *res_off, const uint32_t *n_bins,
                                  const uint32_t *order, size_t n_slices,
                                  uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status);

/* Variants that read the slice-major layout directly (one 16-byte load per lane per 8 bins,
 * uncoalesced across lanes); kept for the layout comparison in DESIGN.md and for tests.  The
 * padding records up to each slice's next multiple of 8 must be no-op records. */
int avr_cabac_encode_slices_device(int device, void *stream,
                                   const uint16_t *recs, const uint64_t *rec_off,
                                   const uint32_t *n_bins, const uint32_t *order, size_t n_slices,
                                   const uint8_t *init_states, size_t n_states,
                                   uint8_t *out, const uint64_t *out_off,
                                   uint32_t *out_len, int32_t *status, uint8_t *final_states);
int avr_range_encode_slices_device(int device, void *stream,
                                   const uint16_t *recs, const uint64_t *rec_off,
                                   const uint32_t *n_bins, const uint32_t *order, size_t n_slices,
                                   uint8_t *out, const uint64_t *out_off,
                                   uint32_t *out_len, int32_t *status);

/* ------------------------------------------------------------------ dense context ids
 * A context is identified by the offset of its state byte in libavcodec's cabac_state[1024]
 * (recode.cpp:325 keys the model on the address), but a stream uses far fewer of them, and K1
 * keeps 64 lanes x n_states state bytes in LDS per wave: renumbering a batch onto the contexts it
 * uses raises occupancy.  EVERY K1 ENTRY POINT ABOVE DOES THIS BY ITSELF, inside the call (census kernel,
 * look-up on load): records and states go in under the caller's numbering and nothing has to be prepared.
 * The three calls below are the same steps as separate passes, for a caller that wants a renumbered copy of
 * its records for its own purposes; no path of this library needs them.  recs is any flat array of CABAC
 * records on the device (a tile buffer or a slice-major buffer), n_records a multiple of 8.
 *   avr_context_census_device   bitmap[32] |= one bit per selector < 1024 that occurs (zero it first)
 *   avr_context_remap_device    selector s < 1024 -> table[s] in place (table: 1024 x uint16 on the device)
 *   avr_states_permute_device   gather (scatter = 0): dst[slice][j] = src[slice][index[j]], j < n_index,
 *                               or scatter back (1): dst[slice][index[j]] = src[slice][j]
 * The caller builds table / index from the bitmap (128 bytes, host side). */
int avr_context_census_device(int device, void *stream, const uint16_t *recs, uint64_t n_records, uint32_t *bitmap);
int avr_context_remap_device(int device, void *stream, uint16_t *recs, uint64_t n_records, const uint16_t *table);
int avr_states_permute_device(int device, void *stream, const uint8_t *src, size_t n_src, uint8_t *dst, size_t n_dst,
                              const uint16_t *index, size_t n_index, size_t n_slices, int scatter);

/* ------------------------------------------------------------------ synthetic bin streams
 * Seeded generators for the BASELINE.json configurations; the same code
 * runs on the host (avr_synth_*_host) and on the device so that CPU checks and GPU runs see
 * identical records.  `workload`: 2 = 1080p30 slices, 3 = ragged "directory of files",
 * 4 = 4K60 8 slices/frame, 5 = residual-only roofline stress.  `scale_permille` scales the
 * per-slice length (1000 = the configuration's own size) so tests can run small cases.
 * Slice i of a run is generated from (seed, first_slice + i) alone, which is what lets ranks
 * shard a workload without exchanging anything. */
typedef struct {
    int      workload;
    uint32_t scale_permille;
    uint64_t seed;
    uint64_t first_slice;
    uint32_t n_states;       /* out: state bytes per slice this workload declares */
} avr_synth_config;

int avr_synth_config_init(avr_synth_config *cfg, int workload, uint32_t scale_permille,
                          uint64_t first_slice);
/* number of records slice i will have (host; cheap closed loop over the generator) */
int avr_synth_count_host(const avr_synth_config *cfg, int kind, size_t n_slices, uint32_t *n_bins);
int avr_synth_generate_host(const avr_synth_config *cfg, int kind, size_t n_slices,
                            const uint64_t *rec_off, uint16_t *recs, uint8_t *init_states);
int avr_synth_count_device(int device, void *stream, const avr_synth_config *cfg, int kind,
                           size_t n_slices, uint32_t *n_bins);
/* slice-major: slice i at recs + rec_off[i], rec_off multiples of 8, chunk padding = no-op records */
int avr_synth_generate_slices_device(int device, void *stream, const avr_synth_config *cfg, int kind,
                                     size_t n_slices, const uint64_t *rec_off, uint16_t *recs,
                                     uint8_t *init_states);
/* writes straight into the tile layout */
int avr_synth_generate_tiles_device(int device, void *stream, const avr_synth_config *cfg, int kind,
                                    size_t n_slices, const uint32_t *order,
                                    const uint64_t *tile_off, void *tiles, uint8_t *init_states);

/* ------------------------------------------------------------------ host epilogue helpers
 * decompressor::cabac_decoder::finish (recode.cpp:1508-1512): length after dropping a
 * trailing 0x80; decompressor::run tail patch (recode.cpp:1354-1360): buf must have room for
 * len+1 bytes; length_parity -1 = "no patch recorded". */
size_t avr_drop_stop_byte(const uint8_t *buf, size_t len);
size_t avr_tail_patch(uint8_t *buf, size_t len, int length_parity, uint8_t last_byte);

#ifdef __cplusplus
}
#endif
#endif /* AVRECODE_MS_AMD_H */
