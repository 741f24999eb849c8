/*
 * oracle/ -- CPU restatement of the reference's arithmetic re-encode path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's `cpu_baseline` leg may load it,
 * and there only as the checker.  The product (avrecode-ms_amd/, include/)
 * never includes, links or calls anything in this directory.
 *
 * Parity status
 *   - a1..a5 (generic range coder, arithmetic_code.h): PINNED -- checked
 *     byte-for-byte against the reference header itself, compiled from
 *     /root/reference/arithmetic_code.h into oracle/_ref (see oracle/Makefile,
 *     tests/test_oracle_vs_ref.py) and against the vectors it produced
 *     (tests/golden/, generator tests/golden/make_golden.py).
 *   - a6..a9 (CABAC layer, cabac_code.h): the layer is restated from the text
 *     of cabac_code.h:26-82; cabac_code.h itself cannot be compiled here
 *     (it includes libavcodec/cabac.h from the absent libavcodec-hooks fork),
 *     so oracle/_ref drives the REAL arithmetic_code<uint32_t,uint16_t,0x200>
 *     with the restated layer, and oracle/spec_cabac.c (bit-serial encoder
 *     written from H.264 9.3.4.2) is the independent second check.
 *   - a10 (table values): parity unpinned by the reference's own runnable
 *     tests (the fork that holds them is absent); values are the normative
 *     H.264 tables -- see avr_oracle_tables.h.
 *   - a11/a12 (adaptive estimator arithmetic, recode.cpp:823-827,1037-1052):
 *     restated from the text; recode.cpp cannot be compiled here (FFmpeg fork,
 *     protoc and libprotobuf absent).
 *
 * Record formats (shared with the C-ABI in include/avrecode_ms_amd.h)
 *   CABAC record  (u16): bit 0 = bin, bits 1..11 = selector
 *                        0..1023 context index, 1024 bypass, 1025 terminate
 *   range record  (u16): bit 0 = bin, bits 1..7 = pos, bits 8..14 = neg
 *                        p(1) = (range / (pos+neg)) * pos   (recode.cpp:823-827)
 */
#ifndef AVR_ORACLE_H
#define AVR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVR_ORACLE_OK             0
#define AVR_ORACLE_ERR_ZERO_PROB  1   /* arithmetic_code.h:116-118 would throw */
#define AVR_ORACLE_ERR_OVERFLOW   2   /* caller's output buffer too small */
#define AVR_ORACLE_ERR_BAD_RECORD 3   /* selector out of range / bin after finish */

#define AVR_SEL_BYPASS    1024
#define AVR_SEL_TERMINATE 1025

/* ---- a1-a4: arithmetic_code<uint64_t,uint16_t>, p(1)=range/2 (test/arithmetic_code.cpp:93-99) */
size_t avr_oracle_half_encode(const uint8_t *bins, size_t n, uint8_t *out, size_t cap, int *status);
/* ---- a5 on the same code (test/arithmetic_code.cpp:103-110) */
void   avr_oracle_half_decode(const uint8_t *bytes, size_t len, size_t n, uint8_t *bins_out);

/* ---- a1-a4 + a11: recoded_code = arithmetic_code<uint64_t,uint8_t> (recode.cpp:322-323,1270)
 * with p(1) = (range/total)*pos taken from each record.  finish() at the end
 * (recode.cpp:1099-1102). */
size_t avr_oracle_range_encode(const uint16_t *recs, size_t n, uint8_t *out, size_t cap, int *status);

/* ---- a5 + a11: incremental decoder on recoded_code (recode.cpp:1429-1430,1447-1448). */
typedef struct avr_oracle_range_decoder avr_oracle_range_decoder;
avr_oracle_range_decoder *avr_oracle_range_decoder_new(const uint8_t *bytes, size_t len);
int  avr_oracle_range_decoder_get(avr_oracle_range_decoder *d, int pos, int neg);
void avr_oracle_range_decoder_free(avr_oracle_range_decoder *d);
/* convenience: decode n bins with the per-bin (pos,neg) of `recs` (bin bit ignored). */
void avr_oracle_range_decode(const uint8_t *bytes, size_t len, const uint16_t *recs, size_t n,
                             uint8_t *bins_out);

/* ---- a6-a10: cabac::encoder (cabac_code.h:26-82).  `states` (n_states bytes,
 * 2*pStateIdx+valMPS each) is updated in place as cabac_code.h:43-47 does.
 * The encoder is finished at the end of the records as ~encoder() would
 * (arithmetic_code.h:100); after put_terminate(1) that is a no-op.
 * Returns the raw byte count (before the decompressor's "drop trailing 0x80",
 * recode.cpp:1508-1512). */
size_t avr_oracle_cabac_encode(const uint16_t *recs, size_t n, uint8_t *states, size_t n_states,
                               uint8_t *out, size_t cap, int *status);

/* The two re-indexed tables, for tests that compare them with the product's. */
void avr_oracle_cabac_tables(uint8_t lps_range[512], uint8_t mlps_state[256]);

/* ---- a11/a12: estimator arithmetic (recode.cpp:1064, 823-827, 1037-1052). */
typedef struct { int pos, neg; } avr_oracle_estimator;          /* starts {1,1} */
uint64_t avr_oracle_probability(uint64_t range, const avr_oracle_estimator *e);
void     avr_oracle_update(avr_oracle_estimator *e, int symbol, int significance_map);

/* ---- a16 tail + a17: decompressor::cabac_decoder::finish (recode.cpp:1508-1512) and the
 * per-block patch of decompressor::run (recode.cpp:1354-1360).  Works in place on
 * buf[0..len); returns the new length (buf needs room for len+1). */
size_t avr_oracle_drop_stop_byte(const uint8_t *buf, size_t len);
size_t avr_oracle_tail_patch(uint8_t *buf, size_t len, int length_parity, uint8_t last_byte);

/* ---- independent second oracle for a6-a9: bit-serial encoder from H.264 9.3.4.2
 * (oracle/spec_cabac.c).  Same record format / state bytes / return value. */
size_t avr_spec_cabac_encode(const uint16_t *recs, size_t n, uint8_t *states, size_t n_states,
                             uint8_t *out, size_t cap, int *status);

/* Decoder of the same standard (9.3.3.2), the stand-in for ff_get_cabac* (recode.cpp:1183,1189,
 * 1195): recovers bins_out[i] for the selector of recs[i] (the bin bit of recs is ignored). */
int avr_spec_cabac_decode(const uint8_t *bytes, size_t len, const uint16_t *recs, size_t n,
                          uint8_t *states, size_t n_states, uint8_t *bins_out);

/* ---- batch helper for bench.py's cpu_baseline leg and large parity checks:
 * encode `n_slices` slices whose records are rec[off[i] .. off[i+1]) with
 * `threads` worker threads (one slice per task).  init_states is
 * n_slices*n_states bytes (copied, not modified).  out_off[i] is where slice
 * i's bytes go in `out` (capacity out_off[i+1]-out_off[i]); out_len[i] gets
 * the produced length.  kind: 0 = CABAC (K1), 1 = range (K2). */
int avr_oracle_encode_batch(int kind, const uint16_t *recs, const uint64_t *off, size_t n_slices,
                            const uint8_t *init_states, size_t n_states,
                            uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                            int32_t *status, int threads);

#ifdef __cplusplus
}
#endif
#endif
