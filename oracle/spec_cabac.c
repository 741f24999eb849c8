/*
 * TEST INFRASTRUCTURE ONLY (oracle/) -- independent second oracle for the
 * CABAC layer ("K1 has two oracles").
 *
 * A bit-serial CABAC encoder written directly from ITU-T H.264 clause
 * 9.3.4.2 (Figures 9-7 .. 9-12: EncodeDecision, RenormE, PutBit, EncodeBypass,
 * EncodeTerminate, EncodeFlush) with 9-bit codIRange, 10-bit codILow,
 * bitsOutstanding and firstBitFlag.  It shares nothing with avr_oracle.c
 * except the two normative tables, and nothing with the reference except the
 * record format: its job is to show that the restated cabac::encoder
 * (/root/reference/cabac_code.h:26-82 on arithmetic_code.h) emits the byte
 * string the standard defines.  The stop bit and zero alignment are written
 * as 7.3.2.11 rbsp_trailing_bits does after end_of_slice_segment.
 */
#include "avr_oracle.h"
#include "avr_oracle_tables.h"

typedef struct {
    unsigned low, range;
    unsigned outstanding;
    int first;
    uint8_t *out; size_t cap;
    size_t nbits;
    int error;
} spec_enc;

static void write_bit(spec_enc *e, int b) {
    size_t byte = e->nbits >> 3;
    if (byte < e->cap) {
        if ((e->nbits & 7) == 0) e->out[byte] = 0;
        e->out[byte] |= (uint8_t)(b << (7 - (e->nbits & 7)));
    } else {
        e->error = AVR_ORACLE_ERR_OVERFLOW;
    }
    e->nbits++;
}

static void put_bit(spec_enc *e, int b) {               /* Figure 9-9 */
    if (e->first) e->first = 0; else write_bit(e, b);
    while (e->outstanding > 0) { write_bit(e, 1 - b); e->outstanding--; }
}

static void renorm(spec_enc *e) {                        /* Figure 9-8 */
    while (e->range < 256) {
        if (e->low < 256) put_bit(e, 0);
        else if (e->low >= 512) { e->low -= 512; put_bit(e, 1); }
        else { e->low -= 256; e->outstanding++; }
        e->range <<= 1; e->low <<= 1;
    }
}

static void encode_decision(spec_enc *e, uint8_t *state, int bin) {    /* Figure 9-7 */
    int p = *state >> 1, mps = *state & 1;
    unsigned q = (e->range >> 6) & 3;
    unsigned rlps = avr_oracle_rangeTabLPS[p][q];
    e->range -= rlps;
    if (bin != mps) {
        e->low += e->range; e->range = rlps;
        if (p == 0) mps = 1 - mps;
        p = avr_oracle_transIdxLPS[p];
    } else {
        p = avr_oracle_transIdxMPS(p);
    }
    *state = (uint8_t)(2 * p + mps);
    renorm(e);
}

static void encode_bypass(spec_enc *e, int bin) {        /* Figure 9-10 */
    e->low <<= 1;
    if (bin) e->low += e->range;
    if (e->low >= 1024) { put_bit(e, 1); e->low -= 1024; }
    else if (e->low < 512) put_bit(e, 0);
    else { e->low -= 512; e->outstanding++; }
}

static void encode_flush(spec_enc *e) {                  /* Figure 9-12 */
    e->range = 2;
    renorm(e);
    put_bit(e, (e->low >> 9) & 1);
    write_bit(e, (e->low >> 8) & 1);
    write_bit(e, 1);        /* ((codILow >> 7) & 3) | 1: the low bit is rbsp_stop_one_bit */
}

size_t avr_spec_cabac_encode(const uint16_t *recs, size_t n, uint8_t *states, size_t n_states,
                             uint8_t *out, size_t cap, int *status) {
    spec_enc e = { 0, 510, 0, 1, out, cap, 0, 0 };       /* 9.3.4.1 initialisation */
    int finished = 0;
    for (size_t i = 0; i < n && !e.error; i++) {
        int bin = recs[i] & 1;
        unsigned sel = (recs[i] >> 1) & 0x7ff;
        if (finished) { e.error = AVR_ORACLE_ERR_BAD_RECORD; break; }
        if (sel < 1024) {
            if (sel >= n_states) { e.error = AVR_ORACLE_ERR_BAD_RECORD; break; }
            encode_decision(&e, &states[sel], bin);
        } else if (sel == AVR_SEL_BYPASS) {
            encode_bypass(&e, bin);
        } else if (sel == AVR_SEL_TERMINATE) {           /* Figure 9-11 */
            e.range -= 2;
            if (bin) { e.low += e.range; encode_flush(&e); finished = 1; }
            else renorm(&e);
        } else {
            e.error = AVR_ORACLE_ERR_BAD_RECORD;
        }
    }
    if (status) *status = e.error;
    /* rbsp_alignment_zero_bit: the partially filled last byte is already zero padded */
    return (e.nbits + 7) >> 3;
}

/* ------------------------------------------------------------------ decoder (H.264 9.3.3.2)
 * The arithmetic decoding engine of the standard (Figures 9-2, 9-3, 9-5, 9-6): the
 * counterpart of libavcodec's ff_get_cabac / ff_get_cabac_bypass /
 * ff_get_cabac_terminate that the reference calls (recode.cpp:1183,1189,1195) and that
 * its disabled unit test uses to check cabac::encoder (test/arithmetic_code.cpp:37-45,
 * 79-90).  Given the selector of every bin it recovers the bin values from coded bytes,
 * which gives the tests an encode -> decode round trip at any size.  Bits past the end
 * of the buffer read as zero. */
typedef struct { const uint8_t *in; size_t nbits, pos; unsigned range, offset; } spec_dec;

static unsigned read_bit(spec_dec *d) {
    unsigned b = 0;
    if (d->pos < d->nbits) b = (d->in[d->pos >> 3] >> (7 - (d->pos & 7))) & 1;
    d->pos++;
    return b;
}

int avr_spec_cabac_decode(const uint8_t *bytes, size_t len, const uint16_t *recs, size_t n,
                          uint8_t *states, size_t n_states, uint8_t *bins_out) {
    spec_dec d = { bytes, len * 8, 0, 510, 0 };
    for (int i = 0; i < 9; i++) d.offset = (d.offset << 1) | read_bit(&d);      /* 9.3.1.2 */
    for (size_t i = 0; i < n; i++) {
        unsigned sel = (recs[i] >> 1) & 0x7ff;
        int bin;
        if (sel < 1024) {                                                        /* Figure 9-3 */
            if (sel >= n_states) return AVR_ORACLE_ERR_BAD_RECORD;
            int p = states[sel] >> 1, mps = states[sel] & 1;
            unsigned rlps = avr_oracle_rangeTabLPS[p][(d.range >> 6) & 3];
            d.range -= rlps;
            if (d.offset >= d.range) {
                bin = !mps; d.offset -= d.range; d.range = rlps;
                if (p == 0) mps = 1 - mps;
                p = avr_oracle_transIdxLPS[p];
            } else {
                bin = mps; p = avr_oracle_transIdxMPS(p);
            }
            states[sel] = (uint8_t)(2 * p + mps);
            while (d.range < 256) { d.range <<= 1; d.offset = (d.offset << 1) | read_bit(&d); }
        } else if (sel == AVR_SEL_BYPASS) {                                      /* Figure 9-5 */
            d.offset = (d.offset << 1) | read_bit(&d);
            bin = d.offset >= d.range;
            if (bin) d.offset -= d.range;
        } else if (sel == AVR_SEL_TERMINATE) {                                   /* Figure 9-6 */
            d.range -= 2;
            bin = d.offset >= d.range;
            if (!bin) while (d.range < 256) { d.range <<= 1; d.offset = (d.offset << 1) | read_bit(&d); }
        } else {
            return AVR_ORACLE_ERR_BAD_RECORD;
        }
        bins_out[i] = (uint8_t)bin;
    }
    return AVR_ORACLE_OK;
}
