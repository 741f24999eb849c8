/*
 * TEST INFRASTRUCTURE ONLY -- see avr_oracle.h for who may use this and for
 * the parity status of each part.
 *
 * CPU restatement of the reference's arithmetic re-encode path:
 *   arithmetic_code.h  (generic binary range coder, encoder + decoder)
 *   cabac_code.h       (H.264 CABAC expressed on top of it)
 *   recode.cpp:823-827, 1037-1052, 1064 (adaptive {pos,neg} estimator)
 *   recode.cpp:1508-1512, 1354-1360     (stop-byte drop and tail patch)
 */
#include "avr_oracle.h"
#include "avr_oracle_tables.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- instantiation 1: arithmetic_code<uint64_t,uint16_t> (test/arithmetic_code.cpp:93),
 * min_range = (fixed_one/digit_base)/16 = 2^43 (arithmetic_code.h:61-62). */
#define AVR_F uint64_t
#define AVR_FBITS 64
#define AVR_DBITS 16
#define AVR_MINRANGE ((uint64_t)1 << 43)
#define AVR_N(x) t64_16_##x
#include "avr_oracle_coder.inc"
#undef AVR_F
#undef AVR_FBITS
#undef AVR_DBITS
#undef AVR_MINRANGE
#undef AVR_N

/* ---- instantiation 2: recoded_code = arithmetic_code<uint64_t,uint8_t> (recode.cpp:322-323),
 * min_range = (2^63/2^8)/16 = 2^51. */
#define AVR_F uint64_t
#define AVR_FBITS 64
#define AVR_DBITS 8
#define AVR_MINRANGE ((uint64_t)1 << 51)
#define AVR_N(x) r64_8_##x
#include "avr_oracle_coder.inc"
#undef AVR_F
#undef AVR_FBITS
#undef AVR_DBITS
#undef AVR_MINRANGE
#undef AVR_N

/* ---- instantiation 3: cabac_arithmetic_code = arithmetic_code<uint32_t,uint16_t,0x200>
 * (cabac_code.h:18-24). */
#define AVR_F uint32_t
#define AVR_FBITS 32
#define AVR_DBITS 16
#define AVR_MINRANGE 0x200u
#define AVR_N(x) c32_16_##x
#include "avr_oracle_coder.inc"
#undef AVR_F
#undef AVR_FBITS
#undef AVR_DBITS
#undef AVR_MINRANGE
#undef AVR_N

static void set_status(int *status, int v) { if (status) *status = v; }

/* ------------------------------------------------------------------ a1-a5, p = 1/2 */

size_t avr_oracle_half_encode(const uint8_t *bins, size_t n, uint8_t *out, size_t cap, int *status) {
    t64_16_enc e;
    t64_16_enc_init(&e, (uint64_t)1 << 63, out, cap);           /* arithmetic_code.h:96-97 */
    for (size_t i = 0; i < n && !e.error; i++)
        t64_16_enc_put(&e, bins[i] != 0, e.range / 2);          /* test/arithmetic_code.cpp:97 */
    if (!e.error) t64_16_enc_finish(&e);                        /* :99 */
    set_status(status, e.error);
    return e.n;
}

void avr_oracle_half_decode(const uint8_t *bytes, size_t len, size_t n, uint8_t *bins_out) {
    t64_16_dec d;
    t64_16_dec_init(&d, bytes, len, (uint64_t)1 << 63);         /* arithmetic_code.h:218-219 */
    for (size_t i = 0; i < n; i++)
        bins_out[i] = (uint8_t)t64_16_dec_get(&d, d.range / 2); /* test/arithmetic_code.cpp:105 */
}

/* ------------------------------------------------------------------ a11 / a12 */

uint64_t avr_oracle_probability(uint64_t range, const avr_oracle_estimator *e) {
    int total = e->pos + e->neg;                                /* recode.cpp:825 */
    return (range / (uint64_t)total) * (uint64_t)e->pos;        /* recode.cpp:826 */
}

void avr_oracle_update(avr_oracle_estimator *e, int symbol, int significance_map) {
    if (symbol) e->pos++; else e->neg++;                        /* recode.cpp:1043-1047 */
    int limit = significance_map ? 0x50 : 0x60;                 /* recode.cpp:1048-1049 */
    if (e->pos + e->neg > limit) {
        e->pos = (e->pos + 1) / 2;                              /* recode.cpp:1050-1051 */
        e->neg = (e->neg + 1) / 2;
    }
}

/* ------------------------------------------------------------------ K2: range records */

static uint64_t rec_probability(uint64_t range, uint16_t rec, int *bad) {
    unsigned pos = (rec >> 1) & 0x7f, neg = (rec >> 8) & 0x7f;
    unsigned total = pos + neg;
    if (total == 0) { *bad = 1; return 0; }
    return (range / total) * pos;
}

size_t avr_oracle_range_encode(const uint16_t *recs, size_t n, uint8_t *out, size_t cap, int *status) {
    r64_8_enc e;
    r64_8_enc_init(&e, (uint64_t)1 << 63, out, cap);
    int bad = 0;
    for (size_t i = 0; i < n && !e.error; i++) {
        uint64_t r1 = rec_probability(e.range, recs[i], &bad);
        if (bad) { e.error = AVR_ORACLE_ERR_BAD_RECORD; break; }
        r64_8_enc_put(&e, recs[i] & 1, r1);                     /* recode.cpp:1081-1082 */
    }
    if (!e.error) r64_8_enc_finish(&e);                         /* recode.cpp:1100 */
    set_status(status, e.error);
    return e.n;
}

struct avr_oracle_range_decoder { r64_8_dec d; };

avr_oracle_range_decoder *avr_oracle_range_decoder_new(const uint8_t *bytes, size_t len) {
    avr_oracle_range_decoder *d = (avr_oracle_range_decoder *)malloc(sizeof *d);
    if (d) r64_8_dec_init(&d->d, bytes, len, (uint64_t)1 << 63);   /* recode.cpp:1429-1430 */
    return d;
}

int avr_oracle_range_decoder_get(avr_oracle_range_decoder *d, int pos, int neg) {
    avr_oracle_estimator e = { pos, neg };
    return r64_8_dec_get(&d->d, avr_oracle_probability(d->d.range, &e));   /* recode.cpp:1447-1448 */
}

void avr_oracle_range_decoder_free(avr_oracle_range_decoder *d) { free(d); }

void avr_oracle_range_decode(const uint8_t *bytes, size_t len, const uint16_t *recs, size_t n,
                             uint8_t *bins_out) {
    r64_8_dec d;
    r64_8_dec_init(&d, bytes, len, (uint64_t)1 << 63);
    for (size_t i = 0; i < n; i++) {
        int bad = 0;
        uint64_t r1 = rec_probability(d.range, recs[i], &bad);
        bins_out[i] = (uint8_t)r64_8_dec_get(&d, r1);
    }
}

/* ------------------------------------------------------------------ K1: CABAC */

static uint8_t g_lps_range[512], g_mlps_state[256];
static pthread_once_t g_tables_once = PTHREAD_ONCE_INIT;
static void build_tables_once(void) { avr_oracle_build_tables(g_lps_range, g_mlps_state); }

void avr_oracle_cabac_tables(uint8_t lps_range[512], uint8_t mlps_state[256]) {
    pthread_once(&g_tables_once, build_tables_once);
    memcpy(lps_range, g_lps_range, 512);
    memcpy(mlps_state, g_mlps_state, 256);
}

/* floor(log2(x)), x > 0 -- what cabac_code.h:70-79 computes by halving. */
static int floor_log2_u32(uint32_t x) { int i = 0; while (x >>= 1) i++; return i; }

size_t avr_oracle_cabac_encode(const uint16_t *recs, size_t n, uint8_t *states, size_t n_states,
                               uint8_t *out, size_t cap, int *status) {
    pthread_once(&g_tables_once, build_tables_once);
    c32_16_enc e;
    /* cabac_code.h:30: (fixed_one/0x200)*0x1FE = 0x7F800000 */
    c32_16_enc_init(&e, (uint32_t)((((uint32_t)1 << 31) / 0x200u) * 0x1FEu), out, cap);
    int finished = 0;
    for (size_t i = 0; i < n && !e.error; i++) {
        int bin = recs[i] & 1;
        unsigned sel = (recs[i] >> 1) & 0x7ff;
        if (finished) { e.error = AVR_ORACLE_ERR_BAD_RECORD; break; }   /* range == 0 after finish() */
        int normalize = floor_log2_u32(e.range / 0x100u);               /* cabac_code.h:37,59 */
        if (sel < 1024) {
            if (sel >= n_states) { e.error = AVR_ORACLE_ERR_BAD_RECORD; break; }
            uint8_t *state = &states[sel];
            int is_lps = bin != (*state & 1);                            /* cabac_code.h:34 */
            unsigned approx = e.range >> (normalize - 1);                /* :39 */
            uint32_t rlps = (uint32_t)g_lps_range[(approx & 0x180) + *state] << normalize;   /* :40-41 */
            c32_16_enc_put(&e, is_lps, rlps);                            /* :35 */
            *state = is_lps ? g_mlps_state[127 - *state]                 /* :43-47 */
                            : g_mlps_state[128 + *state];
        } else if (sel == AVR_SEL_BYPASS) {
            c32_16_enc_put(&e, bin, e.range / 2);                        /* :52-54 */
        } else if (sel == AVR_SEL_TERMINATE) {
            c32_16_enc_put(&e, bin, (uint32_t)2 << normalize);           /* :58-61 */
            if (bin) { c32_16_enc_finish(&e); finished = 1; }            /* :63-65 */
        } else {
            e.error = AVR_ORACLE_ERR_BAD_RECORD;
        }
    }
    if (!e.error && !finished) c32_16_enc_finish(&e);                    /* arithmetic_code.h:100 */
    set_status(status, e.error);
    return e.n;
}

/* ------------------------------------------------------------------ a16 tail / a17 */

size_t avr_oracle_drop_stop_byte(const uint8_t *buf, size_t len) {
    /* recode.cpp:1510-1512 (cabac_out.back() on an empty vector is undefined there;
     * an encoder that has been finished never leaves it empty) */
    return (len > 0 && buf[len - 1] == 0x80) ? len - 1 : len;
}

size_t avr_oracle_tail_patch(uint8_t *buf, size_t len, int length_parity, uint8_t last_byte) {
    if (length_parity == -1) return len;                                 /* recode.cpp:1354 */
    if (length_parity != (int)(len & 1)) { buf[len] = last_byte; return len + 1; }   /* :1356-1357 */
    if (len > 0) buf[len - 1] = last_byte;                               /* :1359 */
    return len;
}

/* ------------------------------------------------------------------ threaded batch helper */

typedef struct {
    int kind;
    const uint16_t *recs; const uint64_t *off; size_t n_slices;
    const uint8_t *init_states; size_t n_states;
    uint8_t *out; const uint64_t *out_off; uint32_t *out_len; int32_t *status;
    size_t next; pthread_mutex_t mu;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    uint8_t states[1024];
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t i = j->next, hi = i + 16;
        if (hi > j->n_slices) hi = j->n_slices;
        j->next = hi;
        pthread_mutex_unlock(&j->mu);
        if (i >= hi) break;
        for (; i < hi; i++) {
            const uint16_t *r = j->recs + j->off[i];
            size_t n = (size_t)(j->off[i + 1] - j->off[i]);
            uint8_t *o = j->out + j->out_off[i];
            size_t cap = (size_t)(j->out_off[i + 1] - j->out_off[i]);
            int st = 0; size_t len;
            if (j->kind == 0) {
                memset(states, 0, sizeof states);
                if (j->n_states) memcpy(states, j->init_states + i * j->n_states, j->n_states);
                len = avr_oracle_cabac_encode(r, n, states, j->n_states, o, cap, &st);
            } else {
                len = avr_oracle_range_encode(r, n, o, cap, &st);
            }
            j->out_len[i] = (uint32_t)len;
            if (j->status) j->status[i] = st;
        }
    }
    return NULL;
}

int avr_oracle_encode_batch(int kind, const uint16_t *recs, const uint64_t *off, size_t n_slices,
                            const uint8_t *init_states, size_t n_states,
                            uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                            int32_t *status, int threads) {
    if (n_states > 1024) return -1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    batch_job j = { kind, recs, off, n_slices, init_states, n_states, out, out_off, out_len, status, 0,
                    PTHREAD_MUTEX_INITIALIZER };
    pthread_t th[256];
    int started = 0;
    for (int t = 0; t < threads - 1; t++)
        if (pthread_create(&th[started], NULL, batch_worker, &j) == 0) started++;
    batch_worker(&j);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    return 0;
}
