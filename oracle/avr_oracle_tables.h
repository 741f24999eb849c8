/*
 * TEST INFRASTRUCTURE ONLY -- part of oracle/, the CPU checker.  Nothing under
 * avrecode-ms_amd/ (the product) may include, link or call this.
 *
 * H.264 CABAC tables, typed from ITU-T H.264 (Table 9-44 rangeTabLPS,
 * Table 9-45 transIdxLPS / transIdxMPS) and re-indexed the way the reference
 * reads them through libavcodec's ff_h264_cabac_tables
 * (/root/reference/cabac_code.h:11-12,40,43-47):
 *
 *   lps_range[(q << 7) + s]   s = 2*pStateIdx + valMPS, q = 2 bits of range
 *   mlps_state[128 + s]       successor of s after an MPS
 *   mlps_state[127 - s]       successor of s after an LPS (valMPS flips at pStateIdx 0)
 *
 * The libavcodec-hooks fork (.gitmodules:1-4, pinned commit unknown) that holds
 * the reference's own copy is absent from /root/reference, so these values are
 * "parity unpinned" by the reference's runnable tests; they are pinned by the
 * standard and by oracle/spec_cabac.c, an independent bit-serial encoder
 * written from H.264 9.3.4.2 that shares only the two normative tables.
 */
#ifndef AVR_ORACLE_TABLES_H
#define AVR_ORACLE_TABLES_H

#include <stdint.h>

static const uint8_t avr_oracle_rangeTabLPS[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205},
    {116, 142, 169, 195}, {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166},
    { 95, 116, 137, 158}, { 90, 110, 130, 150}, { 85, 104, 123, 142}, { 81,  99, 117, 135},
    { 77,  94, 111, 128}, { 73,  89, 105, 122}, { 69,  85, 100, 116}, { 66,  80,  95, 110},
    { 62,  76,  90, 104}, { 59,  72,  86,  99}, { 56,  69,  81,  94}, { 53,  65,  77,  89},
    { 51,  62,  73,  85}, { 48,  59,  69,  80}, { 46,  56,  66,  76}, { 43,  53,  63,  72},
    { 41,  50,  59,  69}, { 39,  48,  56,  65}, { 37,  45,  54,  62}, { 35,  43,  51,  59},
    { 33,  41,  48,  56}, { 32,  39,  46,  53}, { 30,  37,  43,  50}, { 29,  35,  41,  48},
    { 27,  33,  39,  45}, { 26,  31,  37,  43}, { 24,  30,  35,  41}, { 23,  28,  33,  39},
    { 22,  27,  32,  37}, { 21,  26,  30,  35}, { 20,  24,  29,  33}, { 19,  23,  27,  31},
    { 18,  22,  26,  30}, { 17,  21,  25,  28}, { 16,  20,  23,  27}, { 15,  19,  22,  25},
    { 14,  18,  21,  24}, { 14,  17,  20,  23}, { 13,  16,  19,  22}, { 12,  15,  18,  21},
    { 12,  14,  17,  20}, { 11,  14,  16,  19}, { 11,  13,  15,  18}, { 10,  12,  15,  17},
    { 10,  12,  14,  16}, {  9,  11,  13,  15}, {  9,  11,  12,  14}, {  8,  10,  12,  14},
    {  8,   9,  11,  13}, {  7,   9,  11,  12}, {  7,   9,  10,  12}, {  7,   8,  10,  11},
    {  6,   8,   9,  11}, {  6,   7,   9,  10}, {  6,   7,   8,   9}, {  2,   2,   2,   2},
};

static const uint8_t avr_oracle_transIdxLPS[64] = {
     0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9, 11, 11, 12,
    13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
    24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33,
    33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63,
};

/* transIdxMPS(p) = p + 1 for p < 62, 62 for p == 62, 63 for p == 63. */
static inline int avr_oracle_transIdxMPS(int p) { return p < 62 ? p + 1 : p; }

/* Fill the two re-indexed tables (512 + 256 bytes). */
static inline void avr_oracle_build_tables(uint8_t lps_range[512], uint8_t mlps_state[256]) {
    for (int q = 0; q < 4; q++)
        for (int s = 0; s < 128; s++)
            lps_range[(q << 7) + s] = avr_oracle_rangeTabLPS[s >> 1][q];
    for (int s = 0; s < 128; s++) {
        int p = s >> 1, m = s & 1;
        mlps_state[128 + s] = (uint8_t)(2 * avr_oracle_transIdxMPS(p) + m);
        mlps_state[127 - s] = (uint8_t)(p == 0 ? (1 - m) : 2 * avr_oracle_transIdxLPS[p] + m);
    }
}

#endif
