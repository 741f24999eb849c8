// H.264 CABAC probability tables (ITU-T H.264 Table 9-44 rangeTabLPS and Table 9-45
// transIdxLPS), build-owned data for the CABAC layer of the reference
// (/root/reference/cabac_code.h:11-12 reads them from libavcodec's ff_h264_cabac_tables,
// which is not part of the reference snapshot).
//
// Two views are generated from the normative tables at compile time:
//   * the libavcodec layout the reference indexes (cabac_code.h:40,43-47):
//       lps_range[(q << 7) + s], mlps_state[128 + s] (MPS), mlps_state[127 - s] (LPS),
//       s = 2*pStateIdx + valMPS
//   * the kernel's packed view: one 8-byte entry per state s,
//       .x = rangeTabLPS[p][0..3] packed little-endian (q selects the byte)
//       .y = next state after MPS | next state after LPS << 8
//     so one ds_read_b64 serves the LPS range and both successors of a bin.
#pragma once
#include <stdint.h>

namespace avr {

struct CabacTables {
    uint8_t range_lps[64][4];
    uint8_t trans_lps[64];
    uint8_t lps_range[512];     // libavcodec layout
    uint8_t mlps_state[256];    // libavcodec layout
    uint32_t packed[128][2];    // kernel layout
};

constexpr CabacTables make_cabac_tables() {
    CabacTables t{};
    constexpr uint8_t r[64][4] = {
        {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205},
        {116, 142, 169, 195}, {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166},
        {95, 116, 137, 158},  {90, 110, 130, 150},  {85, 104, 123, 142},  {81, 99, 117, 135},
        {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},   {66, 80, 95, 110},
        {62, 76, 90, 104},    {59, 72, 86, 99},     {56, 69, 81, 94},     {53, 65, 77, 89},
        {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},
        {41, 50, 59, 69},     {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},
        {33, 41, 48, 56},     {32, 39, 46, 53},     {30, 37, 43, 50},     {29, 35, 41, 48},
        {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},     {23, 28, 33, 39},
        {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},
        {18, 22, 26, 30},     {17, 21, 25, 28},     {16, 20, 23, 27},     {15, 19, 22, 25},
        {14, 18, 21, 24},     {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},
        {12, 14, 17, 20},     {11, 14, 16, 19},     {11, 13, 15, 18},     {10, 12, 15, 17},
        {10, 12, 14, 16},     {9, 11, 13, 15},      {9, 11, 12, 14},      {8, 10, 12, 14},
        {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},
        {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2},
    };
    constexpr uint8_t tl[64] = {
        0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
        18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
        31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63,
    };
    for (int p = 0; p < 64; p++) {
        for (int q = 0; q < 4; q++) t.range_lps[p][q] = r[p][q];
        t.trans_lps[p] = tl[p];
    }
    for (int s = 0; s < 128; s++) {
        const int p = s >> 1, m = s & 1;
        const int p_mps = p < 62 ? p + 1 : p;                       // transIdxMPS
        const uint8_t next_mps = uint8_t(2 * p_mps + m);
        const uint8_t next_lps = uint8_t(p == 0 ? (1 - m) : 2 * tl[p] + m);   // valMPS flips at pStateIdx 0
        for (int q = 0; q < 4; q++) t.lps_range[(q << 7) + s] = r[p][q];
        t.mlps_state[128 + s] = next_mps;
        t.mlps_state[127 - s] = next_lps;
        t.packed[s][0] = uint32_t(r[p][0]) | uint32_t(r[p][1]) << 8 | uint32_t(r[p][2]) << 16 |
                         uint32_t(r[p][3]) << 24;
        t.packed[s][1] = uint32_t(next_mps) | uint32_t(next_lps) << 8;
    }
    return t;
}

}  // namespace avr
