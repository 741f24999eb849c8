// H.264 CABAC data the slice-data syntax parser (avr_h264.h) needs, from ITU-T H.264:
//   * context variable initialisation (m, n) pairs, Tables 9-12 .. 9-23, for I slices and for P / B slices
//     with cabac_init_idc 0;
//   * ctxIdxInc of significant_coeff_flag / last_significant_coeff_flag for 8x8 blocks, Table 9-43 (frame coded);
//   * ctxIdxOffset / ctxBlockCatOffset of the residual syntax elements, Tables 9-34 and 9-40.
//
// Provenance, stated plainly: libavcodec (where the reference gets per-slice state initialisation from, SURVEY.md
// 8(c) item 3) is not in this container and no file here holds these tables, so the values were typed from the
// standard as remembered and are PINNED ONLY BY DECODING REAL STREAMS: tests/test_h264.py parses every slice of the
// CABAC MP4s found in the image and requires each to end on end_of_slice_flag = 1 at its last macroblock with the
// payload consumed to the byte -- a wrong (m, n) of a context those streams use derails the arithmetic decoder
// within a few bins.  Contexts they do not use stay unverified.  The columns for cabac_init_idc 1 and 2 and the
// field-coded ranges (277..398, 436..459 and their 4:4:4 copies) are NOT reproduced: slices that need them are not
// hooked and stay literal bytes in the container (lossless either way).  The same goes for any slice the parser
// does not get through: compressor asks h264_stream_decoder::payload_decodes() before it commits to a slice.
#pragma once
#include <stdint.h>

namespace avr {
namespace h264 {

struct mn { int8_t m, n; };

// ctxIdx 0..10 (mb_type of SI / I slices), every slice type
constexpr mn kInit0_10[11] = {{20, -15}, {2, 54}, {3, 74}, {20, -15}, {2, 54}, {3, 74}, {-28, 127}, {-23, 104}, {-6, 53}, {-1, 54}, {7, 51}};

// ctxIdx 11..59, P / B slices, cabac_init_idc 0: mb_skip_flag (11-13 P, 24-26 B), mb_type (14-20 P, 27-35 B),
// sub_mb_type (21-23 P, 36-39 B), mvd (40-53), ref_idx (54-59)
constexpr mn kInit11_59_idc0[49] = {
    {23, 33}, {23, 2}, {21, 0}, {1, 9}, {0, 49}, {-37, 118}, {5, 57}, {-13, 78}, {-11, 65}, {1, 62}, {12, 49}, {-4, 73}, {17, 50},
    {18, 64}, {9, 43}, {29, 0}, {26, 67}, {16, 90}, {9, 104}, {-46, 127}, {-20, 104}, {1, 67}, {-13, 78}, {-11, 65}, {1, 62},
    {-6, 86}, {-17, 95}, {-6, 61}, {9, 45},
    {-3, 69}, {-6, 81}, {-11, 96}, {6, 55}, {7, 67}, {-5, 86}, {2, 88}, {0, 58}, {-3, 76}, {-10, 94}, {5, 54}, {4, 69}, {-3, 81}, {0, 88},
    {-7, 67}, {-5, 74}, {-4, 74}, {-5, 80}, {-7, 72}, {1, 58}};

// ctxIdx 60..69 (mb_qp_delta, intra_chroma_pred_mode, prev_intra_pred_mode_flag, rem_intra_pred_mode), every slice type
constexpr mn kInit60_69[10] = {{0, 41}, {0, 63}, {0, 63}, {0, 63}, {-9, 83}, {4, 86}, {0, 97}, {-7, 72}, {13, 41}, {3, 62}};

// ctxIdx 70..104: mb_field_decoding_flag (70-72), coded_block_pattern (73-84), coded_block_flag (85-104)
constexpr mn kInit70_104_I[35] = {
    {0, 11}, {1, 55}, {0, 69}, {-17, 127}, {-13, 102}, {0, 82}, {-7, 74}, {-21, 107}, {-27, 127}, {-31, 127}, {-24, 127}, {-18, 95},
    {-27, 127}, {-21, 114}, {-30, 127}, {-17, 123}, {-12, 115}, {-16, 122}, {-11, 115}, {-12, 63}, {-2, 68}, {-15, 84}, {-13, 104},
    {-3, 70}, {-8, 93}, {-10, 90}, {-30, 127}, {-1, 74}, {-6, 97}, {-7, 91}, {-20, 127}, {-4, 56}, {-5, 82}, {-7, 76}, {-22, 125}};
constexpr mn kInit70_104_idc0[35] = {
    {0, 45}, {-4, 78}, {-3, 96}, {-27, 126}, {-28, 98}, {-25, 101}, {-23, 67}, {-28, 82}, {-20, 94}, {-16, 83}, {-22, 110}, {-21, 91},
    {-18, 102}, {-13, 93}, {-29, 127}, {-7, 92}, {-5, 89}, {-7, 96}, {-13, 108}, {-3, 46}, {-1, 65}, {-1, 57}, {-9, 93}, {-3, 74},
    {-9, 92}, {-8, 87}, {-23, 126}, {5, 54}, {6, 60}, {6, 59}, {6, 69}, {-1, 48}, {0, 68}, {-4, 69}, {-8, 88}};

// ctxIdx 105..165: significant_coeff_flag, frame coded, ctxBlockCat 0..4
constexpr mn kInit105_165_I[61] = {
    {-7, 93}, {-11, 87}, {-3, 77}, {-5, 71}, {-4, 63}, {-4, 68}, {-12, 84}, {-7, 62}, {-7, 65}, {8, 61}, {5, 56}, {-2, 66}, {1, 64},
    {0, 61}, {-2, 78}, {1, 50}, {7, 52}, {10, 35}, {0, 44}, {11, 38}, {1, 45}, {0, 46}, {5, 44}, {31, 17}, {1, 51}, {7, 50}, {28, 19},
    {16, 33}, {14, 62}, {-13, 108}, {-15, 100}, {-13, 101}, {-13, 91}, {-12, 94}, {-10, 88}, {-16, 84}, {-10, 86}, {-7, 83},
    {-13, 87}, {-19, 94}, {1, 70}, {0, 72}, {-5, 74}, {18, 59}, {-8, 102}, {-15, 100}, {0, 95}, {-4, 75}, {2, 72}, {-11, 75},
    {-3, 71}, {15, 46}, {-13, 69}, {0, 62}, {0, 65}, {21, 37}, {-15, 72}, {9, 57}, {16, 54}, {0, 62}, {12, 72}};
constexpr mn kInit105_165_idc0[61] = {
    {-2, 85}, {-6, 78}, {-1, 75}, {-7, 77}, {2, 54}, {5, 50}, {-3, 68}, {1, 50}, {6, 42}, {-4, 81}, {1, 63}, {-4, 70}, {0, 67},
    {2, 57}, {-2, 76}, {11, 35}, {4, 64}, {1, 61}, {11, 35}, {18, 25}, {12, 24}, {13, 29}, {13, 36}, {-10, 93}, {-7, 73}, {-2, 73},
    {13, 46}, {9, 49}, {-7, 100}, {9, 53}, {2, 53}, {5, 53}, {-2, 61}, {0, 56}, {0, 56}, {-13, 63}, {-5, 60}, {-1, 62}, {4, 57},
    {-6, 69}, {4, 57}, {14, 39}, {4, 51}, {13, 68}, {3, 64}, {1, 61}, {9, 63}, {7, 50}, {16, 39}, {5, 44}, {4, 52}, {11, 48},
    {-5, 60}, {-1, 59}, {0, 59}, {22, 33}, {5, 44}, {14, 43}, {-1, 78}, {0, 60}, {9, 69}};

// ctxIdx 166..226: last_significant_coeff_flag, frame coded, ctxBlockCat 0..4
constexpr mn kInit166_226_I[61] = {
    {24, 0}, {15, 9}, {8, 25}, {13, 18}, {15, 9}, {13, 19}, {10, 37}, {12, 18}, {6, 29}, {20, 33}, {15, 30}, {4, 45}, {1, 58}, {0, 62},
    {7, 61}, {12, 38}, {11, 45}, {15, 39}, {11, 42}, {13, 44}, {16, 45}, {12, 41}, {10, 49}, {30, 34}, {18, 42}, {10, 55}, {17, 51},
    {17, 46}, {0, 89}, {26, -19}, {22, -17}, {26, -17}, {30, -25}, {28, -20}, {33, -23}, {37, -27}, {33, -23}, {40, -28}, {38, -17},
    {33, -11}, {40, -15}, {41, -6}, {38, 1}, {41, 17}, {30, -6}, {27, 3}, {26, 22}, {37, -16}, {35, -4}, {38, -8}, {38, -3}, {37, 3},
    {38, 5}, {42, 0}, {35, 16}, {39, 22}, {14, 48}, {27, 37}, {21, 60}, {12, 68}, {2, 97}};
constexpr mn kInit166_226_idc0[61] = {
    {11, 28}, {2, 40}, {3, 44}, {0, 49}, {0, 46}, {2, 44}, {2, 51}, {0, 47}, {4, 39}, {2, 62}, {6, 46}, {0, 54}, {3, 54}, {2, 58},
    {4, 63}, {6, 51}, {6, 57}, {7, 53}, {6, 52}, {6, 55}, {11, 45}, {14, 36}, {8, 53}, {-1, 82}, {7, 55}, {-3, 78}, {15, 46}, {22, 31},
    {-1, 84}, {25, 7}, {30, -7}, {28, 3}, {28, 4}, {32, 0}, {34, -1}, {30, 6}, {30, 6}, {32, 9}, {31, 19}, {26, 27}, {26, 30}, {37, 20},
    {28, 34}, {17, 70}, {1, 67}, {5, 59}, {9, 67}, {16, 30}, {18, 32}, {18, 35}, {22, 29}, {24, 31}, {23, 38}, {18, 43}, {20, 41},
    {11, 63}, {9, 59}, {9, 64}, {-1, 94}, {-2, 89}, {-9, 108}};

// ctxIdx 227..275: coeff_abs_level_minus1, ctxBlockCat 0..4
constexpr mn kInit227_275_I[49] = {
    {-3, 71}, {-6, 42}, {-5, 50}, {-3, 54}, {-2, 62}, {0, 58}, {1, 63}, {-2, 72}, {-1, 74}, {-9, 91}, {-5, 67}, {-5, 27}, {-3, 39},
    {-2, 44}, {0, 46}, {-16, 64}, {-8, 68}, {-10, 78}, {-6, 77}, {-10, 86}, {-12, 92}, {-15, 55}, {-10, 60}, {-6, 62}, {-4, 65},
    {-12, 73}, {-8, 76}, {-7, 80}, {-9, 88}, {-17, 110}, {-11, 97}, {-20, 84}, {-11, 79}, {-6, 73}, {-4, 74}, {-13, 86}, {-13, 96},
    {-11, 97}, {-19, 117}, {-8, 78}, {-5, 33}, {-4, 48}, {-2, 53}, {-3, 62}, {-13, 71}, {-10, 79}, {-12, 86}, {-13, 90}, {-14, 97}};
constexpr mn kInit227_275_idc0[49] = {
    {-6, 76}, {-2, 44}, {0, 45}, {0, 52}, {-3, 64}, {-2, 59}, {-4, 70}, {-4, 75}, {-8, 82}, {-17, 102}, {-9, 77}, {3, 24}, {0, 42},
    {0, 48}, {0, 55}, {-6, 59}, {-7, 71}, {-12, 83}, {-11, 87}, {-30, 119}, {1, 58}, {-3, 29}, {-1, 36}, {1, 38}, {2, 43}, {-6, 55},
    {0, 58}, {0, 64}, {-3, 74}, {-10, 90}, {0, 70}, {-4, 29}, {5, 31}, {7, 42}, {1, 59}, {-2, 58}, {-3, 72}, {-3, 81}, {-11, 97},
    {0, 58}, {8, 5}, {10, 14}, {14, 18}, {13, 27}, {2, 40}, {0, 58}, {-3, 70}, {-6, 79}, {-8, 85}};

// ctxIdx 399..401: transform_size_8x8_flag
constexpr mn kInit399_401_I[3] = {{31, 21}, {31, 31}, {25, 50}};
constexpr mn kInit399_401_idc0[3] = {{12, 40}, {11, 51}, {14, 59}};

// ctxIdx 402..435: 8x8 blocks (ctxBlockCat 5), frame coded: significant (402-416), last (417-425), abs level (426-435)
constexpr mn kInit402_435_I[34] = {
    {-17, 120}, {-20, 112}, {-18, 114}, {-11, 85}, {-15, 92}, {-14, 89}, {-26, 71}, {-15, 81}, {-14, 80}, {0, 68}, {-14, 70},
    {-24, 56}, {-23, 68}, {-24, 50}, {-11, 74}, {23, -13}, {26, -13}, {40, -15}, {49, -14}, {44, 3}, {45, 6}, {44, 34}, {33, 54},
    {19, 82}, {-3, 75}, {-1, 23}, {1, 34}, {1, 43}, {0, 54}, {-2, 55}, {0, 61}, {1, 64}, {0, 68}, {-9, 92}};
constexpr mn kInit402_435_idc0[34] = {
    {-4, 79}, {-7, 71}, {-5, 69}, {-9, 70}, {-8, 66}, {-10, 68}, {-19, 73}, {-12, 69}, {-16, 70}, {-15, 67}, {-20, 62}, {-19, 70},
    {-16, 66}, {-22, 65}, {-20, 63}, {9, -2}, {26, -9}, {33, -9}, {39, -7}, {41, -2}, {45, 3}, {49, 9}, {45, 27}, {36, 59}, {-6, 66},
    {-7, 35}, {-7, 42}, {-8, 45}, {-5, 48}, {-12, 56}, {-6, 60}, {-5, 62}, {-8, 66}, {-8, 76}};

// The (m, n) of context ctxIdx for an I slice (intra = true) or a P / B slice with cabac_init_idc 0; {0, 0} marks a
// context this build has no values for (the state then comes out as pStateIdx 62, valMPS 0: a stream that uses it
// will not parse and stays literal).  ctxIdx 460..1023 (4:4:4 Cb / Cr residuals, and the coded_block_flag of 8x8
// blocks) repeat the values of the luma ranges they mirror (Table 9-34's ctxIdxOffset columns).
inline mn init_pair(int ctx, bool intra) {
    auto pick = [&](const mn *i_tab, const mn *p_tab, int k) { return intra ? i_tab[k] : p_tab[k]; };
    if (ctx <= 10) return kInit0_10[ctx];
    if (ctx <= 59) return intra ? mn{0, 0} : kInit11_59_idc0[ctx - 11];
    if (ctx <= 69) return kInit60_69[ctx - 60];
    if (ctx <= 104) return pick(kInit70_104_I, kInit70_104_idc0, ctx - 70);
    if (ctx <= 165) return pick(kInit105_165_I, kInit105_165_idc0, ctx - 105);
    if (ctx <= 226) return pick(kInit166_226_I, kInit166_226_idc0, ctx - 166);
    if (ctx <= 275) return pick(kInit227_275_I, kInit227_275_idc0, ctx - 227);
    if (ctx <= 398) return mn{0, 0};                          // 276 is end_of_slice (no state); 277..398: field coded
    if (ctx <= 401) return pick(kInit399_401_I, kInit399_401_idc0, ctx - 399);
    if (ctx <= 435) return pick(kInit402_435_I, kInit402_435_idc0, ctx - 402);
    if (ctx <= 459) return mn{0, 0};                          // field coded 8x8
    if (ctx <= 471) return init_pair(85 + (ctx - 460), intra);                 // coded_block_flag, Cb cat 6..8  <- cat 0..2
    if (ctx <= 483) return init_pair(85 + (ctx - 472), intra);                 //                   Cr cat 10..12
    if (ctx <= 527) return init_pair(105 + (ctx - 484), intra);                // significant, Cb cat 6..8
    if (ctx <= 571) return init_pair(105 + (ctx - 528), intra);                //              Cr
    if (ctx <= 615) return init_pair(166 + (ctx - 572), intra);                // last, Cb
    if (ctx <= 659) return init_pair(166 + (ctx - 616), intra);                //       Cr
    if (ctx <= 674) return init_pair(402 + (ctx - 660), intra);                // significant 8x8, Cb (cat 9)
    if (ctx <= 689) return mn{0, 0};                                           //   field
    if (ctx <= 698) return init_pair(417 + (ctx - 690), intra);                // last 8x8, Cb
    if (ctx <= 707) return mn{0, 0};                                           //   field
    if (ctx <= 717) return init_pair(426 + (ctx - 708), intra);                // abs level 8x8, Cb
    if (ctx <= 732) return init_pair(402 + (ctx - 718), intra);                // significant 8x8, Cr (cat 13)
    if (ctx <= 747) return mn{0, 0};
    if (ctx <= 756) return init_pair(417 + (ctx - 748), intra);                // last 8x8, Cr
    if (ctx <= 765) return mn{0, 0};
    if (ctx <= 775) return init_pair(426 + (ctx - 766), intra);                // abs level 8x8, Cr
    if (ctx <= 951) return mn{0, 0};                                           // field coded significant / last, Cb and Cr
    if (ctx <= 981) return init_pair(227 + (ctx - 952), intra);                // abs level, Cb cat 6..8
    if (ctx <= 1011) return init_pair(227 + (ctx - 982), intra);               //            Cr
    return init_pair(93 + ((ctx - 1012) & 3), intra);                          // coded_block_flag of 8x8 blocks (cat 5, 9, 13) <- cat 2
}

// 9.3.1.1: the state byte (2 * pStateIdx + valMPS, libavcodec's form) of a context at the start of a slice
inline uint8_t init_state(int ctx, bool intra, int slice_qp) {
    const mn p = init_pair(ctx, intra);
    const int qp = slice_qp < 0 ? 0 : slice_qp > 51 ? 51 : slice_qp;
    int pre = ((p.m * qp) >> 4) + p.n;
    pre = pre < 1 ? 1 : pre > 126 ? 126 : pre;
    return pre <= 63 ? uint8_t(2 * (63 - pre)) : uint8_t(2 * (pre - 64) + 1);
}

// Table 9-43, frame coded blocks: ctxIdxInc of significant_coeff_flag and last_significant_coeff_flag by levelListIdx
constexpr uint8_t kSig8x8[63] = {0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7,
                                 7, 6, 11, 12, 13, 11, 6, 7, 8, 9, 14, 10, 9, 8, 6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12};
constexpr uint8_t kLast8x8[63] = {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2,
                                  3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};

// Tables 9-34 / 9-40 per ctxBlockCat 0..13: first context of coded_block_flag, significant_coeff_flag,
// last_significant_coeff_flag (frame coded) and coeff_abs_level_minus1
constexpr uint16_t kCbfBase[14] = {85, 89, 93, 97, 101, 1012, 460, 464, 468, 1016, 472, 476, 480, 1020};
constexpr uint16_t kSigBase[14] = {105, 120, 134, 149, 152, 402, 484, 499, 513, 660, 528, 543, 557, 718};
constexpr uint16_t kLastBase[14] = {166, 181, 195, 210, 213, 417, 572, 587, 601, 690, 616, 631, 645, 748};
constexpr uint16_t kAbsBase[14] = {227, 237, 247, 257, 266, 426, 952, 962, 972, 708, 982, 992, 1002, 766};

}  // namespace h264
}  // namespace avr
