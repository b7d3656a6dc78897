// Unrestricted Damerau-Levenshtein in LINEAR space: the cell update shared by the HIP kernel (dist_kernels.hip:
// dl_seg_kernel) and its host harness (tests/dl_host: the CPU suite holds it against the oracle's full-matrix
// Lowrance-Wagner DP on millions of string pairs; the product never loads the harness).
//
// What it computes is evaluate(DamerauLevenshtein(), a, b) of StringDistances.jl >= 0.11 (add_typos.jl:1,56), i.e. the
// Lowrance-Wagner recurrence
//     H[i][j] = min( H[i-1][j-1] + [a_i != b_j],  H[i-1][j] + 1,  H[i][j-1] + 1,
//                    H[k-1][l-1] + (i-k-1) + 1 + (j-l-1) )      k = last row < i with a_k == b_j, l = last column < j with b_l == a_i
// whose last term reaches back to an ARBITRARY earlier row: the full matrix has to stay (dl_wave_kernel keeps it in LDS, one
// matrix per pair: one resident wave per SIMD).  Zhao & Sahni ("String correction using the Damerau-Levenshtein distance",
// BMC Bioinformatics 2019 — published algorithm, restated) show that the term can only win when one of the two gaps is
// empty:
//     l == j-1 :  H[k-1][j-2] + (i-k)      needs, per COLUMN j, the value H[k-1][j-2] saved when row k matched b_j     (FR)
//     k == i-1 :  H[i-2][l-1] + (j-l)      needs, per ROW, the value H[i-2][l-1] saved when column l matched a_i        (T)
// so three numbers per column — H[i-1][j], H[i-2][j] and FR[j] (stored relative to its row: FR - k, so that the term is
// FR' + i) — and a handful per row are all the state there is: one 32-bit word per column.
//
// Word layout (kept in LDS by the kernel, one per column of the lane's segment):
//     bits  0..7   h1  = H[i-1][j]   (<= 254; the kernel takes strings of at most DLZ_MAX_LEN symbols)
//     bits  8..15  h2  = H[i-2][j]   (255: row -1, "infinite")
//     bits 16..31  frk = FR[j] - k + DLZ_BIAS, 0xffff: no earlier row matched b_j
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define DLZ_FN __host__ __device__ __forceinline__
#else
#define DLZ_FN inline
#endif

#define DLZ_MAX_LEN 254   // distances <= 254 fit the byte fields next to the sentinel 255
#define DLZ_BIG 4096      // "infinite" in the registers
#define DLZ_BIAS 512
#define DLZ_NO_FR 0xffffu

// Row state a lane carries along a row — and hands to the lane that owns the next column segment of the same pair.
// Values of H are <= 254 (DLZ_MAX_LEN); 255 stands for "infinite" wherever a value of H is kept (row -1, column -1): a term
// built on it is >= 255 and never wins against a real cell.
struct DlzRow {
  int left;      // H[i][j-1]
  int diag;      // H[i-1][j-1]
  int d2;        // H[i-1][j-2]
  int old_prev;  // H[i-2][j-1]
  int tl;        // T - l  (T = H[i-2][l-1] of the last column l < j with b_l == a_i; DLZ_BIG: none yet)
  int lcol;      // that column l (-1: none)
};

// state at the left border of row i (i >= 1): H[i][0] = i, H[i-1][0] = i-1, H[i-1][-1] = inf, H[i-2][0] = i-2 (row -1: inf)
DLZ_FN DlzRow dlz_row_start(int i) {
  DlzRow s;
  s.left = i;
  s.diag = i - 1;
  s.d2 = 255;
  s.old_prev = i >= 2 ? i - 2 : 255;
  s.tl = DLZ_BIG;
  s.lcol = -1;
  return s;
}

DLZ_FN uint32_t dlz_word_row0(int j) {  // before row 1: H[0][j] = j, H[-1][j] = inf, no FR
  return (uint32_t)j | (255u << 8) | (DLZ_NO_FR << 16);
}

DLZ_FN int dlz_min(int a, int b) { return a < b ? a : b; }

// One cell: row i (symbol ai, previous row's symbol aim1 or a value no symbol has), column j (symbol bj).  w = the column's
// word before the row; returns H[i][j] and leaves the word after the row in w, the state after the column in s.
// Branch-free (selects only): the lanes of a wavefront disagree on `match` in almost every cell.
DLZ_FN int dlz_cell(DlzRow& s, uint32_t& w, int i, int j, uint32_t ai, uint32_t aim1, uint32_t bj) {
  const int up = (int)(w & 255u);
  const int old = (int)((w >> 8) & 255u);
  const int frk = (int)(w >> 16);
  const bool match = bj == ai;
  int v = dlz_min(s.diag + (match ? 0 : 1), dlz_min(s.left, up) + 1);
  // l == j-1: the previous column matched a_i, and some earlier row k matched b_j (none: frk = 0xffff, the term is huge)
  const int ta = s.lcol == j - 1 ? frk + (i - DLZ_BIAS) : DLZ_BIG;
  // k == i-1: the previous row matched b_j, and some earlier column l of this row matched a_i (none: tl = DLZ_BIG)
  const int tb = bj == aim1 ? s.tl + j : DLZ_BIG;
  const int t = dlz_min(ta, tb);
  v = match ? v : dlz_min(v, t);
  const int frk2 = match ? s.d2 - i + DLZ_BIAS : frk;  // FR[j] = H[i-1][j-2] (k = i), kept as FR - k + bias
  s.tl = match ? s.old_prev - j : s.tl;                // T = H[i-2][j-1], l = j
  s.lcol = match ? j : s.lcol;
  s.d2 = s.diag;
  s.diag = up;
  s.old_prev = old;
  s.left = v;
  w = (uint32_t)dlz_min(v, 255) | ((uint32_t)up << 8) | ((uint32_t)frk2 << 16);
  return v;
}

// The row state in two words, as lane s hands it to lane s + 1 (values of H clamp at the sentinel 255; T - l fits 16 signed
// bits: |H - j| <= 255, or DLZ_BIG - j).
DLZ_FN void dlz_pack(const DlzRow& s, uint32_t& e1, uint32_t& e2) {
  e1 = (uint32_t)dlz_min(s.left, 255) | ((uint32_t)dlz_min(s.diag, 255) << 8) | ((uint32_t)dlz_min(s.d2, 255) << 16) |
       ((uint32_t)dlz_min(s.old_prev, 255) << 24);
  e2 = ((uint32_t)s.tl & 0xffffu) | ((uint32_t)(s.lcol + 1) << 16);
}
DLZ_FN DlzRow dlz_unpack(uint32_t e1, uint32_t e2) {
  DlzRow s;
  s.left = (int)(e1 & 255u);
  s.diag = (int)((e1 >> 8) & 255u);
  s.d2 = (int)((e1 >> 16) & 255u);
  s.old_prev = (int)(e1 >> 24);
  s.tl = (int)(int16_t)(e2 & 0xffffu);
  s.lcol = (int)(e2 >> 16) - 1;
  return s;
}
