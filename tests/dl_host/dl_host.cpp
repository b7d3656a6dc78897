// TEST INFRASTRUCTURE: host build of pclean_amd/csrc/dl_cell.h (the linear-space cell update of the unrestricted
// Damerau-Levenshtein kernel dl_seg_kernel) driven the way the kernel drives it — NSEG lanes per pair, each owning a
// segment of the columns, lane s one row behind lane s-1, the row state handed from lane to lane — so that the CPU suite
// can hold the recurrence AND the segment pipeline against the oracle's full-matrix Lowrance-Wagner DP
// (oracle/densities.h: dl_distance) without a GPU.  The product never loads this library.
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../oracle/densities.h"
#include "../../pclean_amd/csrc/dl_cell.h"
using pco::dl_distance;
using pco::osa_distance;

// the kernel's schedule for one pair: returns H[la][lb]
static int seg_distance(const uint16_t* a, int la, const uint16_t* b, int lb, int nseg, int seglen_cap) {
  if (la == 0) return lb;
  if (lb == 0) return la;
  int seglen = (lb + nseg - 1) / nseg;
  if (seglen_cap > seglen) seglen = seglen_cap;  // (a wave runs every pair to its longest segment: padded columns)
  std::vector<std::vector<uint32_t>> words(nseg, std::vector<uint32_t>(seglen));
  for (int s = 0; s < nseg; ++s)
    for (int c = 0; c < seglen; ++c) words[s][c] = dlz_word_row0(s * seglen + c + 1);
  std::vector<uint32_t> e1(nseg, 0), e2(nseg, 0), p1(nseg), p2(nseg);  // packed row states, as the kernel shuffles them
  std::vector<uint32_t> aim1(nseg, 0xffffffffu);
  const int steps = la + nseg - 1;
  for (int step = 1; step <= steps; ++step) {
    p1 = e1;  // every lane reads its left neighbour's state of the PREVIOUS step
    p2 = e2;
    for (int s = 0; s < nseg; ++s) {
      const int i = step - s;
      if (i < 1 || i > la) continue;
      DlzRow st = s == 0 ? dlz_row_start(i) : dlz_unpack(p1[s - 1], p2[s - 1]);
      const uint32_t ai = a[i - 1];
      for (int c = 0; c < seglen; ++c) {
        const int j = s * seglen + c + 1;
        const uint32_t bj = j <= lb ? b[j - 1] : 0xffffu;
        dlz_cell(st, words[s][c], i, j, ai, aim1[s], bj);
      }
      aim1[s] = ai;
      dlz_pack(st, e1[s], e2[s]);
    }
  }
  const int s_last = (lb - 1) / seglen, c_last = (lb - 1) % seglen;
  return (int)(words[s_last][c_last] & 255u);
}

extern "C" {
int dlh_distance(const uint16_t* a, int la, const uint16_t* b, int lb, int nseg, int seglen_cap) {
  return seg_distance(a, la, b, lb, nseg, seglen_cap);
}
int dlh_reference(const uint16_t* a, int la, const uint16_t* b, int lb) { return dl_distance(a, la, b, lb); }
int dlh_osa(const uint16_t* a, int la, const uint16_t* b, int lb) { return osa_distance(a, la, b, lb); }

// n random pairs over an alphabet of `alpha` symbols, lengths in [0, max_len]; half of them a mutated copy of the other
// (substitutions, insertions, deletions, adjacent and gapped transpositions).  Returns the number of mismatches against the
// full-matrix DP; *n_diff_osa = pairs on which the unrestricted distance differs from the restricted one (the cases that
// exercise the two transposition terms).
long dlh_fuzz(uint64_t seed, long n, int alpha, int max_len, int nseg, long* n_diff_osa) {
  uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
  auto rnd = [&]() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
  };
  long bad = 0, diff = 0;
  std::vector<uint16_t> a, b;
  for (long t = 0; t < n; ++t) {
    const int la = (int)(rnd() % (uint64_t)(max_len + 1));
    a.resize(la);
    for (int i = 0; i < la; ++i) a[i] = (uint16_t)(rnd() % (uint64_t)alpha);
    if (rnd() & 1) {
      const int lb = (int)(rnd() % (uint64_t)(max_len + 1));
      b.resize(lb);
      for (int j = 0; j < lb; ++j) b[j] = (uint16_t)(rnd() % (uint64_t)alpha);
    } else {
      b = a;
      const int edits = (int)(rnd() % 6);
      for (int e = 0; e < edits; ++e) {
        const int kind = (int)(rnd() % 5);
        const int n_b = (int)b.size();
        if (kind == 0 && n_b > 0) b[rnd() % n_b] = (uint16_t)(rnd() % (uint64_t)alpha);
        else if (kind == 1 && n_b < max_len) b.insert(b.begin() + (long)(rnd() % (uint64_t)(n_b + 1)), (uint16_t)(rnd() % (uint64_t)alpha));
        else if (kind == 2 && n_b > 0) b.erase(b.begin() + (long)(rnd() % (uint64_t)n_b));
        else if (kind == 3 && n_b > 1) {
          const int p = (int)(rnd() % (uint64_t)(n_b - 1));
          std::swap(b[p], b[p + 1]);
        } else if (kind == 4 && n_b > 2) {  // gapped transposition: swap two symbols, then insert / delete between them
          const int p = (int)(rnd() % (uint64_t)(n_b - 2));
          std::swap(b[p], b[p + 1]);
          if ((rnd() & 1) && n_b < max_len) b.insert(b.begin() + p + 1, (uint16_t)(rnd() % (uint64_t)alpha));
        }
      }
    }
    const int lb = (int)b.size();
    const int want = dl_distance(a.data(), la, b.data(), lb);
    const int cap = (rnd() & 3) == 0 ? (int)(rnd() % 7) + (lb + nseg - 1) / nseg : 0;
    const int got = seg_distance(a.data(), la, b.data(), lb, nseg, cap);
    if (got != want) ++bad;
    if (want != osa_distance(a.data(), la, b.data(), lb)) ++diff;
  }
  if (n_diff_osa) *n_diff_osa = diff;
  return bad;
}
}
