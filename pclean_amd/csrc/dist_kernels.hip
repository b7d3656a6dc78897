// AddTypos pair tables: D[u][v] = DamerauLevenshtein(obs string u, latent string v).
//
// Replaces the (String,String)-keyed memo dict of the reference
// (src/distributions/add_typos.jl:47,55-56): every unique (observed, latent)
// pair of a column is evaluated once, on the GPU, into a dense byte table that
// the scoring kernels gather from.
//
// Kernel osa_tile_kernel (restricted DL / optimal string alignment):
//   one workgroup = one observed string x T latent strings (one per lane).
//   The observed string is wave-uniform, so the outer DP loop never diverges;
//   the two live DP rows and the lane's latent string sit in LDS transposed
//   ([j][lane]) so every ds_read/ds_write of a wave hits 64 consecutive
//   16-bit slots (conflict-free).  d[i-2][j-2] for the transposition case is
//   carried in registers, so only two rows are kept.
// Kernel osa_bitpar_kernel (restricted DL, bit-parallel, below): the default for observed strings up to 256 symbols;
//   the DP-tile kernel above remains for longer ones.
// Kernel dl_lds_kernel (unrestricted DL, Lowrance–Wagner, byte matrix in LDS) and dl_pair_kernel (same recurrence,
//   full (la+2)x(lb+2) matrix in a lane-interleaved global scratch, for strings whose matrix does not fit in LDS).
#include <algorithm>
#include <numeric>

#include "ctx.h"
#include "dl_cell.h"

template <typename OutT>
__global__ void osa_tile_kernel(const uint16_t* __restrict__ sym, const int64_t* __restrict__ off,
                                const int32_t* __restrict__ obs_ids, const int32_t* __restrict__ lat_ids,
                                int n_lat, int max_lb, OutT* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int T = blockDim.x;
  const int t = threadIdx.x;
  const int u = blockIdx.y;
  const int v = blockIdx.x * T + t;
  const int W = max_lb + 1;
  uint16_t* rowA = smem;               // [W][T]
  uint16_t* rowB = rowA + (size_t)W * T;
  uint16_t* bT = rowB + (size_t)W * T;  // [max_lb][T]
  uint16_t* a_s = bT + (size_t)max_lb * T;

  const int64_t a0 = off[obs_ids[u]];
  const int la = (int)(off[obs_ids[u] + 1] - a0);
  for (int i = t; i < la; i += T) a_s[i] = sym[a0 + i];

  int lb = 0;
  if (v < n_lat) {
    const int64_t b0 = off[lat_ids[v]];
    lb = (int)(off[lat_ids[v] + 1] - b0);
    for (int j = 0; j < lb; ++j) bT[j * T + t] = sym[b0 + j];
  }
  for (int j = 0; j <= max_lb; ++j) rowA[j * T + t] = (uint16_t)j;
  __syncthreads();

  for (int i = 1; i <= la; ++i) {
    const uint16_t ai = a_s[i - 1];
    const uint16_t ai1 = i > 1 ? a_s[i - 2] : (uint16_t)0xffff;
    int left = i;       // cur[0]
    int diag = i - 1;   // prev[0]
    int pp1 = rowB[t];  // d[i-2][0]
    int pp2 = 0;
    uint16_t bjm1 = 0xfffe;
    rowB[t] = (uint16_t)i;
    for (int j = 1; j <= max_lb; ++j) {
      const int up = rowA[j * T + t];
      const int oldB = rowB[j * T + t];
      const uint16_t bj = bT[(j - 1) * T + t];
      int vmin = min(min(up + 1, left + 1), diag + (ai != bj ? 1 : 0));
      if (i > 1 && j > 1 && ai == bjm1 && ai1 == bj) vmin = min(vmin, pp2 + 1);
      rowB[j * T + t] = (uint16_t)vmin;
      diag = up;
      left = vmin;
      pp2 = pp1;
      pp1 = oldB;
      bjm1 = bj;
    }
    uint16_t* tmp = rowA;
    rowA = rowB;
    rowB = tmp;
  }
  if (v < n_lat) {
    int d = rowA[lb * T + t];
    if (sizeof(OutT) == 1 && d > 255) d = 255;
    out[(size_t)u * n_lat + v] = (OutT)d;
  }
}

template <typename OutT>
__global__ void dl_pair_kernel(const uint16_t* __restrict__ sym, const int64_t* __restrict__ off,
                               const int32_t* __restrict__ obs_ids, const int32_t* __restrict__ lat_ids,
                               int n_obs, int n_lat, int max_la, int max_lb, int n_symbols,
                               uint16_t* __restrict__ scratch, OutT* __restrict__ out) {
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = max_lb + 2;
  const size_t hcells = (size_t)(max_la + 2) * W;
  uint16_t* H = scratch;                       // [hcells][nthreads]
  uint16_t* da = scratch + hcells * nthreads;  // [n_symbols][nthreads]
#define HH(r, c) H[((size_t)(r) * W + (c)) * nthreads + tid]
  const int64_t npairs = (int64_t)n_obs * n_lat;
  for (int64_t p = tid; p < npairs; p += nthreads) {
    const int u = (int)(p / n_lat), v = (int)(p % n_lat);
    const int64_t a0 = off[obs_ids[u]], b0 = off[lat_ids[v]];
    const int la = (int)(off[obs_ids[u] + 1] - a0), lb = (int)(off[lat_ids[v] + 1] - b0);
    const int maxdist = la + lb;
    for (int s = 0; s < n_symbols; ++s) da[(size_t)s * nthreads + tid] = 0;
    HH(0, 0) = (uint16_t)maxdist;
    for (int i = 0; i <= la; ++i) {
      HH(i + 1, 0) = (uint16_t)maxdist;
      HH(i + 1, 1) = (uint16_t)i;
    }
    for (int j = 0; j <= lb; ++j) {
      HH(0, j + 1) = (uint16_t)maxdist;
      HH(1, j + 1) = (uint16_t)j;
    }
    for (int i = 1; i <= la; ++i) {
      int db = 0;
      const uint16_t ai = sym[a0 + i - 1];
      for (int j = 1; j <= lb; ++j) {
        const uint16_t bj = sym[b0 + j - 1];
        const int k = da[(size_t)bj * nthreads + tid];
        const int l = db;
        int cost = 1;
        if (ai == bj) {
          cost = 0;
          db = j;
        }
        int vv = HH(i, j) + cost;
        vv = min(vv, HH(i + 1, j) + 1);
        vv = min(vv, HH(i, j + 1) + 1);
        vv = min(vv, HH(k, l) + (i - k - 1) + 1 + (j - l - 1));
        HH(i + 1, j + 1) = (uint16_t)vv;
      }
      da[(size_t)ai * nthreads + tid] = (uint16_t)i;
    }
    int d = HH(la + 1, lb + 1);
    if (sizeof(OutT) == 1 && d > 255) d = 255;
    out[p] = (OutT)d;
  }
#undef HH
}

// ---- osa_bitpar_kernel: restricted Damerau-Levenshtein (optimal string alignment), bit-parallel ----------------
// Hyyro's bit-vector algorithm ("A bit-vector algorithm for computing Levenshtein and Damerau edit distances",
// Nordic J. Computing 2003 — published algorithm, restated): the column of the DP matrix is kept as vertical /
// horizontal delta bit-vectors over the PATTERN, one text character per step, ~25 word operations per 64 pattern
// positions instead of 64 DP cells.  Mapping: pattern = the observed string (wave-uniform: its match masks
// Peq[symbol] are built once per workgroup in LDS), text = one latent string per lane (staged once per workgroup
// in LDS, transposed [j][lane] -> conflict-free); a workgroup walks a chunk of observed strings over its 256 latent
// strings.  W = 64-bit words per pattern (observed strings up to 64 W symbols): the additions and shifts carry
// across words exactly like big-integer arithmetic.  d = la initially; +1 / -1 per column from bit la-1 of the
// horizontal deltas.  The 90-character MeasureName column that cost the DP-tile kernel 10 s takes ~0.2 s.
template <int W, typename SymT, typename OutT>
__global__ __launch_bounds__(256) void osa_bitpar_kernel(const uint16_t* __restrict__ sym,
                                                         const int64_t* __restrict__ off,
                                                         const int32_t* __restrict__ obs_ids,
                                                         const int32_t* __restrict__ lat_ids, int n_obs, int n_lat,
                                                         int max_lb, int n_symbols, int u_chunk,
                                                         OutT* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* peq = reinterpret_cast<uint64_t*>(smem_raw);                          // [n_symbols][W]
  SymT* bT = reinterpret_cast<SymT*>(peq + (size_t)n_symbols * W);                // [max_lb][256]
  const int t = threadIdx.x;
  const int v = blockIdx.x * 256 + t;
  int lb = 0;
  if (v < n_lat) {
    const int64_t b0 = off[lat_ids[v]];
    lb = (int)(off[lat_ids[v] + 1] - b0);
    for (int j = 0; j < lb; ++j) bT[(size_t)j * 256 + t] = (SymT)sym[b0 + j];
  }
  const int u0 = blockIdx.y * u_chunk, u1 = min(u0 + u_chunk, n_obs);
  for (int u = u0; u < u1; ++u) {
    const int64_t a0 = off[obs_ids[u]];
    const int la = (int)(off[obs_ids[u] + 1] - a0);
    __syncthreads();  // previous pattern's masks are no longer read
    for (int i = t; i < n_symbols * W; i += 256) peq[i] = 0ull;
    __syncthreads();
    for (int i = t; i < la; i += 256) {
      unsigned int* wlo = reinterpret_cast<unsigned int*>(&peq[(size_t)sym[a0 + i] * W + (i >> 6)]);
      atomicOr(wlo + ((i & 63) >> 5), 1u << (i & 31));
    }
    __syncthreads();
    if (v >= n_lat) continue;
    int score = la;
    if (la == 0) {
      score = lb;
    } else {
      uint64_t VP[W], VN[W], D0[W], PMp[W];
#pragma unroll
      for (int r = 0; r < W; ++r) {
        VP[r] = ~0ull;
        VN[r] = 0ull;
        D0[r] = 0ull;
        PMp[r] = 0ull;
      }
      const int top_w = (la - 1) >> 6;
      const uint64_t top_bit = 1ull << ((la - 1) & 63);
      for (int j = 0; j < lb; ++j) {
        const uint64_t* pm = peq + (size_t)bT[(size_t)j * 256 + t] * W;
        uint64_t carry_add = 0, carry_tr = 0, carry_hp = 1ull, carry_hn = 0;
#pragma unroll
        for (int r = 0; r < W; ++r) {
          const uint64_t PM = pm[r];
          const uint64_t X = (~D0[r]) & PM;                       // transposition (restricted: adjacent pairs)
          const uint64_t TR = ((X << 1) | carry_tr) & PMp[r];
          carry_tr = X >> 63;
          const uint64_t A = PM & VP[r];
          const uint64_t s1 = A + VP[r];
          const uint64_t s2 = s1 + carry_add;
          carry_add = (s1 < A ? 1ull : 0ull) | (s2 < s1 ? 1ull : 0ull);
          const uint64_t d0 = ((s2 ^ VP[r]) | PM | VN[r]) | TR;
          const uint64_t HP = VN[r] | ~(d0 | VP[r]);
          const uint64_t HN = d0 & VP[r];
          if (r == top_w) {
            score += (HP & top_bit) ? 1 : 0;
            score -= (HN & top_bit) ? 1 : 0;
          }
          const uint64_t HPs = (HP << 1) | carry_hp;
          const uint64_t HNs = (HN << 1) | carry_hn;
          carry_hp = HP >> 63;
          carry_hn = HN >> 63;
          VP[r] = HNs | ~(d0 | HPs);
          VN[r] = d0 & HPs;
          D0[r] = d0;
          PMp[r] = PM;
        }
      }
    }
    if (sizeof(OutT) == 1 && score > 255) score = 255;
    out[(size_t)u * n_lat + v] = (OutT)score;
  }
}

// ---- dl_lds_kernel: unrestricted Damerau-Levenshtein (Lowrance-Wagner), one pair per lane, matrix in LDS -------
// The full (la+2) x (lb+2) matrix is needed (the transposition term reaches back to an arbitrary earlier row), so
// it lives in LDS as bytes, transposed [cell][lane] (conflict-free), next to the per-lane "last row of symbol"
// table; the observed string is wave-uniform.  Strings whose matrix does not fit go to dl_pair_kernel (global
// scratch).  Distances are exact whenever they are < 255 (a clamped cell can only feed cells that are >= 255 too).
template <typename OutT>
__global__ __launch_bounds__(64) void dl_lds_kernel(const uint16_t* __restrict__ sym, const int64_t* __restrict__ off,
                                                    const int32_t* __restrict__ obs_ids,
                                                    const int32_t* __restrict__ lat_ids, int n_obs, int n_lat,
                                                    int max_la, int max_lb, int n_symbols, int u_chunk,
                                                    OutT* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x;
  const int Wd = max_lb + 2;
  unsigned char* H = smem_raw;                                          // [(max_la+2) * Wd][64]
  unsigned char* da = H + (size_t)(max_la + 2) * Wd * 64;               // [n_symbols][64]
  unsigned char* bS = da + (size_t)n_symbols * 64;                      // [max_lb][64] latent symbols (low byte)
  unsigned char* bS2 = bS + (size_t)max_lb * 64;                        // high byte
#define HH(r, c) H[((size_t)(r) * Wd + (c)) * 64 + t]
  const int v = blockIdx.x * 64 + t;
  int lb = 0;
  if (v < n_lat) {
    const int64_t b0 = off[lat_ids[v]];
    lb = (int)(off[lat_ids[v] + 1] - b0);
    for (int j = 0; j < lb; ++j) {
      const uint16_t c = sym[b0 + j];
      bS[(size_t)j * 64 + t] = (unsigned char)(c & 0xff);
      bS2[(size_t)j * 64 + t] = (unsigned char)(c >> 8);
    }
  }
  const int u0 = blockIdx.y * u_chunk, u1 = min(u0 + u_chunk, n_obs);
  for (int u = u0; u < u1; ++u) {
    if (v >= n_lat) continue;
    const int64_t a0 = off[obs_ids[u]];
    const int la = (int)(off[obs_ids[u] + 1] - a0);
    const int maxdist = min(la + lb, 255);
    for (int s = 0; s < n_symbols; ++s) da[(size_t)s * 64 + t] = 0;
    HH(0, 0) = (unsigned char)maxdist;
    for (int i = 0; i <= la; ++i) {
      HH(i + 1, 0) = (unsigned char)maxdist;
      HH(i + 1, 1) = (unsigned char)min(i, 255);
    }
    for (int j = 0; j <= lb; ++j) {
      HH(0, j + 1) = (unsigned char)maxdist;
      HH(1, j + 1) = (unsigned char)min(j, 255);
    }
    for (int i = 1; i <= la; ++i) {
      int db = 0;
      const uint16_t ai = sym[a0 + i - 1];
      for (int j = 1; j <= lb; ++j) {
        const uint16_t bj = (uint16_t)bS[(size_t)(j - 1) * 64 + t] | ((uint16_t)bS2[(size_t)(j - 1) * 64 + t] << 8);
        const int k = da[(size_t)bj * 64 + t];
        const int l = db;
        int cost = 1;
        if (ai == bj) {
          cost = 0;
          db = j;
        }
        int vv = HH(i, j) + cost;
        vv = min(vv, HH(i + 1, j) + 1);
        vv = min(vv, HH(i, j + 1) + 1);
        vv = min(vv, HH(k, l) + (i - k - 1) + 1 + (j - l - 1));
        HH(i + 1, j + 1) = (unsigned char)min(vv, 255);
      }
      da[(size_t)ai * 64 + t] = (unsigned char)i;
    }
    int d = HH(la + 1, lb + 1);
    out[(size_t)u * n_lat + v] = (OutT)d;
  }
#undef HH
}

// ---- dl_wave_kernel: unrestricted Damerau-Levenshtein (Lowrance-Wagner), one pair per WG lanes, row-synchronous ----
// The recurrence
//     M[i][j] = min( M[i-1][j-1] + [a_i != b_j],  M[i-1][j] + 1,  M[i][j-1] + 1,
//                    M[k-1][l-1] + (i-k-1) + 1 + (j-l-1) )      k = last row < i with a_k == b_j, l = last column < j with b_l == a_i
// is evaluated a ROW at a time by the WG lanes that share a pair (lane = column; strings longer than WG take C column
// chunks per lane): the three terms without a left neighbour give cand_j, and the dependence on M[i][j-1] is the min-plus
// prefix  M[i][j] = j + min_{j' <= j} (cand_j' - j')  — a log-step shuffle scan.  a_i is uniform over the pair's lanes, so
// l comes from a ballot (highest matching lane below mine, or the last match of an earlier chunk); k is a lane-local
// running value.  Only the transposition term reaches back to an arbitrary earlier row: the matrix is kept as bytes in
// LDS (one per pair), everything else lives in registers.  The LDS-matrix kernel above gives every LANE a whole matrix
// (one wavefront per workgroup, a few resident waves per CU) and the global-scratch one is slower still: on the hospital
// name columns (27 x 27 symbols) this kernel is >10x faster, on the 90-symbol measure names >50x.
// Values are exact whenever they are < 255 (a clamped cell only feeds cells that are >= 255 too), as above.
template <int WG, int C>
__global__ __launch_bounds__(256) void dl_wave_kernel(const uint16_t* __restrict__ sym, const int64_t* __restrict__ off,
                                                      const int32_t* __restrict__ obs_ids, const int32_t* __restrict__ lat_ids,
                                                      int n_obs, int n_lat, int max_la, int max_lb, int u_chunk,
                                                      uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int PW = 64 / WG;  // pairs per wavefront
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const int slot = lane / WG, sl = lane % WG;
  const int Wd = max_lb + 1;
  const size_t mat_bytes = ((size_t)(max_la + 1) * Wd + 15) & ~(size_t)15;
  unsigned char* M = smem_raw + (size_t)(wave * PW + slot) * mat_bytes;  // rows 1 .. la of this pair's matrix
  uint16_t* A = reinterpret_cast<uint16_t*>(smem_raw + (size_t)n_waves * PW * mat_bytes);  // the observed string
  const int v = (blockIdx.x * n_waves + wave) * PW + slot;
  const bool valid = v < n_lat;
  int lb = 0;
  uint16_t bj[C];
#pragma unroll
  for (int c = 0; c < C; ++c) bj[c] = 0xffffu;
  if (valid) {
    const int64_t b0 = off[lat_ids[v]];
    lb = (int)(off[lat_ids[v] + 1] - b0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = c * WG + sl;
      if (j < lb) bj[c] = sym[b0 + j];
    }
  }
  const unsigned long long segmask = WG == 64 ? ~0ull : (((1ull << (WG & 63)) - 1ull) << (slot * WG));
  const int u0 = blockIdx.y * u_chunk, u1 = min(u0 + u_chunk, n_obs);
  for (int u = u0; u < u1; ++u) {
    __syncthreads();  // (the previous observed string is no longer read)
    const int64_t a0 = off[obs_ids[u]];
    const int la = (int)(off[obs_ids[u] + 1] - a0);
    for (int i = tid; i < la; i += blockDim.x) A[i] = sym[a0 + i];
    __syncthreads();
    int prev[C], da[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      prev[c] = min(c * WG + sl + 1, 255);  // row 0: M[0][j] = j
      da[c] = 0;
    }
    for (int i = 1; i <= la; ++i) {
      const uint16_t ai = A[i - 1];
      int carry_left = min(i, 255);       // M[i][0]
      int diag_left = min(i - 1, 255);    // M[i-1][0]
      int l_carry = 0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int j = c * WG + sl + 1;
        const bool active = j <= lb;
        const bool match = active && bj[c] == ai;
        int up = __shfl_up(prev[c], 1, WG);
        if (sl == 0) up = diag_left;
        const int sub = up + (match ? 0 : 1);
        const int del = prev[c] + 1;
        const unsigned long long seg = (__ballot(match) & segmask) >> (slot * WG);
        const unsigned long long below = seg & ((1ull << sl) - 1ull);
        const int l = below ? c * WG + (63 - __clzll((long long)below)) + 1 : l_carry;
        const int k = da[c];
        int cand = min(sub, del);
        if (active && k >= 1 && l >= 1) {
          // M[k-1][l-1]: row 0 and column 0 are the borders, everything else is in LDS
          const int m = (k == 1) ? (l - 1) : (l == 1) ? (k - 1) : (int)M[(size_t)(k - 1) * Wd + (l - 1)];
          cand = min(cand, min(m, 255) + (i - k - 1) + 1 + (j - l - 1));
        }
        int w = active ? cand - j : (1 << 20);
#pragma unroll
        for (int o = 1; o < WG; o <<= 1) {
          const int t = __shfl_up(w, o, WG);
          if (sl >= o) w = min(w, t);
        }
        w = min(w, carry_left - c * WG);
        const int cur = min(w + j, 255);
        diag_left = __shfl(prev[c], WG - 1, WG);  // M[i-1][c*WG + WG] for the next chunk (read before prev moves on)
        carry_left = __shfl(cur, WG - 1, WG);     // M[i][c*WG + WG]
        if (seg) l_carry = c * WG + (63 - __clzll((long long)seg)) + 1;
        if (active) {
          M[(size_t)i * Wd + j] = (unsigned char)cur;
          prev[c] = cur;
          if (match) da[c] = i;
        }
      }
    }
    if (valid) {
      int d = la;  // lb == 0
      if (lb > 0) {
        const int c_last = (lb - 1) / WG, sl_last = (lb - 1) % WG;
        int r = 0;
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (c == c_last) r = prev[c];
        d = __shfl(r, sl_last, WG);
      }
      if (sl == 0) out[(size_t)u * n_lat + v] = (uint8_t)min(d, 255);
    } else {
      (void)__shfl(0, 0, WG);  // (keep the wave's shuffles convergent)
    }
  }
}

// ---- dl_seg_kernel: unrestricted Damerau-Levenshtein in linear space, NSEG lanes per pair ---------------------------------
// The recurrence of dl_cell.h (Zhao & Sahni's form of Lowrance-Wagner: one 32-bit word of state per column) lets a LANE walk
// its columns a row at a time with its state in LDS — no matrix, so the LDS a pair needs is 5 bytes per column instead of
// (la + 1) bytes, and a CU holds 16 waves where dl_wave_kernel's matrices left room for 4.  A pair's columns are cut into NSEG
// segments owned by NSEG consecutive lanes; lane s works on row (step - s) — one row behind its left neighbour, whose row
// state (dl_cell.h: DlzRow, two packed words) it takes over by shuffle — so every lane is busy in all but the NSEG - 1 steps
// the pipeline takes to fill and drain, whatever the length of the strings.  The latent strings arrive SORTED BY LENGTH
// (pclean_launch_dist): the pairs of a wave have the same number of columns to within a symbol, and a launch covers one length
// class (<= 32 NSEG symbols) with <= 32 columns per lane.  The observed string is uniform over the workgroup (LDS, read once per
// row).  Results go to tmp[u][p] in sorted order; dl_unpermute_kernel puts the columns back.
// tests/dl_host holds this schedule (same header, same segment pipeline) against the oracle's full-matrix DP on the CPU.
template <int NSEG, typename SymT>
__global__ __launch_bounds__(256) void dl_seg_kernel(const uint16_t* __restrict__ sym, const int64_t* __restrict__ off,
                                                     const int32_t* __restrict__ obs_ids,
                                                     const int32_t* __restrict__ lat_sorted,  // string ids of this class's pairs
                                                     int n_obs, int n_cls, int p0, int n_lat, int segcap, int max_la, int u_chunk,
                                                     uint8_t* __restrict__ tmp) {
  extern __shared__ __attribute__((aligned(16))) uint32_t dlz_smem[];
  constexpr int SPW = 4 / (int)sizeof(SymT);  // symbols per 32-bit word of the latent strings' LDS copy
  constexpr uint32_t SENT = sizeof(SymT) == 1 ? 0xffu : 0xffffu;  // a symbol no string holds (the host checks n_symbols)
  uint32_t* words = dlz_smem;                                     // [segcap][256]
  uint32_t* bs = words + (size_t)segcap * 256;                    // [segcap / SPW][256]
  uint16_t* A = reinterpret_cast<uint16_t*>(bs + (size_t)(segcap / SPW) * 256);
  const int tid = threadIdx.x;
  const int s = tid % NSEG;
  const int pl = blockIdx.x * (256 / NSEG) + tid / NSEG;  // pair of this class
  const bool valid = pl < n_cls;
  int lb = 0;
  int64_t b0 = 0;
  if (valid) {
    const int id = lat_sorted[pl];
    b0 = off[id];
    lb = (int)(off[id + 1] - b0);
  }
  // columns per lane, uniform over the wavefront: the longest pair's share, in whole groups of four
  int wseg = (lb + NSEG - 1) / NSEG;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) wseg = max(wseg, __shfl_xor(wseg, o, 64));
  wseg = min((wseg + 3) & ~3, segcap);
  const int col0 = s * wseg;  // this lane owns columns col0 + 1 .. col0 + wseg
  for (int c = 0; c < wseg; c += SPW) {
    uint32_t pk = 0;
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
      const int j = col0 + c + k + 1;
      const uint32_t ch = (valid && j <= lb) ? (uint32_t)sym[b0 + j - 1] : SENT;
      pk |= ch << (k * 8 * (int)sizeof(SymT));
    }
    bs[(size_t)(c / SPW) * 256 + tid] = pk;
  }
  const int s_last = lb > 0 ? (lb - 1) / wseg : 0, c_last = lb > 0 ? (lb - 1) % wseg : 0;
  const int u0 = blockIdx.y * u_chunk, u1 = min(u0 + u_chunk, n_obs);
  for (int u = u0; u < u1; ++u) {
    __syncthreads();  // (the previous observed string is no longer read)
    const int64_t a0 = off[obs_ids[u]];
    const int la = (int)(off[obs_ids[u] + 1] - a0);
    for (int i = tid; i < la; i += 256) A[i] = sym[a0 + i];
    __syncthreads();
    for (int c = 0; c < wseg; ++c) words[(size_t)c * 256 + tid] = dlz_word_row0(min(col0 + c + 1, 255));
    uint32_t aim1 = 0xffffffffu;
    uint32_t e1 = 0, e2 = 0;  // the row state after this lane's last column, packed (handed to lane s + 1)
    const int steps = la + NSEG - 1;
    for (int step = 1; step <= steps; ++step) {
      const int i = step - s;
      DlzRow st;
      if (NSEG > 1) {
        const uint32_t g1 = (uint32_t)__shfl_up((int)e1, 1, NSEG), g2 = (uint32_t)__shfl_up((int)e2, 1, NSEG);
        st = dlz_unpack(g1, g2);
      }
      if (s == 0) st = dlz_row_start(i);
      if (i >= 1 && i <= la) {
        const uint32_t ai = A[i - 1];
        for (int c = 0; c < wseg; c += 4) {
          uint32_t bw[4 / SPW];
#pragma unroll
          for (int q = 0; q < 4 / SPW; ++q) bw[q] = bs[(size_t)(c / SPW + q) * 256 + tid];
          uint32_t w[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = words[(size_t)(c + k) * 256 + tid];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t bj = (bw[k / SPW] >> ((k % SPW) * 8 * (int)sizeof(SymT))) & SENT;
            dlz_cell(st, w[k], i, col0 + c + k + 1, ai, aim1, bj);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) words[(size_t)(c + k) * 256 + tid] = w[k];
        }
        aim1 = ai;
        if (NSEG > 1) dlz_pack(st, e1, e2);
      }
    }
    if (valid && s == s_last) {
      int d = la;  // lb == 0
      if (lb > 0) d = la == 0 ? lb : (int)(words[(size_t)c_last * 256 + tid] & 255u);
      tmp[(size_t)u * n_lat + p0 + pl] = (uint8_t)d;
    }
  }
}
// out[u][v] = tmp[u][pos_of[v]]: the columns back in the order of the latent domain (four per thread: one word per store;
// the gathers stay inside one row of tmp)
__global__ __launch_bounds__(256) void dl_unpermute_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ pos_of,
                                                           int n_lat, uint8_t* __restrict__ out) {
  const int v4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (v4 >= n_lat) return;
  const uint8_t* row = tmp + (size_t)blockIdx.y * n_lat;
  uint8_t* orow = out + (size_t)blockIdx.y * n_lat;
  uint32_t pk = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (v4 + k < n_lat) pk |= (uint32_t)row[pos_of[v4 + k]] << (8 * k);
  if (v4 + 3 < n_lat && ((((size_t)blockIdx.y * n_lat) & 3) == 0)) {
    *reinterpret_cast<uint32_t*>(orow + v4) = pk;
  } else {
    for (int k = 0; k < 4 && v4 + k < n_lat; ++k) orow[v4 + k] = (uint8_t)(pk >> (8 * k));
  }
}

__global__ void lat_len_kernel(const int64_t* __restrict__ off, const int32_t* __restrict__ lat_ids, int n,
                               uint16_t* __restrict__ len) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) len[v] = (uint16_t)(off[lat_ids[v] + 1] - off[lat_ids[v]]);
}

// unrestricted DL through dl_seg_kernel: latent strings sorted by length on the host, one launch per length class, results
// un-permuted.  Returns 1 when it ran, 0 when the table is not its kind (strings too long, too many symbols), < 0 on error.
static int launch_dl_seg(pclean_ctx* ctx, PairTable& pt, const int32_t* d_obs_ids, const int32_t* h_lat_ids) {
  const int max_lb = pt.max_lat_len, max_la = pt.max_obs_len, n_lat = pt.n_lat;
  if (pt.elem_bytes != 1 || max_lb > DLZ_MAX_LEN || max_la > DLZ_MAX_LEN || !h_lat_ids || ctx->n_symbols > 65535) return 0;
  const bool sym8 = ctx->n_symbols <= 255;  // (0xff / 0xffff must be a symbol no string holds)
  std::vector<int32_t> len(n_lat), perm(n_lat), pos_of(n_lat), sorted_ids(n_lat);
  for (int v = 0; v < n_lat; ++v) len[v] = (int32_t)(ctx->h_off[h_lat_ids[v] + 1] - ctx->h_off[h_lat_ids[v]]);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) { return len[x] < len[y]; });
  for (int p = 0; p < n_lat; ++p) {
    pos_of[perm[p]] = p;
    sorted_ids[p] = h_lat_ids[perm[p]];
  }
  DevBuf<int32_t> d_sorted, d_pos;
  DevBuf<uint8_t> tmp;
  if (d_sorted.alloc(n_lat) || d_pos.alloc(n_lat) || tmp.alloc(std::max<size_t>((size_t)pt.n_obs * n_lat, 16)))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed (DL table build)");
  hipError_t e = hipMemcpyAsync(d_sorted.p, sorted_ids.data(), (size_t)n_lat * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pos.p, pos_of.data(), (size_t)n_lat * 4, hipMemcpyHostToDevice, ctx->stream);
  int rc = e == hipSuccess ? PCLEAN_OK : pclean_fail(ctx, PCLEAN_ERR_HIP, "copy failed: %s", hipGetErrorString(e));
  const int u_chunk = 32;
  // length classes: lengths <= 32 on one lane, <= 64 on two, <= 128 on four, the rest on eight (<= 32 columns per lane)
  int p = 0;
  while (p < n_lat && !rc) {
    const int l0 = len[perm[p]];
    const int nseg = l0 <= 32 ? 1 : l0 <= 64 ? 2 : l0 <= 128 ? 4 : 8;
    const int hi = nseg == 8 ? DLZ_MAX_LEN : 32 * nseg;
    int q = p;
    while (q < n_lat && len[perm[q]] <= hi) ++q;
    const int cls_max = len[perm[q - 1]];
    const int segcap = std::max(4, (((cls_max + nseg - 1) / nseg) + 3) & ~3);
    const size_t lds = (size_t)segcap * 256 * 4 + (size_t)segcap * 256 * (sym8 ? 1 : 2) + (size_t)std::max(max_la, 1) * 2 + 16;
    const int n_cls = q - p, ppw = 256 / nseg;
    dim3 grid((n_cls + ppw - 1) / ppw, (pt.n_obs + u_chunk - 1) / u_chunk);
#define LAUNCH_DLS(NS, ST)                                                                                               \
  do {                                                                                                                   \
    if (lds > 48 * 1024) {                                                                                               \
      hipError_t ea = hipFuncSetAttribute((const void*)dl_seg_kernel<NS, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)lds);                                                                     \
      if (ea != hipSuccess) rc = pclean_fail(ctx, PCLEAN_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(ea));       \
    }                                                                                                                    \
    if (!rc)                                                                                                             \
      hipLaunchKernelGGL((dl_seg_kernel<NS, ST>), grid, dim3(256), lds, ctx->stream, ctx->sym.p, ctx->off.p, d_obs_ids,  \
                         d_sorted.p + p, pt.n_obs, n_cls, p, n_lat, segcap, max_la, u_chunk, tmp.p);                     \
  } while (0)
#define LAUNCH_DLS_N(ST)      \
  do {                        \
    if (nseg == 1)            \
      LAUNCH_DLS(1, ST);      \
    else if (nseg == 2)       \
      LAUNCH_DLS(2, ST);      \
    else if (nseg == 4)       \
      LAUNCH_DLS(4, ST);      \
    else                      \
      LAUNCH_DLS(8, ST);      \
  } while (0)
    if (sym8)
      LAUNCH_DLS_N(uint8_t);
    else
      LAUNCH_DLS_N(uint16_t);
#undef LAUNCH_DLS_N
#undef LAUNCH_DLS
    p = q;
  }
  if (!rc) {
    hipLaunchKernelGGL(dl_unpermute_kernel, dim3((n_lat + 1023) / 1024, pt.n_obs), dim3(256), 0, ctx->stream, tmp.p, d_pos.p, n_lat,
                       (uint8_t*)pt.d.p);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (the host vectors and the scratch go away)
    if (e != hipSuccess) rc = pclean_fail(ctx, PCLEAN_ERR_HIP, "DL table build failed: %s", hipGetErrorString(e));
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  d_sorted.release();
  d_pos.release();
  tmp.release();
  return rc ? rc : 1;
}

int pclean_launch_dist(pclean_ctx* ctx, PairTable& pt, const int32_t* d_obs_ids, const int32_t* d_lat_ids,
                       int dist_mode, const int32_t* h_lat_ids) {
  const int max_lb = pt.max_lat_len, max_la = pt.max_obs_len;
  hipLaunchKernelGGL(lat_len_kernel, dim3((pt.n_lat + 255) / 256), dim3(256), 0, ctx->stream, ctx->off.p,
                     d_lat_ids, pt.n_lat, pt.lat_len.p);
  if (pt.n_obs == 0) {  // empty observed domain (every cell of the column missing): nothing to fill
    HIPCHK(ctx, hipGetLastError());
    return PCLEAN_OK;
  }
  static const bool no_bitpar = getenv("PCLEAN_NO_BITPAR") != nullptr;
  if (dist_mode == PCLEAN_DIST_OSA && max_la <= 256 && !no_bitpar) {
    // bit-parallel OSA: W words of 64 pattern (observed) positions
    const int W = std::max(1, (max_la + 63) / 64);
    const bool sym8 = ctx->n_symbols <= 255;
    const size_t lds = (size_t)ctx->n_symbols * W * 8 + (size_t)std::max(max_lb, 1) * 256 * (sym8 ? 1 : 2);
    if (lds <= 160 * 1024) {
      const int u_chunk = 64;
      dim3 grid((pt.n_lat + 255) / 256, (pt.n_obs + u_chunk - 1) / u_chunk);
#define LAUNCH_BITPAR(WW, ST, OT)                                                                                      \
  do {                                                                                                                 \
    HIPCHK(ctx, hipFuncSetAttribute((const void*)osa_bitpar_kernel<WW, ST, OT>,                                        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
    hipLaunchKernelGGL((osa_bitpar_kernel<WW, ST, OT>), grid, dim3(256), lds, ctx->stream, ctx->sym.p, ctx->off.p,     \
                       d_obs_ids, d_lat_ids, pt.n_obs, pt.n_lat, max_lb, ctx->n_symbols, u_chunk, (OT*)pt.d.p);        \
  } while (0)
#define LAUNCH_BITPAR_W(ST, OT)          \
  do {                                   \
    if (W == 1)                          \
      LAUNCH_BITPAR(1, ST, OT);          \
    else if (W == 2)                     \
      LAUNCH_BITPAR(2, ST, OT);          \
    else if (W == 3)                     \
      LAUNCH_BITPAR(3, ST, OT);          \
    else                                 \
      LAUNCH_BITPAR(4, ST, OT);          \
  } while (0)
      if (pt.elem_bytes == 1) {
        if (sym8)
          LAUNCH_BITPAR_W(uint8_t, uint8_t);
        else
          LAUNCH_BITPAR_W(uint16_t, uint8_t);
      } else {
        if (sym8)
          LAUNCH_BITPAR_W(uint8_t, uint16_t);
        else
          LAUNCH_BITPAR_W(uint16_t, uint16_t);
      }
#undef LAUNCH_BITPAR_W
#undef LAUNCH_BITPAR
      HIPCHK(ctx, hipGetLastError());
      return PCLEAN_OK;
    }
  }
  // unrestricted DL: the linear-space kernel (dl_seg_kernel); PCLEAN_DL_KERNEL=wave / lds / pair pick the older ones
  const char* dlk = getenv("PCLEAN_DL_KERNEL");
  if (dist_mode == PCLEAN_DIST_DL && (!dlk || !strcmp(dlk, "seg"))) {
    const int rs = launch_dl_seg(ctx, pt, d_obs_ids, h_lat_ids);
    if (rs < 0) return rs;
    if (rs > 0) return PCLEAN_OK;
  }
  if (dist_mode == PCLEAN_DIST_DL && pt.elem_bytes == 1 && max_lb <= 256 && max_la <= 255 && !getenv("PCLEAN_NO_DL_WAVE") &&
      (!dlk || !strcmp(dlk, "seg") || !strcmp(dlk, "wave"))) {
    // unrestricted DL, one pair per 16 / 32 / 64 lanes, a row at a time (dl_wave_kernel)
    const int WG = max_lb <= 16 ? 16 : max_lb <= 32 ? 32 : 64;
    const int C = WG < 64 ? 1 : (max_lb + 63) / 64;
    const int PW = 64 / WG;
    const size_t mat = (((size_t)(max_la + 1) * (max_lb + 1)) + 15) & ~(size_t)15;
    int n_waves = 4;
    while (n_waves > 1 && (size_t)n_waves * PW * mat + (size_t)max_la * 2 + 64 > 150 * 1024) n_waves >>= 1;
    const size_t lds = (size_t)n_waves * PW * mat + (size_t)std::max(max_la, 1) * 2 + 64;
    if (lds <= 150 * 1024) {
      const int u_chunk = 32;
      dim3 grid((pt.n_lat + n_waves * PW - 1) / (n_waves * PW), (pt.n_obs + u_chunk - 1) / u_chunk);
#define LAUNCH_DLW(WGv, Cv)                                                                                            \
  do {                                                                                                                 \
    HIPCHK(ctx, hipFuncSetAttribute((const void*)dl_wave_kernel<WGv, Cv>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                    (int)lds));                                                                        \
    hipLaunchKernelGGL((dl_wave_kernel<WGv, Cv>), grid, dim3(64 * n_waves), lds, ctx->stream, ctx->sym.p, ctx->off.p,  \
                       d_obs_ids, d_lat_ids, pt.n_obs, pt.n_lat, max_la, max_lb, u_chunk, (uint8_t*)pt.d.p);           \
  } while (0)
      if (WG == 16)
        LAUNCH_DLW(16, 1);
      else if (WG == 32)
        LAUNCH_DLW(32, 1);
      else if (C == 1)
        LAUNCH_DLW(64, 1);
      else if (C == 2)
        LAUNCH_DLW(64, 2);
      else if (C == 3)
        LAUNCH_DLW(64, 3);
      else
        LAUNCH_DLW(64, 4);
#undef LAUNCH_DLW
      HIPCHK(ctx, hipGetLastError());
      return PCLEAN_OK;
    }
  }
  if (dist_mode == PCLEAN_DIST_DL && pt.elem_bytes == 1 && !getenv("PCLEAN_NO_DL_LDS") && (!dlk || strcmp(dlk, "pair"))) {
    // unrestricted DL with the matrix in LDS (bytes), when it fits
    const size_t per_lane = (size_t)(max_la + 2) * (max_lb + 2) + (size_t)ctx->n_symbols + 2 * (size_t)std::max(max_lb, 1);
    const size_t lds = per_lane * 64;
    if (lds <= 160 * 1024 && max_la < 255 && ctx->n_symbols <= 65535) {
      const int u_chunk = 16;
      dim3 grid((pt.n_lat + 63) / 64, (pt.n_obs + u_chunk - 1) / u_chunk);
      HIPCHK(ctx, hipFuncSetAttribute((const void*)dl_lds_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
      hipLaunchKernelGGL(dl_lds_kernel<uint8_t>, grid, dim3(64), lds, ctx->stream, ctx->sym.p, ctx->off.p, d_obs_ids,
                         d_lat_ids, pt.n_obs, pt.n_lat, max_la, max_lb, ctx->n_symbols, u_chunk, (uint8_t*)pt.d.p);
      HIPCHK(ctx, hipGetLastError());
      return PCLEAN_OK;
    }
  }
  if (dist_mode == PCLEAN_DIST_OSA) {
    // LDS bytes = ((2*(max_lb+1) + max_lb) * T + max_la) * 2
    int T = 256;
    auto lds_bytes = [&](int t) { return ((size_t)(3 * max_lb + 2) * t + (size_t)max_la + 8) * 2; };
    while (T > 64 && lds_bytes(T) > 64 * 1024) T >>= 1;
    const size_t lds = lds_bytes(T);
    if (lds > 160 * 1024)
      return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "pair table strings too long for the OSA tile kernel (%d)",
                         max_lb);
    dim3 grid((pt.n_lat + T - 1) / T, pt.n_obs);
    if (pt.elem_bytes == 1) {
      HIPCHK(ctx, hipFuncSetAttribute((const void*)osa_tile_kernel<uint8_t>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(osa_tile_kernel<uint8_t>, grid, dim3(T), lds, ctx->stream, ctx->sym.p, ctx->off.p,
                         d_obs_ids, d_lat_ids, pt.n_lat, max_lb, (uint8_t*)pt.d.p);
    } else {
      HIPCHK(ctx, hipFuncSetAttribute((const void*)osa_tile_kernel<uint16_t>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(osa_tile_kernel<uint16_t>, grid, dim3(T), lds, ctx->stream, ctx->sym.p, ctx->off.p,
                         d_obs_ids, d_lat_ids, pt.n_lat, max_lb, (uint16_t*)pt.d.p);
    }
  } else {
    const int threads = 64;
    const int64_t npairs = (int64_t)pt.n_obs * pt.n_lat;
    int blocks = (int)std::min<int64_t>((npairs + threads - 1) / threads, 1024);
    const size_t per_thread = (size_t)(max_la + 2) * (max_lb + 2) + (size_t)ctx->n_symbols;
    // keep the scratch under 2 GiB
    while (blocks > 1 && per_thread * (size_t)blocks * threads * 2 > (2ull << 30)) blocks >>= 1;
    DevBuf<uint16_t> scratch;
    if (scratch.alloc(per_thread * (size_t)blocks * threads))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed for DL kernel");
    if (pt.elem_bytes == 1)
      hipLaunchKernelGGL(dl_pair_kernel<uint8_t>, dim3(blocks), dim3(threads), 0, ctx->stream, ctx->sym.p,
                         ctx->off.p, d_obs_ids, d_lat_ids, pt.n_obs, pt.n_lat, max_la, max_lb,
                         ctx->n_symbols, scratch.p, (uint8_t*)pt.d.p);
    else
      hipLaunchKernelGGL(dl_pair_kernel<uint16_t>, dim3(blocks), dim3(threads), 0, ctx->stream, ctx->sym.p,
                         ctx->off.p, d_obs_ids, d_lat_ids, pt.n_obs, pt.n_lat, max_la, max_lb,
                         ctx->n_symbols, scratch.p, (uint16_t*)pt.d.p);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    scratch.release();
  }
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
