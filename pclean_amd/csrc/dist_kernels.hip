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
// Kernel dl_pair_kernel (unrestricted DL, Lowrance–Wagner): thread per pair,
//   full (la+2)x(lb+2) matrix in a lane-interleaved global scratch.  Exact but
//   slow; used only when PCLEAN_DIST_DL is requested.
#include "ctx.h"

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

__global__ void lat_len_kernel(const int64_t* __restrict__ off, const int32_t* __restrict__ lat_ids, int n,
                               uint16_t* __restrict__ len) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) len[v] = (uint16_t)(off[lat_ids[v] + 1] - off[lat_ids[v]]);
}

int pclean_launch_dist(pclean_ctx* ctx, PairTable& pt, const int32_t* d_obs_ids, const int32_t* d_lat_ids,
                       int dist_mode) {
  const int max_lb = pt.max_lat_len, max_la = pt.max_obs_len;
  hipLaunchKernelGGL(lat_len_kernel, dim3((pt.n_lat + 255) / 256), dim3(256), 0, ctx->stream, ctx->off.p,
                     d_lat_ids, pt.n_lat, pt.lat_len.p);
  if (pt.n_obs == 0) {  // empty observed domain (every cell of the column missing): nothing to fill
    HIPCHK(ctx, hipGetLastError());
    return PCLEAN_OK;
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
