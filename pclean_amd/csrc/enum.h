// Device-side views used by the enumeration kernels (internal).
#pragma once
#include "ctx.h"

#define PCLEAN_MAX_CHILDREN 16
#define PCLEAN_MAX_TERMS 16

struct TermDev {
  const int32_t* obs_col;   // [n_rows] observed value index, -1 = missing
  const int32_t* cand_col;  // [n_cand] latent value index of each candidate
  const uint8_t* pair;      // D[obs][lat], elem_bytes wide
  const uint16_t* lat_len;  // [n_lat]
  const int32_t* fn;        // ctx lookup table or nullptr
  int32_t n_lat, elem_bytes, dens_kind, max_typos, ctx_slot, fn_nb;
};

struct NodeDev {
  int32_t kind, n_cand, n_terms, pad;
  const int64_t* counts;
  const double* logc_full;
  const double* logc_m1;
  double scal[4];
  TermDev terms[PCLEAN_MAX_TERMS];
};

struct DensDev {
  const double* nb;
  const double* logl;
  int32_t nb_stride, pad;
};

// Work items of one enumeration launch. Item t scores evidence row row[t]
// (identity when null) under ctx[t][.], with candidate excl[t] having lost one
// reference; draws use particle id particle[t] (n_draws==1) or 0..n_draws-1.
struct ItemsDev {
  int32_t n, pad;
  const int32_t* row;
  const int32_t* ctx;
  const int32_t* excl;
  const int32_t* particle;
  int64_t row_offset;  // global id of local row 0 (RNG counter), multi-GPU sharding
};

// Log-marginals of the children of a "new row": either one value per item, or a
// per-unique-observed-value cache indexed through an observed column.
struct ChildrenDev {
  int32_t n, pad;
  const double* arr[PCLEAN_MAX_CHILDREN];
  const int32_t* obs_col[PCLEAN_MAX_CHILDREN];
  int32_t n_obs[PCLEAN_MAX_CHILDREN];
};

int pclean_launch_enum(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const ChildrenDev& ch, uint64_t seed,
                       uint32_t sweep, uint32_t site, int n_draws, double* lse_out, double* scores_out,
                       int32_t* draws_out);
