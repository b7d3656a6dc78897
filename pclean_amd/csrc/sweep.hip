// Enumeration / sweep entry points (stage-1 placeholder; filled in next).
#include "ctx.h"

void pclean_sweep_state_free(pclean_ctx* ctx) { (void)ctx; }

extern "C" int pclean_score_node(pclean_ctx* ctx, int32_t, int32_t, int32_t, const int32_t*, const int32_t*,
                                 const int32_t*, const double*, uint64_t, uint32_t, int32_t, double*, double*,
                                 int32_t*) {
  return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_score_node: not built yet");
}
extern "C" int pclean_sweep(pclean_ctx* ctx, const pclean_infer_config*, uint64_t, uint32_t, int32_t, const int32_t*,
                            int32_t*, int32_t*, double*) {
  return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_sweep: not built yet");
}
extern "C" int pclean_get_new_rows(pclean_ctx* ctx, int32_t, int32_t*, int32_t*, int32_t*) {
  return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_get_new_rows: not built yet");
}
extern "C" int pclean_stats_device_ptr(pclean_ctx* ctx, int32_t, void**, int64_t*) {
  return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_stats_device_ptr: not built yet");
}
extern "C" int pclean_get_timing(pclean_ctx* ctx, pclean_timing* out) {
  if (!ctx || !out) return PCLEAN_ERR_ARG;
  *out = ctx->timing;
  return PCLEAN_OK;
}
extern "C" int pclean_maybe_resample(pclean_ctx* ctx, int32_t, int32_t, const double*, int32_t, uint64_t, uint32_t,
                                     uint32_t, int32_t*, double*, double*) {
  return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_maybe_resample: not built yet");
}
extern "C" int pclean_final_choice(pclean_ctx* ctx, int32_t, int32_t, const double*, int32_t, int32_t, uint64_t,
                                   uint32_t, int32_t*, double*) {
  return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_final_choice: not built yet");
}

// ---- numeric-contract probes -------------------------------------------------
#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"

__global__ void debug_detmath_kernel(int n, const double* x, double* e, double* l, uint64_t* f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  e[i] = pclean_exp(x[i]);
  l[i] = pclean_log(x[i]);
  f[i] = pclean_fixw(x[i]);
}
__global__ void debug_rand64_kernel(int n, uint64_t seed, const uint32_t* rows, uint32_t site, uint32_t particle,
                                    uint32_t sweep, uint64_t* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pclean_rand64(seed, rows[i], site, particle, sweep);
}

extern "C" int pclean_debug_detmath(pclean_ctx* ctx, int32_t n, const double* x, double* exp_out, double* log_out,
                                    uint64_t* fixw_out) {
  if (!ctx || n <= 0 || !x || !exp_out || !log_out || !fixw_out) return pclean_fail(ctx, PCLEAN_ERR_ARG, "bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<double> dx, de, dl;
  DevBuf<uint64_t> df;
  if (dx.alloc(n) || de.alloc(n) || dl.alloc(n) || df.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "alloc");
  hipError_t e = hipMemcpy(dx.p, x, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(debug_detmath_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, dx.p, de.p, dl.p, df.p);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpy(exp_out, de.p, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(log_out, dl.p, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(fixw_out, df.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost);
  dx.release(); de.release(); dl.release(); df.release();
  if (e != hipSuccess) return pclean_fail(ctx, PCLEAN_ERR_HIP, "debug_detmath: %s", hipGetErrorString(e));
  return PCLEAN_OK;
}

extern "C" int pclean_debug_rand64(pclean_ctx* ctx, int32_t n, uint64_t seed, const uint32_t* rows, uint32_t site,
                                   uint32_t particle, uint32_t sweep, uint64_t* out) {
  if (!ctx || n <= 0 || !rows || !out) return pclean_fail(ctx, PCLEAN_ERR_ARG, "bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<uint32_t> dr;
  DevBuf<uint64_t> d_out;
  if (dr.alloc(n) || d_out.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "alloc");
  hipError_t e = hipMemcpy(dr.p, rows, n * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(debug_rand64_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, seed, dr.p, site,
                       particle, sweep, d_out.p);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost);
  dr.release(); d_out.release();
  if (e != hipSuccess) return pclean_fail(ctx, PCLEAN_ERR_HIP, "debug_rand64: %s", hipGetErrorString(e));
  return PCLEAN_OK;
}
