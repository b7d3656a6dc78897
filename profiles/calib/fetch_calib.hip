// FETCH_SIZE calibration for the access patterns of fk_root_wave_kernel (profiles/collect_r03.sh runs this under
// `rocprofv3 --pmc FETCH_SIZE`): kernels with a KNOWN number of bytes / distinct cache lines touched, over a buffer far
// larger than the 256 MiB Infinity Cache, so that the counter value per known byte can be read off per pattern.
//   stream16   : every lane loads 16 B, consecutive (the pre-filter scan's pattern)          -> bytes = N
//   gather_u8  : every lane loads ONE byte at a pseudo-random address (exact scoring's global_load_ubyte of a
//                survivor's distance / length byte)                                           -> lines = n_gathers
//   gather_u8x64: the 64 lanes of a wave load one byte each from 64 different 128-byte lines of ONE 16 KB row
//                (survivors of one byte row)                                                  -> lines = n_gathers
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void calib_stream16(const uint4* __restrict__ p, size_t n16, uint32_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n16; i += stride) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  return x ^ (x >> 33);
}

// one byte per lane at a random LINE of the buffer (every gather a different 128-byte line with high probability)
__global__ void calib_gather_u8(const uint8_t* __restrict__ p, size_t n_lines, size_t n_gathers, uint32_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n_gathers; i += stride) {
    const uint64_t h = mix(i * 0x9e3779b97f4a7c15ull + 1);
    acc += p[(h % n_lines) * 128 + ((h >> 40) & 127)];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// a wave gathers 64 bytes from 64 distinct lines of one 16 KB row (row chosen at random per wave iteration)
__global__ void calib_gather_row(const uint8_t* __restrict__ p, size_t n_rows, size_t n_wave_iters, uint32_t* __restrict__ out) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  uint32_t acc = 0;
  for (size_t it = wave; it < n_wave_iters; it += n_waves) {
    const uint64_t h = mix(it * 0x9e3779b97f4a7c15ull + 7);
    const size_t row = h % n_rows;
    acc += p[row * 16384 + (size_t)lane * 256 + ((h >> 40) & 127)];  // lane l -> line 2 l of the row
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// ---- stores and the particle kernels' patterns (round 4): does a coalesced 4- / 8- / 16-byte-per-lane store of FRESH lines
// make the L2 fetch them first?  Does a thread that walks its own 80 consecutive bytes (row-major draws, one row per
// lane, 20 loop iterations) fetch each line once?
template <typename T>
__global__ void calib_store(T* __restrict__ p, size_t n, T v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
// the shape of particle_update_kernel: thread i (one row) loops over P particles; reads a[i * P + p] (row-major), writes
// b[p * N + i] (4 B) and c[p * N + i] (8 B) particle-major
__global__ void calib_rowloop(const int32_t* __restrict__ a, int32_t* __restrict__ b, double* __restrict__ c, int N, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int p = 0; p < P; ++p) {
    const int d = a[(size_t)i * P + p];
    b[(size_t)p * N + i] = d;
    c[(size_t)p * N + i] = 0.0 + (double)d;
  }
}
__global__ void calib_rowloop_read(const int32_t* __restrict__ a, int32_t* __restrict__ out, int N, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int acc = 0;
  for (int p = 0; p < P; ++p) acc += a[(size_t)i * P + p];
  if (acc == 0x12345678) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)8 << 30;  // 8 GiB >> Infinity Cache
  uint8_t* buf;
  uint32_t* out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  const size_t n16 = ((size_t)2 << 30) / 16;                // stream 2 GiB
  const size_t n_gathers = (size_t)16 << 20;                // 16 Mi single-byte gathers
  const size_t n_wave_iters = ((size_t)16 << 20) / 64;      // 16 Mi bytes gathered, 64 per wave iteration
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_stream16, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)buf + ((size_t)rep << 27), n16, out);
    hipLaunchKernelGGL(calib_gather_u8, dim3(256 * 16), dim3(256), 0, 0, buf, bytes / 128, n_gathers, out);
    hipLaunchKernelGGL(calib_gather_row, dim3(256 * 16), dim3(256), 0, 0, buf, bytes / 16384, n_wave_iters, out);
  }
  CHECK(hipDeviceSynchronize());
  {  // round 4: stores of 1 GiB of fresh lines per flavour; the particle-update shape at N = 4 Mi rows, P = 20
    const size_t nb = (size_t)1 << 30;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(calib_store<uint32_t>, dim3(256 * 32), dim3(256), 0, 0, (uint32_t*)buf, nb / 4, 7u);
      hipLaunchKernelGGL(calib_store<uint64_t>, dim3(256 * 32), dim3(256), 0, 0, (uint64_t*)(buf + nb), nb / 8, (uint64_t)7);
      hipLaunchKernelGGL(calib_store<uint4>, dim3(256 * 32), dim3(256), 0, 0, (uint4*)(buf + 2 * nb), nb / 16, make_uint4(1, 2, 3, 4));
      const int N = 4 << 20, P = 20;
      int32_t* a = (int32_t*)(buf + 3 * nb);                       // 320 MiB read
      int32_t* b = (int32_t*)(buf + 4 * nb);                       // 320 MiB written
      double* c = (double*)(buf + 5 * nb);                         // 640 MiB written
      hipLaunchKernelGGL(calib_rowloop_read, dim3((N + 1023) / 1024), dim3(1024), 0, 0, a, (int32_t*)out, N, P);
      hipLaunchKernelGGL(calib_rowloop, dim3((N + 1023) / 1024), dim3(1024), 0, 0, a, b, c, N, P);
      CHECK(hipDeviceSynchronize());
    }
  }
  printf("calib: stream16 bytes %zu; gather_u8 gathers %zu (distinct 128-B lines ~ the same); gather_row gathers %zu\n",
         n16 * 16, n_gathers, n_wave_iters * 64);
  return 0;
}
