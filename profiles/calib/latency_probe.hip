// Loaded latency of DEPENDENT global loads on this GPU (what the root scan's per-group chain pays):
// every wave walks a random cycle through a table (one dependent load after the other), with 1 or 64 active lanes,
// for grids of 1 ... 5120 waves (the root scan's persistent grid: 1280 workgroups x 4 waves).
// build: hipcc --offload-arch=gfx950 -O3 -o latency_probe latency_probe.hip ; run: ./latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void chase(const unsigned int* __restrict__ next, unsigned int n, int steps, int lanes, unsigned int* out) {
  const int lane = threadIdx.x & 63;
  const unsigned int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (lane >= lanes) return;
  unsigned int i = (wave * 2654435761u + lane * 40503u) % n;
  for (int s = 0; s < steps; ++s) i = next[i];
  if (i == 0xffffffffu) out[0] = i;  // keep the chain
}

int main() {
  const size_t n = (size_t)256 << 20;  // 256 M entries = 1 GiB
  std::vector<unsigned int> h(n);
  std::iota(h.begin(), h.end(), 0u);
  std::mt19937_64 rng(1);
  for (size_t i = n - 1; i > 0; --i) {  // Sattolo: one cycle
    const size_t j = rng() % i;
    std::swap(h[i], h[j]);
  }
  unsigned int *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int steps = 2000;
  for (size_t span : {(size_t)256 << 20, (size_t)4 << 20}) {  // whole table / a 16 MiB window (cache-resident)
    for (int lanes : {1, 64}) {
      for (int wgs : {1, 64, 256, 1280}) {
        hipLaunchKernelGGL(chase, dim3(wgs), dim3(256), 0, 0, d, (unsigned int)span, 10, lanes, o);  // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(chase, dim3(wgs), dim3(256), 0, 0, d, (unsigned int)span, steps, lanes, o);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("start window %4zu MiB  lanes %2d  waves %5d : %7.1f ns per dependent load\n", span * 4 >> 20, lanes, wgs * 4,
               1e6 * ms / steps);
      }
    }
  }
  return 0;
}
