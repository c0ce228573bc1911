// microbench_gather_big.hip — dependent random 32-byte / 64-byte reads over tables of 0.5 .. 128 GB: what a lane's chain of
// scattered fetches costs once the index no longer fits the caches and the TLBs (configs[3]: 15 GB, configs[4]: 160 GB).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_gather_big tools/microbench_gather_big.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void fill(uint4 *t, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    t[i] = make_uint4((uint32_t)(i * 2246822519u + 12345u), (uint32_t)(i >> 7), (uint32_t)(i * 40503u), (uint32_t)i);
}
template <int QUADS>  // 16-byte pieces per access: 2 = a 32-byte text record, 4 = a 64-byte rank block / hit record
__global__ void chase(const uint4 *table, size_t n_units, uint32_t steps, uint32_t *out) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  size_t idx = ((size_t)gid * 2654435761ull) % n_units;
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; ++s) {
    const uint4 *p = table + idx * QUADS;
    uint32_t v = 0;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
      const uint4 a = p[q];
      v ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    acc += v;
    idx = (idx * 6364136223846793005ull + v + 1442695040888963407ull) % n_units;
  }
  out[gid] = acc;
}
int main() {
  const size_t gb_list[] = {1, 2, 8, 16, 64, 128};
  uint32_t *out;
  const uint32_t lanes = 1u << 20, steps = 64;
  hipMalloc(&out, lanes * 4);
  for (size_t gb : gb_list) {
    const size_t bytes = gb << 30;
    uint4 *d;
    if (hipMalloc(&d, bytes) != hipSuccess) {
      printf("table %3zu GB: allocation failed\n", gb);
      continue;
    }
    fill<<<4096, 256>>>(d, bytes / 16);
    hipDeviceSynchronize();
    for (int quads : {2, 4}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      const size_t n_units = bytes / (16 * quads);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (quads == 2) chase<2><<<lanes / 256, 256>>>(d, n_units, steps, out);
        else chase<4><<<lanes / 256, 256>>>(d, n_units, steps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("table %3zu GB, %2d-byte units: %8.3f ms for %u lanes x %u dependent steps = %6.1f G accesses/s, %.2f us per step of a lane's chain\n",
             gb, 16 * quads, ms, lanes, steps, (double)lanes * steps / (ms * 1e6), ms * 1e3 / steps);
    }
    hipFree(d);
  }
  return 0;
}
