// microbench_gather.hip — how many random 64-byte lines per second can MI355X serve from an L2-resident table
// when every lane runs a dependent chain (the access pattern of the FM-index LF step)?
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_gather tools/microbench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void chase(const uint4 *table, uint32_t n_lines, uint32_t steps, uint32_t lanes_per_line, uint32_t *out) {
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (gid * 2654435761u) % n_lines;
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; ++s) {
    const uint4 *p = table + (size_t)idx * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    uint32_t v = a.x ^ b.y ^ c.z ^ d.w;
    acc += v;
    idx = (idx * 1664525u + v + 1013904223u) % n_lines;  // next line depends on the loaded data
  }
  out[gid] = acc;
}

int main(int argc, char **argv) {
  size_t mb_list[] = {2, 8, 64, 512};
  for (size_t mb : mb_list) {
    uint32_t n_lines = (uint32_t)(mb * 1024 * 1024 / 64);
    std::vector<uint32_t> h(n_lines * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2246822519u + 12345u);
    uint4 *d;
    uint32_t *out;
    hipMalloc(&d, (size_t)n_lines * 64);
    hipMemcpy(d, h.data(), (size_t)n_lines * 64, hipMemcpyHostToDevice);
    uint32_t lanes = 1u << 20, steps = 134;
    hipMalloc(&out, lanes * 4);
    for (int block : {256}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      chase<<<lanes / block, block>>>(d, n_lines, steps, 1, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) chase<<<lanes / block, block>>>(d, n_lines, steps, 1, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 5;
      double lines = (double)lanes * steps;
      printf("table %4zu MB  block %d: %.3f ms per launch, %.1f G lines/s, %.2f TB/s of 64-B lines\n", mb, block, ms,
             lines / ms / 1e6, lines * 64 / ms / 1e9);
    }
    hipFree(d);
    hipFree(out);
  }
  return 0;
}
