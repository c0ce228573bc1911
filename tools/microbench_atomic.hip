// Random 32-bit atomic increments into a small table (the coverage accumulators' access pattern):
// agent scope (what atomicAdd emits; executed at the memory side on a multi-XCD part) against workgroup
// scope (executed in the issuing XCD's L2). hipcc --offload-arch=gfx950 -O3 -o microbench_atomic microbench_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCOPE>
__global__ void k_atomic(uint32_t *table, uint32_t mask, uint32_t per_lane) {
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  for (uint32_t i = 0; i < per_lane; ++i) {
    x = x * 1664525u + 1013904223u;
    uint32_t *p = table + ((x >> 8) & mask);
    if (SCOPE == 0) atomicAdd(p, 1u);
    else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int main() {
  const uint32_t words = 1u << 17;  // 512 KB
  uint32_t *d;
  hipMalloc(&d, words * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int scope = 0; scope < 2; ++scope) {
    for (uint32_t per_lane : {1u, 8u}) {
      hipMemset(d, 0, words * 4);
      const uint32_t lanes = 1u << 20;
      hipLaunchKernelGGL(k_atomic<0>, dim3(64), dim3(256), 0, 0, d, words - 1, 1u);  // warm-up
      hipDeviceSynchronize();
      hipMemset(d, 0, words * 4);
      hipEventRecord(a);
      if (scope == 0) hipLaunchKernelGGL(k_atomic<0>, dim3(lanes / 256), dim3(256), 0, 0, d, words - 1, per_lane);
      else hipLaunchKernelGGL(k_atomic<1>, dim3(lanes / 256), dim3(256), 0, 0, d, words - 1, per_lane);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      std::vector<uint32_t> h(words);
      hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost);
      unsigned long long sum = 0;
      for (auto v : h) sum += v;
      printf("%s scope, %u per lane: %.3f ms, %.1f G atomics/s, sum %llu (expected %llu)\n", scope ? "workgroup" : "agent",
             per_lane, ms, (double)lanes * per_lane / ms / 1e6, sum, (unsigned long long)lanes * per_lane);
    }
  }
  return 0;
}
