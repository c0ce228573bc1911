// gmx_suffixsort.hip — the suffix array of the index build, pre-sorted on the GPU (round 3).
//
// The PRG text is DNA with variant markers: suffixes differ within a few dozen symbols unless the PRG has long repeats.
// Two rounds of one LSD radix sort each (hipCUB) order the suffixes by their first 24 symbols; what still ties after
// that — real repeats — is finished on the host with plain suffix comparisons (gmx_index.cpp, the same routine and the
// same comparison budget as the host's parallel sort; SA-IS when the budget is spent). The result is THE suffix array
// (unique: the text ends with a unique smallest sentinel), as `tests/test_device_build.py` checks byte for byte.
//
//   key K(i): up to 12 symbols from position i, 3 bits each — sentinel 0, bases 1..4, "marker" 5 — ending at (and
//   including) the first marker or the sentinel, zero-padded, then 28 bits holding that marker's value - 4 (0: none).
//   Order-preserving and prefix-complete: K(i) < K(j) implies suffix i < suffix j, and K(i) == K(j) implies the two
//   suffixes agree on their first len(i) == len(j) symbols. (Markers up to 2^28 + 3: 134 M sites; beyond, the host sorts.)
//   round 1: sort (K(i), i)                                   -> groups of equal keys
//   round 2: tied elements by (group, K(i + len(i)))           -> sort by the second key, then stable by the group's start
//   host:    runs still tied (equal group and second key)      -> comparisons
//
// HBM streaming and radix passes; no LDS code of its own, no MFMA. 3.13 G symbols: see HISTORY.md §5.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "gmx_index.h"

namespace gmx {
namespace {

#define SCK(x)                                                                                                       \
  do {                                                                                                               \
    hipError_t e_ = (x);                                                                                             \
    if (e_ != hipSuccess) throw std::runtime_error(std::string("device suffix sort: ") + #x + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class T>
struct SBuf {
  T *p = nullptr;
  size_t n = 0;
  SBuf() = default;
  SBuf(const SBuf &) = delete;
  SBuf &operator=(const SBuf &) = delete;
  ~SBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    SCK(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T)));
    n = count;
  }
};

constexpr int TPB = 256;
constexpr uint32_t KEY_SYMBOLS = 12;
inline unsigned grid_for(size_t n) { return (unsigned)std::min<size_t>((n + TPB - 1) / TPB, 1u << 22); }
#define GRID_STRIDE(i, n) for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

// the key of the suffix at `i` and the number of symbols it covers
__device__ __forceinline__ unsigned long long suffix_key(const uint32_t *text, size_t n, size_t i, uint32_t &len) {
  unsigned long long key = 0;
  uint32_t marker = 0;
  len = 0;
  for (uint32_t j = 0; j < KEY_SYMBOLS && i + j < n; ++j) {
    const uint32_t c = text[i + j];
    const uint32_t d = c == 0 ? 0u : (c <= 4 ? c : 5u);
    key = (key << 3) | d;
    ++len;
    if (d == 0 || d == 5) {
      marker = d == 5 ? c - 4u : 0u;
      break;
    }
  }
  key <<= 3 * (KEY_SYMBOLS - len);
  return (key << 28) | marker;
}

__global__ void __launch_bounds__(TPB) key1_kernel(const uint32_t *text, size_t n, unsigned long long *key, uint32_t *idx) {
  GRID_STRIDE(i, n) {
    uint32_t len;
    key[i] = suffix_key(text, n, i, len);
    idx[i] = (uint32_t)i;
  }
}

// tied[p] = 1 when the element at sorted position p shares its key with a neighbour (n + 1 entries, the last 0);
// start[p] = p at a group's first element, else 0 (a running maximum gives every element its group's start)
__global__ void __launch_bounds__(TPB) tie1_kernel(const unsigned long long *key, size_t n, uint32_t *tied, uint32_t *start) {
  GRID_STRIDE(p, n + 1) {
    if (p == n) {
      tied[p] = 0;
      continue;
    }
    const unsigned long long k = key[p];
    const bool prev = p > 0 && key[p - 1] == k, next = p + 1 < n && key[p + 1] == k;
    tied[p] = (prev || next) ? 1u : 0u;
    start[p] = prev ? 0u : (uint32_t)p;
  }
}

__global__ void __launch_bounds__(TPB) carry_max_kernel(uint32_t *first, uint32_t carry) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *first = max(*first, carry);
}

// the tied elements, compacted (at[] = exclusive sum of tied[]): position, group start, suffix, second key
__global__ void __launch_bounds__(TPB) key2_kernel(const uint32_t *text, size_t n, const uint32_t *idx, const uint32_t *tied, const uint32_t *group,
                                                   const uint32_t *at, uint32_t *t_pos, uint32_t *t_group, uint32_t *t_idx, unsigned long long *t_key) {
  GRID_STRIDE(p, n) {
    if (!tied[p]) continue;
    const uint32_t i = idx[p];
    uint32_t len, len2;
    (void)suffix_key(text, n, i, len);
    const unsigned long long k2 = (size_t)i + len < n ? suffix_key(text, n, (size_t)i + len, len2) : 0ull;
    const uint32_t t = at[p];
    t_pos[t] = (uint32_t)p;
    t_group[t] = group[p];
    t_idx[t] = i;
    t_key[t] = k2;
  }
}

__global__ void __launch_bounds__(TPB) iota_kernel(uint32_t *p, size_t n) {
  GRID_STRIDE(i, n) p[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(TPB) gather32_kernel(const uint32_t *src, const uint32_t *slot, size_t n, uint32_t *dst) {
  GRID_STRIDE(i, n) dst[i] = src[slot[i]];
}
__global__ void __launch_bounds__(TPB) gather64_kernel(const unsigned long long *src, const uint32_t *slot, size_t n, unsigned long long *dst) {
  GRID_STRIDE(i, n) dst[i] = src[slot[i]];
}
// the tied elements back into their positions, in (group, second key) order: the i-th of them belongs at the i-th tied position
__global__ void __launch_bounds__(TPB) scatter_kernel(const uint32_t *t_pos, const uint32_t *idx_sorted, size_t n_tied, uint32_t *idx) {
  GRID_STRIDE(t, n_tied) idx[t_pos[t]] = idx_sorted[t];
}
// still tied after both rounds: same group, same second key as the element before (bit p of the mask)
__global__ void __launch_bounds__(TPB) tie2_kernel(const uint32_t *t_pos, const uint32_t *group_sorted, const unsigned long long *key_sorted, size_t n_tied,
                                                   uint32_t *mask) {
  GRID_STRIDE(t, n_tied) {
    if (t == 0) continue;
    if (group_sorted[t] == group_sorted[t - 1] && key_sorted[t] == key_sorted[t - 1]) {
      const uint32_t p = t_pos[t];
      atomicOr(&mask[p >> 5], 1u << (p & 31));
    }
  }
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Scratch {
  SBuf<unsigned char> tmp;
  void *get(size_t bytes) {
    if (bytes > tmp.n) tmp.alloc(bytes + (bytes >> 3));
    return tmp.p;
  }
};

template <class K, class V>
void sort_pairs(Scratch &sc, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n, int end_bit) {
  size_t bytes = 0;
  SCK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit));
  SCK(hipcub::DeviceRadixSort::SortPairs(sc.get(bytes), bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit));
}

// hipCUB's scans count their items in an int: pieces of 2^30 with the running value carried over
void exclusive_sum(Scratch &sc, const uint32_t *in, uint32_t *out, size_t n) {
  const size_t piece = (size_t)1 << 30;
  uint32_t carry = 0;
  for (size_t off = 0; off < n; off += piece) {
    const int len = (int)std::min(piece, n - off);
    size_t bytes = 0;
    SCK(hipcub::DeviceScan::ExclusiveScan(nullptr, bytes, in + off, out + off, hipcub::Sum(), carry, len));
    SCK(hipcub::DeviceScan::ExclusiveScan(sc.get(bytes), bytes, in + off, out + off, hipcub::Sum(), carry, len));
    if (off + piece < n) {
      uint32_t last_sum = 0, last_in = 0;
      SCK(hipMemcpy(&last_sum, out + off + len - 1, 4, hipMemcpyDeviceToHost));
      SCK(hipMemcpy(&last_in, in + off + len - 1, 4, hipMemcpyDeviceToHost));
      carry = last_sum + last_in;
    }
  }
}
void running_max_in_place(Scratch &sc, uint32_t *v, size_t n) {
  const size_t piece = (size_t)1 << 30;
  uint32_t carry = 0;
  for (size_t off = 0; off < n; off += piece) {
    const int len = (int)std::min(piece, n - off);
    if (off) hipLaunchKernelGGL(carry_max_kernel, dim3(1), dim3(1), 0, nullptr, v + off, carry);
    size_t bytes = 0;
    SCK(hipcub::DeviceScan::InclusiveScan(nullptr, bytes, v + off, v + off, hipcub::Max(), len));
    SCK(hipcub::DeviceScan::InclusiveScan(sc.get(bytes), bytes, v + off, v + off, hipcub::Max(), len));
    if (off + piece < n) SCK(hipMemcpy(&carry, v + off + len - 1, 4, hipMemcpyDeviceToHost));
  }
}

// sa: the suffixes ordered by their first 24 symbols; tie_mask: bit p set = sa[p] still ties with sa[p - 1]
bool device_suffix_presort(const uint32_t *text, size_t n, uint32_t *sa, std::vector<uint32_t> &tie_mask) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    (void)hipGetLastError();
    return false;
  }
  const bool trace = getenv("GMX_BUILD_TRACE") != nullptr;
  const double t0 = now_s();
  SBuf<uint32_t> d_text;
  d_text.alloc(n);
  SCK(hipMemcpy(d_text.p, text, n * sizeof(uint32_t), hipMemcpyHostToDevice));
  Scratch sc;
  // ---- round 1: (K(i), i) ---------------------------------------------------------------------
  SBuf<uint32_t> idx;  // the order so far
  SBuf<uint32_t> tied, group, at;
  {
    SBuf<unsigned long long> key_in, key_out;
    SBuf<uint32_t> idx_in;
    key_in.alloc(n);
    key_out.alloc(n);
    idx_in.alloc(n);
    idx.alloc(n);
    hipLaunchKernelGGL(key1_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, d_text.p, n, key_in.p, idx_in.p);
    sort_pairs(sc, key_in.p, key_out.p, idx_in.p, idx.p, n, 64);
    key_in.release();
    idx_in.release();
    tied.alloc(n + 1);
    group.alloc(n + 1);
    at.alloc(n + 1);
    hipLaunchKernelGGL(tie1_kernel, dim3(grid_for(n + 1)), dim3(TPB), 0, nullptr, key_out.p, n, tied.p, group.p);
    SCK(hipDeviceSynchronize());
  }
  const double t1 = now_s();
  exclusive_sum(sc, tied.p, at.p, n + 1);
  running_max_in_place(sc, group.p, n);
  uint32_t n_tied32 = 0;
  SCK(hipMemcpy(&n_tied32, at.p + n, 4, hipMemcpyDeviceToHost));
  const size_t n_tied = n_tied32;
  SBuf<uint32_t> d_mask;
  d_mask.alloc((n + 31) / 32);
  SCK(hipMemset(d_mask.p, 0, ((n + 31) / 32) * sizeof(uint32_t)));
  // ---- round 2: the tied elements by (group, K(i + len(i))) --------------------------------------
  if (n_tied) {
    SBuf<uint32_t> t_pos, t_group, t_idx;
    SBuf<unsigned long long> t_key;
    t_pos.alloc(n_tied);
    t_group.alloc(n_tied);
    t_idx.alloc(n_tied);
    t_key.alloc(n_tied);
    hipLaunchKernelGGL(key2_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, d_text.p, n, idx.p, tied.p, group.p, at.p, t_pos.p, t_group.p, t_idx.p,
                       t_key.p);
    SCK(hipDeviceSynchronize());
    tied.release();
    group.release();
    at.release();
    SBuf<uint32_t> slot, s1, s2, g1, g2;
    {  // by the second key, carrying each element's slot
      SBuf<unsigned long long> key_sorted;
      slot.alloc(n_tied);
      s1.alloc(n_tied);
      key_sorted.alloc(n_tied);
      hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n_tied)), dim3(TPB), 0, nullptr, slot.p, n_tied);
      sort_pairs(sc, t_key.p, key_sorted.p, slot.p, s1.p, n_tied, 64);
      slot.release();
    }
    // then stable by the group's start
    g1.alloc(n_tied);
    g2.alloc(n_tied);
    s2.alloc(n_tied);
    hipLaunchKernelGGL(gather32_kernel, dim3(grid_for(n_tied)), dim3(TPB), 0, nullptr, t_group.p, s1.p, n_tied, g1.p);
    sort_pairs(sc, g1.p, g2.p, s1.p, s2.p, n_tied, 32);
    g1.release();
    s1.release();
    t_group.release();
    SBuf<uint32_t> idx_sorted;
    SBuf<unsigned long long> key_sorted;
    idx_sorted.alloc(n_tied);
    key_sorted.alloc(n_tied);
    hipLaunchKernelGGL(gather32_kernel, dim3(grid_for(n_tied)), dim3(TPB), 0, nullptr, t_idx.p, s2.p, n_tied, idx_sorted.p);
    hipLaunchKernelGGL(gather64_kernel, dim3(grid_for(n_tied)), dim3(TPB), 0, nullptr, t_key.p, s2.p, n_tied, key_sorted.p);
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(n_tied)), dim3(TPB), 0, nullptr, t_pos.p, idx_sorted.p, n_tied, idx.p);
    hipLaunchKernelGGL(tie2_kernel, dim3(grid_for(n_tied)), dim3(TPB), 0, nullptr, t_pos.p, g2.p, key_sorted.p, n_tied, d_mask.p);
    SCK(hipDeviceSynchronize());
  }
  const double t2 = now_s();
  SCK(hipMemcpy(sa, idx.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  tie_mask.resize((n + 31) / 32);
  SCK(hipMemcpy(tie_mask.data(), d_mask.p, tie_mask.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (trace)
    fprintf(stderr, "    device suffix pre-sort: %.2f s (upload + first 12 symbols %.2f s, %zu tied -> next 12 symbols %.2f s, copies back %.2f s)\n",
            now_s() - t0, t1 - t0, n_tied, t2 - t1, now_s() - t2);
  return true;
}

struct Registrar {
  Registrar() { g_device_suffix_presort = &device_suffix_presort; }
} registrar;

}  // namespace
}  // namespace gmx
