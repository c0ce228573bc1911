// Host memory experiment of round 3 (the index builder's seed tables): first-touch time of 2 GB with and without
// MADV_HUGEPAGE, then 200 M random 8-byte reads over it from T threads. Usage: thp_probe [THREADS]
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 64;
  const size_t bytes = (size_t)2 << 30, n = bytes / 8;
  for (int mode = 0; mode < 2; ++mode) {
    double t0 = now();
    uint64_t *p = (uint64_t *)aligned_alloc(2 << 20, bytes);
    if (mode == 1) madvise(p, bytes, MADV_HUGEPAGE);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { memset((char *)p + bytes / T * t, 1, bytes / T); });
    for (auto &x : th) x.join();
    double t1 = now();
    th.clear();
    std::vector<uint64_t> sums(T);
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        uint64_t x = 88172645463325252ull + t, s = 0;
        for (size_t i = 0; i < 200000000ull / T; ++i) {
          x ^= x << 13; x ^= x >> 7; x ^= x << 17;
          s += p[x % n];
        }
        sums[t] = s;
      });
    for (auto &x : th) x.join();
    double t2 = now();
    printf("%s: first touch %.3f s, 200 M random reads on %d threads %.3f s (%llu)\n", mode ? "MADV_HUGEPAGE" : "4 KB pages   ", t1 - t0, T, t2 - t1, (unsigned long long)sums[0]);
    free(p);
  }
  return 0;
}
