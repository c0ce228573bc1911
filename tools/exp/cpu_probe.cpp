// How many cores does this container really get? T threads spin on a private counter for one second each; the sum
// of the counts per second, relative to one thread's, is the number of cores' worth of CPU time. Usage: cpu_probe
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
int main() {
  double one = 0;
  for (int T : {1, 16, 32, 64, 128, 256}) {
    std::vector<unsigned long long> cnt(T * 16, 0);
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        unsigned long long x = 1, n = 0;
        while (!stop.load(std::memory_order_relaxed)) {
          for (int i = 0; i < 1000; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull;
          ++n;
        }
        cnt[t * 16] = n + (x & 1);
      });
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    stop = true;
    for (auto &x : th) x.join();
    double sum = 0;
    for (int t = 0; t < T; ++t) sum += (double)cnt[t * 16];
    if (T == 1) one = sum;
    printf("%3d threads: %.1f cores' worth\n", T, sum / one);
  }
}
