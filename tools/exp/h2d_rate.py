"""Raw H2D rate of the box: page-locked host memory -> HBM, copies of the bench's batch size back to back on one stream, and split
over two streams. Usage: python tools/exp/h2d_rate.py"""
import time, torch
for mb in (37.5, 75.0, 150.0, 9.375):
    n = int(mb * 1e6)
    h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
    d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for mode in ("one stream", "two streams"):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 40
            for i in range(K):
                st = s1 if mode == "one stream" or i % 2 == 0 else s2
                with torch.cuda.stream(st):
                    d[i % 4].copy_(h[i % 4], non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{mb:7.3f} MB x {K}, {mode}: {K * n / dt / 1e9:.1f} GB/s ({dt / K * 1e3:.3f} ms per copy)", flush=True)
