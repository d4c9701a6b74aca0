"""Developer tool: pinned H2D / D2H bandwidth of this box (the ceiling of the MA_MEM_HOST route)."""
import time
import torch
n = 840 * 1024 * 1024
h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(name, round(n / dt / 1e9, 1), "GB/s")
# four streams at once
ss = [torch.cuda.Stream() for _ in range(4)]
q = n // 4
t = time.perf_counter()
for _ in range(5):
    for i, s in enumerate(ss):
        with torch.cuda.stream(s):
            d[i * q:(i + 1) * q].copy_(h[i * q:(i + 1) * q], non_blocking=True)
torch.cuda.synchronize()
print("H2D x4 streams", round(n / ((time.perf_counter() - t) / 5) / 1e9, 1), "GB/s")
hp = torch.empty(n, dtype=torch.uint8)
t = time.perf_counter()
d.copy_(hp); torch.cuda.synchronize()
print("H2D pageable", round(n / (time.perf_counter() - t) / 1e9, 1), "GB/s")
