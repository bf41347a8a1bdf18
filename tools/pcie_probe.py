"""PCIe ceiling of the box: page-locked host <-> device copies of 128 MiB / 1 GiB through torch (one stream, and both directions at once)."""
import time, torch
dev = torch.device("cuda:0")
for mib in (128, 1024):
    n = mib << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    d2 = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        print(f"{mib} MiB {name}: {5*n/(time.perf_counter()-t)/1e9:.1f} GB/s")
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{mib} MiB both directions at once: {5*n/dt/1e9:.1f} GB/s each way")
