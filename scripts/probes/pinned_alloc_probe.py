import time, torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for mb in (64, 256, 256, 64):
    t0 = time.perf_counter(); b = torch.empty(mb << 18, dtype=torch.int32, pin_memory=True); t1 = time.perf_counter()
    print(f"pinned {mb} MiB: {1e3*(t1-t0):.1f} ms"); del b
d = torch.zeros(64 << 20, dtype=torch.int32, device="cuda")
b = torch.empty(64 << 20, dtype=torch.int32, pin_memory=True)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); b.copy_(d, non_blocking=True); torch.cuda.synchronize(); print(f"D2H 256 MiB pinned: {1e3*(time.perf_counter()-t0):.1f} ms")
t0 = time.perf_counter(); h = d.cpu(); print(f"D2H 256 MiB pageable: {1e3*(time.perf_counter()-t0):.1f} ms")
