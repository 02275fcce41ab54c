"""Host-to-device and device-to-host copy rate of the box from pinned and from pageable memory (torch, 256 MiB buffers): the ceiling of every BH_MEM_HOST call.
usage (gpurun): python tools/pcie_rate.py"""
import time, json, torch
n = 256 << 20
dev = torch.device("cuda:0")
d = torch.empty(n, dtype=torch.uint8, device=dev)
out = {}
for name, h in (("pinned", torch.empty(n, dtype=torch.uint8).pin_memory()), ("pageable", torch.empty(n, dtype=torch.uint8).fill_(1))):
    for direction in ("h2d", "d2h"):
        for _ in range(2):
            (d.copy_(h, non_blocking=True) if direction == "h2d" else h.copy_(d, non_blocking=True)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            (d.copy_(h, non_blocking=True) if direction == "h2d" else h.copy_(d, non_blocking=True))
        torch.cuda.synchronize()
        out[f"{name}_{direction}_GBps"] = round(5 * n / (time.perf_counter() - t0) / 1e9, 2)
print(json.dumps(out))
