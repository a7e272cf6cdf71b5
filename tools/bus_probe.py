"""Host <-> device copy rates of the box (page-locked and pageable memory, one direction and both at once): what bounds
the host-buffer path of run! (acme_batch_run(ACME_MEM_HOST)), which moves 8 (nu + ny) bytes per instance and sample."""
import time
import torch
n = 2 * 1024 ** 3 // 8
dev = torch.device("cuda:0")
d1 = torch.empty(n, dtype=torch.float64, device=dev)
d2 = torch.empty(n, dtype=torch.float64, device=dev)
hp = torch.empty(n, dtype=torch.float64).pin_memory()
hp2 = torch.empty(n, dtype=torch.float64).pin_memory()
hq = torch.empty(n, dtype=torch.float64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


gb = n * 8 / 1e9
print("H2D page-locked  %.1f GB/s" % (gb / t(lambda: d1.copy_(hp, non_blocking=True))))
print("D2H page-locked  %.1f GB/s" % (gb / t(lambda: hp.copy_(d1, non_blocking=True))))
print("H2D pageable     %.1f GB/s" % (gb / t(lambda: d1.copy_(hq))))
print("D2H pageable     %.1f GB/s" % (gb / t(lambda: hq.copy_(d1))))


def both():
    with torch.cuda.stream(s1):
        d1.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2):
        hp2.copy_(d2, non_blocking=True)


print("H2D + D2H at once, page-locked: %.1f GB/s each way" % (gb / t(both)))
