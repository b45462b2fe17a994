import torch, time
dev = torch.device("cuda:0")
rows, n = 37, 7_200_000
g = torch.Generator(device=dev).manual_seed(0)
src = torch.rand((rows, n), device=dev, generator=g, dtype=torch.float64)
out = torch.empty_like(src)
def T(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
base = torch.arange(n, device=dev)
for win in (n, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 13):
    # a permutation that only moves elements inside windows of `win` elements
    nb = (n + win - 1) // win
    key = torch.rand((rows, n), device=dev, generator=g) + (base // win)[None, :].float() * 2.0
    perm = torch.argsort(key, dim=1)
    del key
    t = T(lambda: out.scatter_(1, perm, src))
    print("window %9d elements (%7.1f MB): scatter of 8 B %6.2f ms   (gather %6.2f ms)" % (win, win * 8 / 1e6, t, T(lambda: torch.gather(src, 1, perm, out=out))))
    del perm
print("copy 8 B: %.2f ms" % T(lambda: out.copy_(src)))
