import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from digdriver_amd import parallel, _lib
from digdriver_amd.data_tools.genome import PackedGenome
from digdriver_amd.sequence_model import nb_model
dev = torch.device("cuda:0")
Rr, Cr, Wr, Br = 36_000, 37, 10_000, 50
n_chrom = 3; per = Rr // n_chrom
rng = np.random.default_rng(4)
wh = rng.integers(0, 2 ** 32, (per * Wr * n_chrom) // 8 + 2, dtype=np.uint64).astype(np.uint32) & np.uint32(0x33333333)
names = ["chr%d" % (i + 1) for i in range(n_chrom)]
genome = PackedGenome(names, np.arange(n_chrom, dtype=np.int64) * per * Wr, np.full(n_chrom, per * Wr, np.int64), wh)
chroms = np.repeat(names, per); starts = np.tile(np.arange(per, dtype=np.int64) * Wr, n_chrom)
mu_r, sg_r = rng.uniform(5, 45, (Cr, Rr)), rng.uniform(1, 7, (Cr, Rr))
M = 500_000
mci, msr, cor = rng.integers(0, n_chrom, M), rng.integers(0, per * Wr, M).astype(np.int64), rng.integers(0, Cr, M).astype(np.int32)
Sr = rng.uniform(0, 1e-2, (Cr, 64))
sh = parallel.ShardedTiles(genome, chroms, starts, starts + Wr, Sr, mu_r, sg_r, np.array(names)[mci], msr, msr + 1, cor, Br, dev, 0, 1)
sh.run(); torch.cuda.synchronize()
pv = sh.result["pval"].reshape(Cr, -1)
print("p == 1:", float((pv == 1.0).double().mean()), " distinct in row 0:", int(torch.unique(pv[0]).numel()), " nan:", int(torch.isnan(pv).sum()))
def T(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("get_q_vals_rows(route p)  ms", T(lambda: nb_model.get_q_vals_rows(pv)))
print("q_values_all              ms", T(lambda: sh.q_values_all()))
print("q_values_all single       ms", T(lambda: sh.q_values_all(), reps=1))
t = torch.arange(sh.n_tiles, device=dev)[None, :]
print("mask+sum                  ms", T(lambda: int((t < sh.result["n_valid"][:, None]).sum())))
print("empty 8.5 GB              ms", T(lambda: torch.empty(8_500_000_000, dtype=torch.uint8, device=dev)))
