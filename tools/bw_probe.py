"""HBM ceilings on this box with plain torch ops: write-only (fill), read-only (sum), copy."""
import torch, time
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for gb in (0.25, 2.0):
    n = int(gb * 2**30 / 8)
    x = torch.empty(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    tf = t(lambda: x.fill_(1.5)); tc = t(lambda: y.copy_(x)); ts = t(lambda: x.sum())
    print(f"{gb} GiB: fill {n*8/tf/1e12:.2f} TB/s  copy(r+w) {2*n*8/tc/1e12:.2f} TB/s  sum(read) {n*8/ts/1e12:.2f} TB/s")
# floor for a launch that moves the bench workload's bytes (B = 10k: 4.72 MB in, 19.2 MB out)
n_out = 10_000 * 240
x = torch.empty(n_out, dtype=torch.float64, device="cuda")
src = torch.empty(10_000 * 59, dtype=torch.float64, device="cuda").normal_()
tf = t(lambda: x.fill_(1.5), 200)
y = torch.empty(10_000 * 59, dtype=torch.float64, device="cuda")
tc = t(lambda: (x.fill_(1.5), y.copy_(src)), 200)
print(f"10k-workload floor: fill 19.2 MB {tf*1e6:.2f} us/launch ({n_out*8/tf/1e12:.2f} TB/s); fill + 4.7 MB copy (2 launches) {tc*1e6:.2f} us")
