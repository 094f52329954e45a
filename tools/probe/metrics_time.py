import sys, time, torch
sys.path.insert(0, '/root/repo')
from gecco_amd import metrics
torch.manual_seed(0)
a = torch.randn(64, 2048, 3, device="cuda"); b = torch.randn(64, 2048, 3, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("chamfer ms", t(lambda: metrics.chamfer_distance(a, b)))
print("dist matrix ms", t(lambda: metrics.distance_matrix(a, b)))
print("sinkhorn ms", t(lambda: metrics.sinkhorn_emd(a, b), 2))
