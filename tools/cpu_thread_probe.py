"""How the oracle's CPU step scales with the torch thread count on this host (bench.py cpu_baseline picks the best)."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = torch.Generator().manual_seed(0)
x = torch.randn(4096, 3072, generator=g); W = torch.randn(3072, 3072, generator=g)
q = torch.randn(1, 48, 4096, 64, generator=g)
for n in [int(a) for a in (sys.argv[1:] or [256, 128, 64, 32, 16])]:
    torch.set_num_threads(n)
    F.linear(x, W); F.scaled_dot_product_attention(q, q, q)
    t0 = time.perf_counter()
    for _ in range(3):
        F.linear(x, W)
    t1 = time.perf_counter()
    for _ in range(2):
        F.scaled_dot_product_attention(q, q, q)
    t2 = time.perf_counter()
    print(f"threads {n:4d}: linear {2 * 4096 * 3072 * 3072 * 3 / (t1 - t0) / 1e9:8.1f} GFLOP/s   sdpa {4 * 48 * 4096 * 4096 * 64 * 2 / (t2 - t1) / 1e9:8.1f} GFLOP/s", flush=True)
