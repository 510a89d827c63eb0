"""BatchNorm+ReLU backward (reduce + apply) on the plane sizes of one mc-hourglass step: us per call and GB/s (20 B per element)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.ops import layers

tot = 0.0
for (H, W, C, n) in [(384, 224, 208, 1), (384, 224, 64, 1), (192, 112, 128, 6), (192, 112, 96, 6), (96, 56, 160, 6), (96, 56, 192, 6),
                     (48, 28, 160, 5), (48, 28, 192, 5), (24, 14, 160, 3), (24, 14, 192, 3)]:
    N = 8
    dA = torch.randn(N, C, H, W, device="cuda"); x = torch.randn(N, C, H, W, device="cuda")
    mi = torch.rand(C, 2, device="cuda") + 0.5
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    sums = torch.zeros(C, 2, dtype=torch.float64, device="cuda")
    def run():
        layers.bn_relu_bwd(dA, 0, x, 0, C, mi, sums, scale=sc, shift=sh)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tot += us * n
    print(f"{H}x{W} C={C} x{n}: {us:8.1f} us  {20.0 * N * C * H * W / us / 1e3:7.1f} GB/s")
print(f"per step (these shapes): {tot / 1e3:.3f} ms  [CD_AMD_BN_PER_THREAD={os.environ.get('CD_AMD_BN_PER_THREAD', 'default')}]")
