"""Streaming layers of one mc-hourglass step (N = 8 images of 384x224), scalar kernels of rounds 1-5 (cd_debug_set_layers_mode(1)) vs the
16-byte / LDS-band kernels of round 6: us per launch, algorithmic GB/s, and the bits compared.  python tools/exp/layers_stream_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.ops import layers

N = 8


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {0: 0.0, 1: 0.0}
# (name, low-res h, w, C, launches of this shape per step)
for (h, w, C, n) in [(192, 112, 64, 1), (96, 56, 128, 1), (48, 28, 256, 1), (24, 14, 256, 1)]:
    H, W = 2 * h, 2 * w
    lo = torch.randn(N, C, h, w, device="cuda"); hi = torch.randn(N, C, H, W, device="cuda"); do = torch.randn(N, C, H, W, device="cuda")
    out = torch.empty(N, C, H, W, device="cuda"); dlo = torch.zeros(N, C, h, w, device="cuda"); dhi = torch.zeros(N, C, H, W, device="cuda")
    cases = [
        ("upsample2x_add_fwd", lambda: layers.upsample2x_add_fwd(lo, 0, C, out, 0, hi=hi, lo_relu=True, hi_relu=True), lambda: out, (C * h * w + 2 * C * H * W) * 4 * N),
        ("upsample2x_bwd    ", lambda: layers.upsample2x_bwd(do, 0, dlo, 0, C, accumulate=False), lambda: dlo, (C * h * w + C * H * W) * 4 * N),
        ("add_slice (+=)    ", lambda: layers.add_slice(do, 0, dhi, 0, C, accumulate=True), None, 3 * C * H * W * 4 * N),
        ("avgpool2_fwd      ", lambda: layers.avgpool2_fwd(hi, 0, C, dlo, 0, in_relu=True), lambda: dlo, (C * h * w + C * H * W) * 4 * N),
        ("avgpool2_bwd (+=) ", lambda: layers.avgpool2_bwd(lo, 0, dhi, 0, C, accumulate=True), None, (C * h * w + 2 * C * H * W) * 4 * N),
        ("avgpool2_bwd (=)  ", lambda: layers.avgpool2_bwd(lo, 0, dhi, 0, C, accumulate=False), lambda: dhi, (C * h * w + C * H * W) * 4 * N),
    ]
    for name, fn, res, nbytes in cases:
        us, bits = {}, {}
        for mode in (1, 0):
            layers.set_layers_mode(mode)
            us[mode] = timed(fn)
            if res is not None:
                fn(); torch.cuda.synchronize(); bits[mode] = res().clone()
            tot[mode] += us[mode]
        layers.set_layers_mode(0)
        same = "" if res is None else ("bits equal" if torch.equal(bits[0].view(torch.int32), bits[1].view(torch.int32)) else "BITS DIFFER")
        print(f"{name} {h:3d}x{w:3d}->x2 C={C:3d}: scalar {us[1]:7.1f} us {nbytes / us[1] / 1e3:7.0f} GB/s | round 6 {us[0]:7.1f} us {nbytes / us[0] / 1e3:7.0f} GB/s  {same}")
print(f"sum of the lines: scalar {tot[1] / 1e3:.3f} ms, round 6 {tot[0] / 1e3:.3f} ms")
