"""cd_warp_image on 512 frames of 384x224 (7 floats / pixel algorithmic: depth, C = 3 image in and out)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd import _native
N, C, H, W = 512, 3, 384, 224
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
images = torch.rand(N, C, H, W, device=dev, generator=g)
depths = 2.0 + torch.rand(N, H, W, device=dev, generator=g)
intr = torch.tensor([200.0, 200.0, W / 2, H / 2], device=dev).repeat(N, 1).contiguous()
extr = torch.eye(3, 4, device=dev).repeat(N, 1, 1).contiguous()
extr[:, 0, 3] = torch.arange(N, device=dev) * 0.01
tgt = ((torch.arange(N, device=dev) + 1) % N).int()
warped = torch.empty_like(images)
lib = _native.lib()
call = lambda: _native.check(lib.cd_warp_image(_native.dev_ptr(images), _native.dev_ptr(depths), _native.dev_ptr(intr), _native.dev_ptr(extr),
                                                tgt.data_ptr(), N, C, H, W, None, _native.dev_ptr(warped), _native.stream_ptr(dev)), "cd_warp_image")
for _ in range(5): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{N} frames {H}x{W}: {ms:.3f} ms  {28.0 * N * H * W / ms / 1e9:.2f} TB/s")
