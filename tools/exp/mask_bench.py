"""cd_flow_consistency_masks on 256 pairs of 384x224 (12 floats / pixel / pair algorithmic: 2+2 flow, 3+3 colour, 2 masks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.utils import consistency
B, H, W = 256, 384, 224
g = torch.Generator(device="cuda").manual_seed(0)
yy, xx = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
if os.environ.get("MASK_FLOW", "smooth") == "smooth":     # camera-motion-like flow: a few pixels, varying slowly over the image
    f0 = torch.stack([3.3 + 2 * torch.sin(yy / 60), -1.7 + 1.5 * torch.cos(xx / 45)])[None].repeat(B, 1, 1, 1).contiguous()
    f0 += torch.randn(B, 2, 1, 1, device="cuda", generator=g)
    f1 = -f0 + torch.randn(B, 2, H, W, device="cuda", generator=g) * 0.3
else:                                                    # worst case: independent +-9 pixel offsets per pixel (scattered gathers)
    f0 = torch.randn(B, 2, H, W, device="cuda", generator=g) * 3; f1 = -f0 + torch.randn(B, 2, H, W, device="cuda", generator=g) * 0.5
c0 = torch.rand(B, 3, H, W, device="cuda", generator=g); c1 = torch.rand(B, 3, H, W, device="cuda", generator=g)
consistency.consistent_flow_masks_batch(f0, f1, c0, c1); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m = consistency.consistent_flow_masks_batch(f0, f1, c0, c1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{B} pairs {H}x{W}: {ms:.3f} ms  {48.0 * B * H * W / ms / 1e9:.2f} TB/s  kept {float(m[0].mean()):.3f}  [{os.environ.get('MASK_FLOW', 'smooth')} flow]")
