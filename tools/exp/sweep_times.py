"""Reads the five wall-clock stamps (100 MHz) each workgroup of the `times` variant of loss_sweep.hip leaves in the first 40 bytes of its
pair's gradient (tools/exp/build_variants.sh times ...; CD_AMD_LIB=tools/exp/variants/libcd_amd_times.so): kernel entry, after the pair
constants, loop entry, loop exit, after the epilogue.  Prints the means over the workgroups of a call, in microseconds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd import synthetic
from consistent_depth_amd.loss import consistency_loss as CL
dev = torch.device("cuda", 0)
H, W = 384, 224
base = synthetic.make_scene_batch(8, H, W, seed=99, noise_px=0.25)
for B in (256, 1024):
    rep = (B + 7) // 8
    t = lambda a: torch.tensor(a, device=dev).repeat((rep,) + (1,) * (a.ndim - 1))[:B].contiguous()
    x = (torch.log(t(base["depth"])) + 0.01 * torch.randn(B, 2, H, W, device=dev)).requires_grad_(True)
    flows, masks = [t(f) for f in base["flows"]], [t(m) for m in base["masks"]]
    intr, extr = t(base["intrinsics"]), t(base["extrinsics"])
    msum, twin = CL.mask_sums(masks[0], masks[1]), CL.tile_windows(flows, masks)
    xd = x.detach()
    for it in range(60):
        _, _, _, grad = CL._launch(xd, flows[0], flows[1], masks[0], masks[1], msum, twin, intr, extr, 1.0, 0.1, 1, True)
    torch.cuda.synchronize()
    g = grad.reshape(B, -1)[:, :10].contiguous().cpu().numpy().view(np.int64).reshape(B, 5).astype(np.float64) / 100.0   # us
    t0 = g[:, 0].min()
    d = g - t0
    print(f"B={B}: entry mean {d[:,0].mean():.2f} max {d[:,0].max():.2f} | constants {np.mean(g[:,1]-g[:,0]):.2f} | to loop {np.mean(g[:,2]-g[:,1]):.2f} | "
          f"loop mean {np.mean(g[:,3]-g[:,2]):.2f} min {np.min(g[:,3]-g[:,2]):.2f} max {np.max(g[:,3]-g[:,2]):.2f} | epilogue {np.mean(g[:,4]-g[:,3]):.2f} | "
          f"exit mean {d[:,4].mean():.2f} max {d[:,4].max():.2f}")
    lo = g[:, 3] - g[:, 2]
    print("   loop by pair type (8 plans):", [round(float(lo[i::8].mean()), 1) for i in range(8)])
