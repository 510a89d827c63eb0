import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from consistent_depth_amd import _native
from consistent_depth_amd.ops import conv
lib = _native.lib()
N, Cin, Cout, H, W, ks = 2, 32, 32, 17, 31, 7
g = torch.Generator().manual_seed(1)
x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)
ref = torch.nn.functional.conv2d(x.double(), w.double(), None, padding=3)
pk = conv.pack_weights(w.cuda())
for pipe in (0, 1):
    for ty in (4, 8, 16):
        for cot in (1, 2):
            lib.cd_debug_set_conv_pipeline(pipe)
            y = conv.conv2d(x.cuda(), pk, Cin, Cout, ks, cfg=(ty, cot)).cpu().double()
            err = (y - ref).abs()
            bad = (err > 1e-4)
            msg = ""
            if bad.any():
                idx = bad.nonzero()
                msg = f" bad={int(bad.sum())} n={sorted(set(idx[:,0].tolist()))} co={sorted(set(idx[:,1].tolist()))[:20]} rows={sorted(set(idx[:,2].tolist()))} cols={sorted(set(idx[:,3].tolist()))}"
            print(f"pipe={pipe} ty={ty} cot={cot} maxerr={err.max().item():.3e}{msg}")
