import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["CD_AMD_CONV_AUTOTUNE"] = "0"
os.environ["CD_AMD_ENGINE_STREAMS"] = "none"
import torch
from consistent_depth_amd import _native
from consistent_depth_amd.monodepth.hourglass import HourglassModel
from consistent_depth_amd.monodepth.hourglass_engine import HourglassEngine
lib = _native.lib()
torch.manual_seed(0)
net = HourglassModel().cuda().train()
N, H, W = 2, 64, 96
x = torch.rand(N, 3, H, W).cuda(); dpred = torch.randn(N, 1, H, W).cuda()
eng = HourglassEngine(net)

def run(ty):
    lib.cd_debug_force_conv_tile_rows(ty)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    pred = eng.forward(x)
    pred.backward(dpred)
    torch.cuda.synchronize()
    plan = eng._last
    snap = {"pred": plan["pred"].clone()}
    for i, a in enumerate(plan["acts"]):
        snap[f"act{i}.buf[{a.coff}:{a.coff + a.C}] {tuple(a.buf.shape)}"] = a.buf[:, a.coff:a.coff + a.C].clone()
        if a.gbuf is not None:
            snap[f"act{i}.gbuf[{a.coff}:{a.coff + a.C}] {tuple(a.buf.shape)}"] = a.gbuf[:, a.coff:a.coff + a.C].clone()
    for n, p in net.named_parameters():
        snap["grad " + n] = p.grad.clone()
    return snap

a = run(8); b = run(0); c = run(8)
for k in a:
    d = (a[k] - b[k]).abs().max().item(); d2 = (a[k] - c[k]).abs().max().item()
    s = a[k].abs().max().item()
    if d > 1e-5 * max(s, 1e-20) or d2 > 0:
        print(f"{k:70s} max|a|={s:.3e} diff(ty8,heur)={d:.3e} diff(ty8,ty8)={d2:.3e}")
