"""Per-tensor distances of one inception block (engine, both conv arithmetics) and of torch fp32 autograd to fp64 autograd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from consistent_depth_amd import _native
from consistent_depth_amd.monodepth.hourglass import HourglassModel, INCEPTION, Inception
from consistent_depth_amd.monodepth.hourglass_engine import BlockRunner, HourglassEngine
from consistent_depth_amd.monodepth import hourglass as HG

def rel(a, b):
    return (a.double() - b.double()).abs().sum().item() / max(1e-30, b.double().abs().sum().item())

def run(kind, N, H, W):
    lib = _native.lib()
    torch.manual_seed(7)
    net = HourglassModel().cuda().train()
    mod = next(m for m in net.modules() if isinstance(m, HG.Inception) and m.kind == kind)
    c_in = INCEPTION[kind][0]
    sd0 = {k: v.cpu() for k, v in mod.state_dict().items()}
    ref = Inception(kind).double(); ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd0.items()}); ref.train()
    ref32 = Inception(kind); ref32.load_state_dict(sd0); ref32.train()
    g = torch.Generator().manual_seed(11)
    x_raw = torch.randn(N, c_in, H, W, generator=g, dtype=torch.float64)
    co = sum(c[-1] if len(c) > 1 else c[0] for c in INCEPTION[kind][1])
    wc = (0.5 + torch.rand(1, co, 1, 1, generator=g, dtype=torch.float64))
    a = torch.relu(x_raw).requires_grad_(True); yr = ref(a); (0.5 * (wc * yr * yr).sum()).backward()
    a32 = torch.relu(x_raw).float().requires_grad_(True); y32 = ref32(a32); (0.5 * (wc.float() * y32 * y32).sum()).backward()
    gref, g32 = dict(ref.named_parameters()), dict(ref32.named_parameters())
    out = {}
    for mode in (2, 0):
        lib.cd_set_conv_arith(mode)
        eng = HourglassEngine(net)
        for p in mod.parameters():
            p.grad = torch.zeros_like(p)
        blk = BlockRunner(eng, mod, N, H, W, relu_in=True)
        y = blk.forward(x_raw.float().cuda()); dx = blk.backward(wc.float().cuda() * y); torch.cuda.synchronize()
        out[mode] = {n: rel(p.grad.cpu(), gref[n].grad) for n, p in mod.named_parameters() if n.endswith(".weight")}
        out[mode]["dx"] = rel(dx.cpu(), a.grad)
    lib.cd_set_conv_arith(2)
    print(f"== {kind} {N}x{H}x{W}")
    for n in out[2]:
        t32 = rel(g32[n].grad, gref[n].grad) if n != "dx" else rel(a32.grad, a.grad)
        shape = tuple(gref[n].shape) if n != "dx" else ""
        print(f"  {n:18s} {str(shape):18s} split {out[2][n]:.2e}  fp32mfma {out[0][n]:.2e}  torch32 {t32:.2e}")

for case in [("A2", 2, 384, 224), ("A", 2, 384, 224), ("B2", 4, 192, 112), ("F", 8, 96, 56), ("E", 8, 48, 28)]:
    run(*case)
