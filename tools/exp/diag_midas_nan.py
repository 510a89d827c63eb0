"""Diagnostic: where does the MiDaS fine-tuning loss become non-finite at the bench shape (random init)?"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.engine import FineTuneStep
from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
from consistent_depth_amd.loaders.pair_store import PairStore

H, W, B = int(os.environ.get("DH", 224)), int(os.environ.get("DW", 384)), int(os.environ.get("DB", 4))
cls = get_depth_model("midas2")
params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=cls.lambda_view_baseline, lambda_parameter=0,
                            learning_rate=cls.learning_rate, optimizer="Adam")
model = cls(seed=0); model.train()
step = FineTuneStep(model, params, world=1)
store = PairStore.synthetic(20, H, W, seed=0, device="cuda")
with torch.no_grad():
    d0 = model.estimate_depth(store.color[:8])
    f = float(d0.median() / torch.as_tensor(store.gt_depth[:8]).float().median())
store.scale_scene_(f)
print("scene scale", f)
g = torch.Generator().manual_seed(0)
for it in range(int(os.environ.get("DSTEPS", 40))):
    ids = torch.randperm(len(store), generator=g)[:B].cuda()
    images, meta = store.batch(ids)
    raw = model.estimate_raw(images)
    step.opt.zero_grad()
    loss, parts = step.criterion(raw, meta, parameters=step._plist)
    loss.backward()
    fg = step.opt.flat_grad
    print(f"step {it}: raw finite={bool(torch.isfinite(raw).all())} [{float(raw.min()):.4g},{float(raw.max()):.4g}] loss={float(loss):.6g} "
          f"parts={ {k: [round(float(x), 5) for x in v] for k, v in parts.items()} } grad finite={bool(torch.isfinite(fg).all())} "
          f"|g|max={float(fg.abs().max()):.4g} nonfinite={int((~torch.isfinite(fg)).sum())}", flush=True)
    if not torch.isfinite(fg).all():
        off = 0
        for n, p in model.named_parameters():
            k = p.numel()
            bad = int((~torch.isfinite(p.grad)).sum()) if p.grad is not None else -1
            if bad: print("   non-finite grad:", n, tuple(p.shape), bad)
        break
    step.opt.step(grad_scale=1.0, guard_loss=loss.detach())
    pf = step.opt.flat_param if hasattr(step.opt, "flat_param") else None
    if pf is not None:
        print("   params finite:", bool(torch.isfinite(pf).all()), "absmax", float(pf.abs().max()))
