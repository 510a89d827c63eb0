#!/usr/bin/env python3
"""TEST / EVIDENCE TOOL (build container; no GPU).  The full-length parity run, three ways (VERDICT r05 "next round" item 1):

    python tools/parity_decompose.py a gpurun_out/snap384 [more product files ...] > profiles/parity_direct_r06.txt

Inputs: the golden of the spec (tests/golden/loop_*_384x224.npz: the fp64 continuation `T` of the burn-in state AND -- since round 6 --
the artefacts of the reference's own arithmetic `R`, fp32 on the CPU, oracle/gen_golden_loop_384.py ref32) and the product's artefacts
`P` of the same run (gpurun_out/snap384/product_<spec>.npz, written on the GPU box by `gen_golden_loop_384 snapshot`; any further
.npz files given are treated as more realisations, e.g. the configs[1] run).

  1. per epoch: P vs T | R vs T | P vs R  (relative L1; mean loss, per-pair losses, eval depth maps, checkpoint) -- the third block is
     the comparison BASELINE.json names ("matching the reference PyTorch CPU path").
  2. the depth-map distance split into a COMMON part |(P + R)/2 - T| (what every fp32 evaluation shares: a deterministic
     fp32-vs-fp64 term) and the SPREAD |P - R|/2 (what differs between two fp32 evaluations: amplified round-off).  Two independent
     realisations of equal noise give common ~ spread ~ each one's distance / sqrt(2); a common-mode term gives common >> spread.
  3. the checkpoint distance of the last epoch by parameter class (every 16th element of every tensor): convolution biases in front
     of an affine-less BatchNorm (mathematically ZERO gradient: in fp32 Adam normalises their round-off gradient to +-lr steps),
     convolution weights in front of a BatchNorm (scale-invariant), the stem's BatchNorm gamma / beta, the head, running statistics.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-300))


def sample_classes(stride):
    """class label of every element of `ckpt_sample` (oracle/gen_golden_loop_384._ckpt_sample: floating tensors of the state dict in
    order, `uncertainty` excluded, flat[::stride])."""
    import torch
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    torch.manual_seed(0)
    sd = HourglassModel().state_dict()
    labels = []
    for k, v in sd.items():
        if not v.is_floating_point() or "uncertainty" in k:
            continue
        n = len(range(0, v.numel(), stride))
        if k.startswith("pred_layer"):
            c = "head (3x3 64->1, weight + bias)"
        elif k.startswith("seq.0."):
            c = "stem conv bias (pre-BN, zero gradient)" if k.endswith("bias") else "stem conv weight (pre-BN)"
        elif k.startswith("seq.1."):
            c = "stem BN gamma / beta" if k.endswith(("weight", "bias")) else "BN running mean / var"
        elif k.endswith("running_mean") or k.endswith("running_var"):
            c = "BN running mean / var"
        elif k.endswith("bias"):
            c = "conv biases before affine-less BN (zero gradient)"
        else:
            c = "conv weights before affine-less BN"
        labels += [c] * n
    return np.array(labels)


def main():
    from oracle import gen_golden_loop_384 as G
    spec, src = sys.argv[1], sys.argv[2]
    S = G.SPECS[spec]
    z = np.load(G.golden_path(spec))
    assert "ref32_ckpt_sample" in z.files, "the golden has no ref32 artefacts yet: python -m oracle.gen_golden_loop_384 ref32 <spec> <dir>"
    runs = {"product": np.load(os.path.join(src, f"product_{spec}.npz"))}
    for f in sys.argv[3:]:
        runs[os.path.splitext(os.path.basename(f))[0]] = np.load(f)
    epochs = [int(e) for e in z["epochs"]]
    R = {k[len("ref32_"):]: z[k] for k in z.files if k.startswith("ref32_")}
    cols = ("mean", "perpair", "evaldepth", "ckpt")
    print(f"# clip '{spec}' ({S['clip']['n_frames']} frames 384x224, K = {S['K']} burn-in epochs on the GPU, epochs {epochs[0]}..{epochs[-1]} compared).")
    print("# T = fp64 continuation of the burn-in state, R = the reference's own arithmetic (fp32, torch CPU kernels) from the same state,")
    print("# P = the product (HIP engine) re-run from the seeds (burn-in state bitwise the golden's).  Relative L1.")
    for name, P in runs.items():
        pe = [e for e in epochs if f"val_e{e}_mean" in P.files]
        a, b, c = G.distances(P, z, pe), G.distances(R, z, pe), G.distances(P, z, pe, prefix="ref32_")
        print(f"\n## 1. {name}: per epoch")
        print("# epoch | P vs T: " + " ".join(f"{x:>10s}" for x in cols) + " | R vs T: " + " ".join(f"{x:>10s}" for x in cols) +
              " | P vs R (BASELINE's comparison): " + " ".join(f"{x:>10s}" for x in cols))
        for e in pe:
            print(f"  {e:5d} | " + " ".join(f"{a[e][x]:10.3e}" for x in cols) + " | " + " ".join(f"{b[e][x]:10.3e}" for x in cols) + " | " +
                  " ".join(f"{c[e][x]:10.3e}" for x in cols))
        worst = {x: max(c[e][x] for e in pe) for x in cols}
        first_over = {x: next((e for e in pe if c[e][x] > 1e-3), None) for x in cols}
        print("#  P vs R, worst over the run: " + "  ".join(f"{x} {worst[x]:.3e}" for x in cols))
        print("#  P vs R, first epoch above 1e-3: " + "  ".join(f"{x} {first_over[x]}" for x in cols))
        if "depth" in P.files and "depth" in R:
            print(f"#  final depth/frame_*.raw export: P vs T {rel(P['depth'], z['depth']):.3e}   R vs T {rel(R['depth'], z['depth']):.3e}   P vs R {rel(P['depth'], R['depth']):.3e}")
        print(f"\n## 2. {name}: eval depth maps, common mode vs spread of the two fp32 evaluations")
        print("# epoch |   P vs T     R vs T   | common |(P+R)/2 - T|   spread |P - R|/2   common / spread")
        for e in pe:
            p_, r_, t_ = (np.asarray(x[f"evaldepth_e{e}"], np.float64) for x in (P, R, z))
            den = np.abs(t_).sum()
            com, spr = np.abs((p_ + r_) / 2 - t_).sum() / den, np.abs(p_ - r_).sum() / 2 / den
            print(f"  {e:5d} | {rel(p_, t_):10.3e} {rel(r_, t_):10.3e} | {com:22.3e} {spr:18.3e} {com / max(spr, 1e-300):17.2f}")
        if "ckpt_sample" in P.files:
            labels = sample_classes(int(z["ckpt_stride"]))
            p_, r_, t_ = (np.asarray(x["ckpt_sample"], np.float64) for x in (P, R, z))
            assert len(labels) == len(t_), (len(labels), len(t_))
            print(f"\n## 3. {name}: checkpoint of the last epoch by parameter class (every {int(z['ckpt_stride'])}th element)")
            print(f"# {'class':52s} {'elements':>9s} {'sum|T|':>11s} {'P vs T':>10s} {'R vs T':>10s} {'P vs R':>10s}  share of sum|P - T|")
            tot = np.abs(p_ - t_).sum()
            for cls in sorted(set(labels), key=lambda s_: -np.abs(p_ - t_)[labels == s_].sum()):
                m = labels == cls
                print(f"  {cls:52s} {int(m.sum()):9d} {np.abs(t_[m]).sum():11.4e} {rel(p_[m], t_[m]):10.3e} {rel(r_[m], t_[m]):10.3e} {rel(p_[m], r_[m]):10.3e}  "
                      f"{np.abs(p_ - t_)[m].sum() / tot:6.1%}")
            print(f"  {'all':52s} {len(t_):9d} {np.abs(t_).sum():11.4e} {rel(p_, t_):10.3e} {rel(r_, t_):10.3e} {rel(p_, r_):10.3e}")
            # how many steps of size lr would explain the bias walk: 200 steps of +-lr (Adam at the noise floor) vs the measured mean |delta|
            mb = labels == "conv biases before affine-less BN (zero gradient)"
            print(f"#  mean |P - T| of those biases {np.abs(p_ - t_)[mb].mean():.3e}, mean |R - T| {np.abs(r_ - t_)[mb].mean():.3e}, mean |P - R| {np.abs(p_ - r_)[mb].mean():.3e}"
                  f"  (lr = 4e-4: a +-lr random walk of {len(epochs) * 10} steps has mean |.| = {4e-4 * np.sqrt(len(epochs) * 10 * 2 / np.pi):.3e})")


if __name__ == "__main__":
    main()
