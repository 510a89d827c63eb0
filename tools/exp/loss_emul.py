#!/usr/bin/env python3
"""Numpy fp32 emulation of the arithmetic of the fused HIP loss kernels (csrc/loss_common.h, loss_slab.hip).

Development aid, CPU only: every array op below rounds to fp32 exactly where the kernel does (fma emulated through
fp64), so a change of formulation can be priced -- distance to the fp64 truth, next to the reference's own
fp32-vs-fp64 distance -- before a GPU minute is spent.  The hardware approximations (v_rcp_f32, v_rsq_f32, v_exp_f32:
1 ulp) are modelled as correctly rounded, so real distances are a little higher than printed.

    python tools/exp/loss_emul.py            # goldens + a 4 x 384 x 224 scene batch
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

F = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)


def rcp(a):
    return (F(1) / a).astype(F)


def div_by_const(u, c):
    """u / c correctly rounded from q0 = u*rc, r = fma(-q0, c, u), q = fma(r, rc, q0) (rc = fl(1/c))."""
    rc = F(1) / F(c)
    q0 = (u * rc).astype(F)
    r = fma(-q0, F(c), u)
    return fma(r, rc, q0)


def tap_axis(m, n, coords):
    """Sample position along one axis (size n) of matched coordinate m -> (i0, i1, t)."""
    if coords == "fma":          # round-1 kernel: one fma
        i = fma(m, F(n) / F(n - 1), F(-0.5))
    elif coords == "ref":        # reference op order: ((2m/(n-1) - 1 + 1) * n - 1) / 2, five roundings
        q = div_by_const((F(2) * m).astype(F), n - 1)
        g = (q - F(1)).astype(F)
        g1 = (g + F(1)).astype(F)
        i = (((g1 * F(n)).astype(F) - F(1)).astype(F) * F(0.5)).astype(F)
    else:
        raise ValueError(coords)
    i = np.minimum(np.maximum(i, F(0)), F(n - 1))
    f0 = np.floor(i)
    t = (i - f0).astype(F)
    i0 = f0.astype(np.int64)
    return i0, np.minimum(i0 + 1, n - 1), t


def emulate(batch, lam_r, lam_b, coords="fma", fold=True, mode=0, contract=True):
    """Returns dict(total, reprojection, disparity, grad_depth) computed the way the kernel computes them.
    mode: 0 depth given, 1 depth = exp(v) (gradient w.r.t. v).  contract: a*b+c pairs the compiler fuses."""
    v = batch["depth"].astype(F)
    B, _, H, W = v.shape
    depth = np.exp(v.astype(np.float64)).astype(F) if mode == 1 else v
    intr, extr = batch["intrinsics"].astype(F), batch["extrinsics"].astype(F)
    grad = np.zeros((B, 2, H, W), np.float64)
    rk, dk = np.zeros((B, 2)), np.zeros((B, 2))
    xs = np.arange(W, dtype=F)[None, :].repeat(H, 0)
    ys = np.arange(H, dtype=F)[:, None].repeat(W, 1)
    mad = fma if contract else (lambda a, b, c: ((a * b).astype(F) + c).astype(F))
    for k in range(2):
        fbar = F((intr[:, k, 0] + intr[:, k, 1]).astype(F).sum(dtype=F) / F(2 * B))
        for b in range(B):
            ir, it = intr[b, k], intr[b, 1 - k]
            er, et = extr[b, k], extr[b, 1 - k]
            M = np.zeros((3, 3), F)
            c = np.zeros(3, F)
            for j in range(3):
                for l in range(3):
                    M[j, l] = F(F(et[0, j] * er[0, l]) + F(et[1, j] * er[1, l])) + F(et[2, j] * er[2, l])
                c[j] = F(F(et[0, j] * F(er[0, 3] - et[0, 3])) + F(et[1, j] * F(er[1, 3] - et[1, 3]))) + F(et[2, j] * F(er[2, 3] - et[2, 3]))
            fl = batch["flows"][k][b].astype(F)
            m = batch["masks"][k][b, 0].astype(F)
            S = max(F(m.sum(dtype=np.float64)), F(1e-6))
            gr = F(lam_r) / F(F(2) * F(B) * S) if lam_r > 0 else F(0)
            gb = F(F(lam_b) * fbar) / F(F(2) * F(B) * S) if lam_b > 0 else F(0)
            d = depth[b, k]
            dk_ = depth[b, 1 - k]
            ifx, ify = F(1) / ir[0], F(1) / ir[1]
            r0 = ((xs - ir[2]) * ifx).astype(F)
            r1 = (-(ys - ir[3]) * ify).astype(F)
            a = [(mad(M[i, 0], r0, mad(M[i, 1], r1, -M[i, 2]))) for i in range(3)]
            X, Y, Z = mad(d, a[0], c[0]), mad(d, a[1], c[1]), mad(d, a[2], c[2])
            iZ = rcp(Z)
            mx, my = (xs + fl[0]).astype(F), (ys + fl[1]).astype(F)
            g = np.zeros((H, W), F)
            if lam_r > 0:
                XiZ, YiZ = (X * iZ).astype(F), (Y * iZ).astype(F)
                ex = (mad(-it[0], XiZ, it[2]) - mx).astype(F)
                ey = (mad(it[1], YiZ, it[3]) - my).astype(F)
                e2 = mad(ex, ex, (ey * ey).astype(F))
                with np.errstate(divide="ignore", invalid="ignore"):
                    ie = np.where(e2 > 0, (F(1) / np.sqrt(e2)).astype(F), F(0)).astype(F)
                rk[b, k] = (m.astype(np.float64) * (e2 * ie).astype(F)).sum() / S
                dpx = ((it[0] * iZ).astype(F) * mad(XiZ, a[2], -a[0])).astype(F)
                dpy = ((it[1] * iZ).astype(F) * mad(-YiZ, a[2], a[1])).astype(F)
                g = (g + (((gr * m).astype(F) * mad(ex, dpx, (ey * dpy).astype(F))).astype(F) * ie).astype(F)).astype(F)
            if lam_b > 0:
                xa, xb, tx = tap_axis(mx, W, coords)
                ya, yb, ty = tap_axis(my, H, coords)
                w00 = ((F(1) - tx) * (F(1) - ty)).astype(F); w01 = (tx * (F(1) - ty)).astype(F)
                w10 = ((F(1) - tx) * ty).astype(F); w11 = (tx * ty).astype(F)
                d00, d01, d10, d11 = dk_[ya, xa], dk_[ya, xb], dk_[yb, xa], dk_[yb, xb]
                zs = -mad(d11, w11, mad(d10, w10, mad(d01, w01, (d00 * w00).astype(F))))
                izs = rcp(zs)
                dd = (iZ - izs).astype(F)
                dk[b, k] = fbar * ((m.astype(np.float64) * np.abs(dd)).sum() / S)
                sg = np.sign(dd).astype(F)
                gm = ((gb * m).astype(F) * sg).astype(F)
                g = (g - ((gm * a[2]).astype(F) * iZ).astype(F) * iZ).astype(F)
                gz = ((gm * izs).astype(F) * izs).astype(F)
                for (yy, xx, ww, dt) in ((ya, xa, w00, d00), (ya, xb, w01, d01), (yb, xa, w10, d10), (yb, xb, w11, d11)):
                    cc = (-(gz * ww).astype(F)).astype(F)
                    if mode == 1:
                        cc = (cc * dt).astype(F)
                    np.add.at(grad[b, 1 - k], (yy.ravel(), xx.ravel()), np.where(m.ravel() != 0, cc.ravel(), 0).astype(np.float64))
            grad[b, k] += (g * d).astype(F) if mode == 1 else g
    reproj = lam_r * rk.mean(1) if lam_r > 0 else np.zeros(B)
    disp = lam_b * dk.mean(1) if lam_b > 0 else np.zeros(B)
    return {"total": np.array([(reproj + disp).mean()]), "reprojection": reproj, "disparity": disp,
            "grad_depth": grad.astype(F)}


def main():
    from conftest import golden_loss_cases, load_loss_case
    from oracle import oracle as o
    from consistent_depth_amd import synthetic
    o.build()
    cases = [(n,) + load_loss_case(n)[:4] for n in golden_loss_cases()]
    for gen, nm in ((synthetic.make_scene_batch, "scene4x384x224"), (synthetic.make_pair_batch, "unrelated4x384x224")):
        bt = gen(4, 384, 224, seed=11)
        r64 = o.consistency_loss(bt["depth"], bt["flows"], bt["masks"], bt["intrinsics"], bt["extrinsics"], 1.0, 0.1)
        cases.append((nm, bt, 1.0, 0.1, r64))
    for name, batch, lr, lb, r64 in cases:
        o32 = o.consistency_loss(batch["depth"], batch["flows"], batch["masks"], batch["intrinsics"], batch["extrinsics"], lr, lb, dtype=np.float32)
        line = f"{name:22s} ref-order fp32: grad {o.rel_l1(o32['grad_depth'], r64['grad_depth']):.2e} loss {abs(o32['total'][0] - r64['total'][0]) / max(abs(r64['total'][0]), 1e-30):.1e}"
        for coords in ("fma", "ref"):
            e = emulate(batch, lr, lb, coords=coords)
            line += f" | {coords}: grad {o.rel_l1(e['grad_depth'], r64['grad_depth']):.2e} loss {abs(e['total'][0] - r64['total'][0]) / max(abs(r64['total'][0]), 1e-30):.1e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
