"""TEST INFRASTRUCTURE ONLY -- seeded input generator shared by oracle/gen_golden_masks.py (runs the reference) and the
GPU parity tests (which must not touch /root/reference)."""
import numpy as np


def make_case(H, W, seed, wild=False):
    """A pair with a smooth forward flow, its (approximately) inverse backward flow plus noise that breaks the
    consistency in places, colours that follow the flow plus noise; `wild` adds flows that leave the image."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    amp = 6.0 if not wild else 0.6 * max(H, W)
    u = amp * np.sin(2 * np.pi * yy / H + 0.3) * np.cos(np.pi * xx / W)
    v = 0.5 * amp * np.cos(2 * np.pi * xx / W + 0.7)
    fwd = np.stack((u, v), -1)
    # backward flow at the forward-warped position ~ -fwd; evaluate the analytic field at (x+u, y+v)
    xs, ys = xx + u, yy + v
    u_b = -amp * np.sin(2 * np.pi * ys / H + 0.3) * np.cos(np.pi * xs / W)
    v_b = -0.5 * amp * np.cos(2 * np.pi * xs / W + 0.7)
    # store it on the TARGET grid approximately (nearest): good enough -- most pixels consistent within ~1 px, some not
    bwd = np.stack((-u, -v), -1) + rng.normal(0, 0.45, (H, W, 2))
    del u_b, v_b
    base = rng.random((H, W, 3))
    k = np.ones(5) / 5
    for ax in (0, 1):
        base = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), ax, base)
    c0 = base + rng.normal(0, 0.02, base.shape)
    c1 = np.roll(base, (2, -3), (0, 1)) + rng.normal(0, 0.3, base.shape) * (rng.random((H, W, 1)) < 0.3)
    return [fwd.astype(np.float32), bwd.astype(np.float32)], [c0.astype(np.float32), c1.astype(np.float32)]


