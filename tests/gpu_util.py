"""Helpers for the -m gpu tests (import torch lazily; the GPU box has no /root/reference)."""
import numpy as np


def to_dev(batch, torch, device="cuda"):
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)  # noqa: E731
    return {
        "depth": t(batch["depth"]),
        "flows": [t(f) for f in batch["flows"]],
        "masks": [t(m) for m in batch["masks"]],
        "intrinsics": t(batch["intrinsics"]),
        "extrinsics": t(batch["extrinsics"]),
    }


def metadata_of(d):
    return {"intrinsics": d["intrinsics"], "extrinsics": d["extrinsics"],
            "geometry_consistency": {"flows": d["flows"], "masks": d["masks"]}}


class Opt:
    def __init__(self, lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=0):
        self.lambda_reprojection = lambda_reprojection
        self.lambda_view_baseline = lambda_view_baseline
        self.lambda_parameter = lambda_parameter


def report(test, **metrics):
    """Print the measured parity distances of a test (pytest -s / -rP shows them) and append them to
    gpurun_out/parity_log.txt so the numbers of a GPU run can be committed under profiles/."""
    import os
    line = f"PARITY {test}: " + "  ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in metrics.items())
    print(line, flush=True)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_log.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
