import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_loss_cases():
    return sorted(os.path.basename(p)[len("loss_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "loss_*.npz")))


def load_loss_case(name):
    z = np.load(os.path.join(GOLDEN, f"loss_{name}.npz"))
    batch = {
        "depth": z["depth"],
        "flows": [z["flow_fwd"], z["flow_bwd"]],
        "masks": [z["mask_fwd"].astype(np.float32), z["mask_bwd"].astype(np.float32)],
        "intrinsics": z["intrinsics"],
        "extrinsics": z["extrinsics"],
    }
    lam = z["lambdas"]
    ref64 = {k[len("ref64_"):]: z[k] for k in z.files if k.startswith("ref64_")}
    ref32 = {k[len("ref32_"):]: z[k] for k in z.files if k.startswith("ref32_")}
    return batch, float(lam[0]), float(lam[1]), ref64, ref32


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
