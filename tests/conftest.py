import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity, cgroup CPU quota).  The GPU box shows 256 cores and grants 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def reference_threads() -> int:
    """Threads for the CPU references: torch's default where the visible cores are all usable (the build container: the thread count the
    fp32 goldens were produced with); under a CPU quota (the GPU box: 256 visible / 16 granted) the quota minus two -- a team that
    burns the whole quota gets the entire cgroup throttled, the box's own agent included."""
    u = usable_cores()
    return u if (os.cpu_count() or 1) <= u else max(1, u - 2)


if (os.cpu_count() or 1) > usable_cores():
    # OpenMP / MKL teams sized by the visible cores crawl under the CPU quota (round 4: a 2-pair 64x48 fp64 reference took 213 s).
    # Set before torch is imported, inherited by the subprocesses the multi-rank tests start.
    for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(_v, str(reference_threads()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU references (fp64 hourglass, oracle loops) with as many threads as the cgroup grants: torch's default is the number of
    # cores it SEES (256 on the GPU box, of which the container may use 16) -- oversubscribed OpenMP / MKL teams crawl
    try:
        import torch
        if torch.get_num_threads() > reference_threads():  # only ever DOWN: fewer threads than torch's default would change the
            torch.set_num_threads(reference_threads())     # summation order of the fp32 goldens produced with the default
    except ImportError:
        pass


# ---- the GPU suite's time budget ---------------------------------------------------------------------------------------------
# The driver gives `pytest -m gpu` 1200 s on the GPU box (round 3: 704 s; a round-4 run with oversubscribed CPU references: 1207 s,
# profiles/gpu_suite_r04.txt).  Nearly all of that is the LIVE CPU references of three tests (fp64 / fp32 restatements of the
# reference's training step computed on the box's host cores); everything they assert at the headline shape is also asserted against
# committed offline goldens (tests/golden/loop_16f_384x224.npz, engine_ref_8x384x224.npz).  So: those three run LAST, cheapest first,
# and one of them is SKIPPED -- loudly, with the numbers -- if the time left cannot hold its estimated cost; a killed suite would
# report nothing at all.  CD_AMD_TEST_BUDGET_S moves the budget (0 = no guard).
_HEAVY = {      # nodeid suffix -> estimated seconds on the GPU box (upper estimates from profiles/gpu_suite_r04.txt)
    "test_finetune_gpu.py::test_short_finetune_matches_cpu_reference[hip]": 220,
    "test_loop_gpu.py::test_epochs_after_burn_in_within_1e_3": 180,
    "test_finetune_gpu.py::test_run_level_parity_after_burn_in": 320,
}
_T0 = [None]


def _heavy_cost(nodeid):
    for k, v in _HEAVY.items():
        if nodeid.endswith(k):
            return v
    return 0


# ... and the one test whose convolutions are a THIRD PARTY's (BASELINE configs[1]: PyTorch-ROCm / MIOpen) runs at the very end: under
# `pytest -x` its failure must not hide any test of this package (it is also the longest test: three continuations, ~4 minutes).
_THIRD_PARTY_LAST = ("test_loop_gpu.py::test_config1_torch_convs_hip_loss_from_the_same_snapshot",)


def order_heavy_last(items):
    """The items with the live-CPU-reference tests moved to the end, cheapest first (stable otherwise); the third-party test last of all."""
    last = [it for it in items if it.nodeid.endswith(_THIRD_PARTY_LAST)]
    rest = [it for it in items if not it.nodeid.endswith(_THIRD_PARTY_LAST)]
    light = [it for it in rest if not _heavy_cost(it.nodeid)]
    heavy = sorted((it for it in rest if _heavy_cost(it.nodeid)), key=lambda it: _heavy_cost(it.nodeid))
    return light + heavy + last


def budget_verdict(nodeid, elapsed, budget):
    """None to run the test; otherwise the reason for skipping it."""
    cost = _heavy_cost(nodeid)
    if not cost or budget <= 0 or elapsed + cost <= budget:
        return None
    return (f"time budget: {elapsed:.0f} s of {budget:.0f} s used, this test's live CPU reference needs ~{cost} s "
            f"(its assertions at the headline shape are covered by the committed goldens; CD_AMD_TEST_BUDGET_S=0 runs it regardless)")


def pytest_collection_modifyitems(config, items):
    items[:] = order_heavy_last(items)


def _process_age() -> float:
    """Seconds since this interpreter was started (the driver's clock includes the first `import torch` of a fresh box: 1-2 minutes)."""
    import time
    try:
        import psutil
        return time.time() - psutil.Process().create_time()
    except Exception:   # noqa: BLE001
        return time.monotonic() - (_T0[0] if _T0[0] is not None else time.monotonic())


def budget_left() -> float:
    """Seconds until the suite's time budget ends (inf with CD_AMD_TEST_BUDGET_S=0): what a test with a long live CPU reference checks
    between its steps (tests/test_finetune_gpu.py::test_run_level_parity_after_burn_in)."""
    budget = float(os.environ.get("CD_AMD_TEST_BUDGET_S", "1050"))
    return float("inf") if budget <= 0 else budget - _process_age()


def pytest_sessionstart(session):
    import time
    _T0[0] = time.monotonic()


BUDGET_SKIPPED = []     # nodeids of the heavy tests skipped for the time budget in this session (also appended to by tests that stop mid-way)


def pytest_runtest_setup(item):
    if _T0[0] is None or not _heavy_cost(item.nodeid):
        return
    budget = float(os.environ.get("CD_AMD_TEST_BUDGET_S", "1050"))
    why = budget_verdict(item.nodeid, _process_age(), budget)
    if why:
        BUDGET_SKIPPED.append(item.nodeid)
        pytest.skip("BUDGET-SKIP " + why)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """A green suite must say whether its heaviest parity tests ran: one line with a COUNT (0 in a normal run), the names, and -- with
    CD_AMD_TEST_STRICT_BUDGET=1 -- a failing exit status when the count is not 0.  The same line goes to gpurun_out/budget_skips.txt
    when that directory exists (the GPU box), so that it is merged back with the run's artefacts."""
    heavy = [k for k in _HEAVY]
    line = (f"BUDGET-SKIPPED heavy live-reference tests: {len(BUDGET_SKIPPED)} of {len(heavy)}"
            + (f" ({', '.join(n.split('::')[-1] for n in BUDGET_SKIPPED)})" if BUDGET_SKIPPED else ""))
    terminalreporter.write_line(line)
    out = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(out):
        try:
            with open(os.path.join(out, "budget_skips.txt"), "w") as f:
                f.write(line + "\n")
        except OSError:
            pass
    if BUDGET_SKIPPED and os.environ.get("CD_AMD_TEST_STRICT_BUDGET") == "1":
        terminalreporter.write_line("CD_AMD_TEST_STRICT_BUDGET=1: treating budget skips as a failure")
        config._cd_budget_failure = True


def pytest_sessionfinish(session, exitstatus):
    if getattr(session.config, "_cd_budget_failure", False) and session.exitstatus == 0:
        session.exitstatus = 1


_BG = {}


def pytest_collection_finish(session):
    """Start the slow fp64 CPU reference of the BASELINE-shape engine test in the background (tests/bg_reference.py) as soon
    as it is known that the test will run, so that it overlaps with the other GPU tests."""
    import subprocess
    import tempfile
    if os.environ.get("CD_AMD_TEST_NO_BG") or not any("baseline_8x384x224" in it.nodeid for it in session.items):
        return
    if not os.environ.get("CD_AMD_TEST_LIVE_ENGINE_REF"):
        import bg_reference
        if os.path.exists(bg_reference.GOLDEN):       # the committed golden replaces the live computation (tests/bg_reference.py)
            return
    if len(session.items) < 20:       # run on its own: nothing to overlap with, the test computes inline
        return
    dst = os.path.join(tempfile.mkdtemp(prefix="cd_bg_"), "engine_ref_8x384x224.npz")
    threads = max(2, usable_cores() // 2)
    proc = subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "bg_reference.py"), dst, str(threads)],
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
    _BG["engine_ref"] = (proc, dst)


def pytest_sessionfinish(session, exitstatus):
    for proc, _ in _BG.values():
        if proc.poll() is None:
            proc.kill()


def background_engine_reference(timeout=1500):
    """The fp64 reference of the BASELINE-shape engine test as a dict of numpy arrays: the committed golden (sampled: key "sampled"),
    or -- CD_AMD_TEST_LIVE_ENGINE_REF=1, or no golden -- the result of the background process; None if neither exists (the test then
    computes inline); "skip" if the host lacks the memory for the live computation."""
    if not os.environ.get("CD_AMD_TEST_LIVE_ENGINE_REF"):
        import bg_reference
        g = bg_reference.load_golden()
        if g is not None:
            return g
    if "engine_ref" not in _BG:
        return None
    proc, dst = _BG["engine_ref"]
    try:
        rc = proc.wait(timeout=timeout)
    except Exception:   # noqa: BLE001
        proc.kill()
        return None
    if rc == 3:
        return "skip"
    if rc != 0 or not os.path.exists(dst):
        return None
    with np.load(dst) as z:
        return {k: z[k] for k in z.files}


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
