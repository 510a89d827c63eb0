"""The weight-gradient kernel's staging map on the host (no GPU): tests/emul/wgrad_map_emul.cpp compiles
consistent_depth_amd/csrc/wgrad_stage_map.h -- the same header wgrad_split.hip compiles for gfx950 -- with g++ and checks, for every
block shape (k = 3 with one and two output-channel groups, 5, 7, 11), that the (thread, slot) -> tile element -> LDS word map
writes every pixel pair of the dY and X tiles exactly once at the address the MFMA fragments read, and nothing else."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def test_staging_map_covers_both_tiles_exactly_once(tmp_path):
    exe = tmp_path / "wgrad_map_emul"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(REPO, "consistent_depth_amd", "csrc"), "-o", str(exe),
                           os.path.join(HERE, "emul", "wgrad_map_emul.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("KS=")]
    assert len(lines) == 5 and all(ln.endswith("bad=0") for ln in lines), r.stdout
