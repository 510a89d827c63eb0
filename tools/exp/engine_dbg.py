import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import pytest
from consistent_depth_amd import _native
lib = _native.lib()
os.environ["CD_AMD_CONV_AUTOTUNE"] = "0"
for name, streams, ty in [("streams_none_heur", "none", 0), ("level_ty4", "level", 4), ("level_heur", "level", 0), ("branch_heur", "branch", 0)]:
    os.environ["CD_AMD_ENGINE_STREAMS"] = streams
    lib.cd_debug_force_conv_tile_rows(ty)
    rc = pytest.main(["-x", "-q", "-m", "gpu", os.path.join(root, "tests/test_hourglass_engine_gpu.py"), "-k", "2-64-96", "--tb=line", "-p", "no:cacheprovider"])
    print("RESULT", name, rc, flush=True)
