"""A/B of the launch shape of the fp32 1x1 kernels on the configs[4] step: cd_debug_force_conv_co_tiles(n) before bench.py's main."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd import _native
cot, ty = int(sys.argv[1]), int(sys.argv[2])
lib = _native.lib()
if cot: assert lib.cd_debug_force_conv_co_tiles(cot) == 0
if ty: assert lib.cd_debug_force_conv_tile_rows(ty) == 0
sys.argv = ["bench.py", "--model", "midas2", "--height", "384", "--width", "384", "--batch-size", "8", "--frames", "20", "--steps", "5", "--warmup", "2",
            "--no-cpu-baseline", "--no-loss-microbench"]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), run_name="__main__")
