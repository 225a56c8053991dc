"""GPU box: bench.py with other input statistics (timing only): tools/scratch/bench_data.py <zeros|const|uniform|default> [bench.py args]
How much of the pass's energy follows the operand VALUES: all-zero images make every tensor constant per channel."""
import os, runpy, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import roomnet_amd.synth as synth
mode = sys.argv[1]
orig = synth.perf_batch
def patched(n, side=224, seed=0):
    if mode == "zeros":
        return np.zeros((n, side, side, 3), np.uint8)
    if mode == "const":
        return np.full((n, side, side, 3), 137, np.uint8)
    if mode == "uniform":
        return np.random.default_rng(seed).integers(0, 256, (n, side, side, 3), dtype=np.uint8)
    return orig(n, side, seed)
synth.perf_batch = patched
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
