"""A/B two builds of the kernel library inside ONE gpurun session (box-to-box variance is ~5 %, far more than most
kernel changes).  Usage on the GPU box:
    python tools/ab_bench.py base                  # emernerf_amd/lib/libemernerf_hip.so
    python tools/ab_bench.py <tag> [bench args]    # emernerf_amd/lib/libemernerf_<tag>.so (built by hand from a variant source)
Prints bench.py's JSON line; alternate the two a few times and compare ms_per_step / roofline.avg_us."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import emernerf_amd._build as B
import emernerf_amd._lib as L
tag = sys.argv[1] if len(sys.argv) > 1 else "base"
if tag != "base":
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libemernerf_{tag}.so")
    B.build = lambda *a, **k: L.LIB_PATH
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
