"""A/B two builds of the kernel library inside ONE gpurun session (box-to-box variance is ~5 %, far more than most
kernel changes).  Usage on the GPU box:
    python tools/ab_bench.py base                  # emernerf_amd/lib/libemernerf_hip.so
    python tools/ab_bench.py <tag> [bench args]    # emernerf_amd/lib/libemernerf_<tag>.so (built by hand from a variant source)
Prints bench.py's JSON line; alternate the two a few times and compare ms_per_step / roofline.avg_us."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "base"
sys.argv = [sys.argv[0], "--lib", tag] + sys.argv[2:]
from tools import _libsel  # noqa: E402,F401  (selects the library, tolerates entry points an older build lacks)
if os.environ.get("EMER_NO_FUSED_WGRAD"):
    import emernerf_amd.fused as _F
    _F.FUSED_WGRAD = False
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
