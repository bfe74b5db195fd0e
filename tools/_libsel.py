"""`--lib <tag>` for the probe tools: load emernerf_amd/lib/libemernerf_<tag>.so instead of the in-tree build (same-session
A/B of kernel variants, e.g. the round-2 library kept as libemernerf_r02.so).  Import before emernerf_amd.fused / ops."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emernerf_amd._lib as _L0  # noqa: E402

TAG = "base"
if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    TAG = sys.argv[i + 1]
    del sys.argv[i:i + 2]
    if TAG != "base":
        _L0.LIB_PATH = os.path.join(os.path.dirname(_L0.LIB_PATH), f"libemernerf_{TAG}.so")
        import emernerf_amd._build as _B
        _B.build = lambda *a, **k: _L0.LIB_PATH
        if os.environ.get("EMER_LIBSEL_SAME_ABI") != "1":   # (a variant of the CURRENT sources keeps every entry point)
            _L0.ALLOW_MISSING_SYMBOLS = True   # an older build: entry points added since are absent ...
            import emernerf_amd.fused as _F
            _F.FUSED_WGRAD = False             # ... so the paths that need them are switched off
