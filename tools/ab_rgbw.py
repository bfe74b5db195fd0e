"""Same-session A/B of the rgb head's backward: streamed layer-0 / 1 weight gradients (round 3) vs emer_rgb_head_bwd_fused with one
row tile per step vs two paired tiles per step.  Kernel-level timing (HIP events over the whole backward of the head at the metric
shape) and the full bench step in each mode.  Usage (GPU box): python tools/ab_rgbw.py [--bench]"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def head_backward_us(mode: str, R=8192, S=128, Kh=49, reps=12):
    from emernerf_amd import fused
    fused.FUSED_RGB_WGRAD = mode != "streamed"
    fused.RGB_WGRAD_PAIR = mode == "paired"
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    rnd = lambda *sh, s=1.0: (torch.randn(*sh, generator=g) * s).to(dev).requires_grad_(True)
    N = R * S
    hray, geo = rnd(R, Kh), rnd(N, 64)
    wc = [rnd(64, Kh + 64, s=0.1), rnd(64, s=0.1), rnd(64, 64 + Kh + 64, s=0.1), rnd(64, s=0.1), rnd(3, 64, s=0.1), rnd(3, s=0.1)]
    gw = torch.randn(N, 3, generator=g).to(dev)
    ts = []
    for i in range(reps + 3):
        for t in wc + [hray, geo]:
            t.grad = None
        out = fused.rgb_head(hray, geo, S, *wc)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out.backward(gw)
        b.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return {"median_us": ts[len(ts) // 2], "min_us": ts[0], "dW1_checksum": float(wc[2].grad.double().abs().sum()), "dW0_checksum": float(wc[0].grad.double().abs().sum())}


if __name__ == "__main__":
    if "--lib" in sys.argv:   # time one mode on another build of the library (tools/build_variant.sh <tag> ...): fresh process per build
        i = sys.argv.index("--lib")
        tag, mode = sys.argv[i + 1], (sys.argv[i + 2] if len(sys.argv) > i + 2 else "tile")
        import emernerf_amd._lib as L
        import emernerf_amd._build as B
        if tag != "base":
            L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libemernerf_{tag}.so")
            B.build = lambda *a, **k: L.LIB_PATH
        r = [head_backward_us(mode) for _ in range(2)]
        print(json.dumps({"lib": tag, "mode": mode, "median_us": [round(x["median_us"], 1) for x in r], "min_us": [round(x["min_us"], 1) for x in r]}))
        sys.exit(0)
    res = {"head_backward": {m: head_backward_us(m) for m in ("streamed", "tile", "paired", "streamed", "tile", "paired")[:3]}}
    res["head_backward_again"] = {m: head_backward_us(m) for m in ("paired", "tile", "streamed")}
    if "--bench" in sys.argv:
        res["bench"] = {}
        for mode, env in (("streamed", {"EMER_FUSE_RGB_WGRAD": "0"}), ("tile", {}), ("paired", {"EMER_RGBW_PAIR": "1"}), ("streamed_again", {"EMER_FUSE_RGB_WGRAD": "0"})):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--no-second-state", "--no-secondary",
                                "--no-fp16-state", "--steps", "60", "--warmup", "10"], capture_output=True, text=True, env={**os.environ, **env}, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
            if line:
                j = json.loads(line[0])
                k = j["kernels"]
                res["bench"][mode] = {"ms_per_step": j["ms_per_step"], "eager_ms": (j["config"]["other_launch_mode"] or {}).get("ms_per_step"),
                                      "kernels_ms": {n: round(v["ms_per_step"], 4) for n, v in k.items() if n in ("emer_rgb_head_bwd", "emer_rgb_head_bwd_fused", "emer_wgrad_segmented", "emer_field_fwd", "emer_neck_bwd_fused")}}
            else:
                res["bench"][mode] = {"error": r.stderr[-500:]}
    print(json.dumps(res, indent=1))
