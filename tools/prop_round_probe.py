"""Time one proposal round outside autograd: fused (emer_prop_density_fwd) vs ray_points -> hashgrid_fwd -> density MLP."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from emernerf_amd import ops, radiance_field as rf
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[iters // 2]
R = 8192
for S, mx in ((128, 512), (64, 2048)):
    net = rf.build_density_field(n_levels=8, base_resolution=16, max_resolution=mx, log2_hashmap_size=20, n_features_per_level=1).to(dev)
    with torch.no_grad():
        net.xyz_encoder.tcnn_encoding.params.uniform_(-0.5, 0.5)
    o = (torch.rand(R, 3, device=dev) * 0.4 - 0.2); d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    ts = torch.sort(torch.rand(R, S + 1, device=dev) * 40, dim=-1).values
    t0, t1 = ts[:, :-1].contiguous(), ts[:, 1:].contiguous()
    with torch.no_grad():
        f = timeit(lambda: net.density_from_rays(o, d, t0, t1))
        u = timeit(lambda: net.density_from_normed(ops.ray_points(o, d, t0, t1, net.aabb, net.unbounded)[0]))
    print(json.dumps({"samples": R * S, "max_res": mx, "fused_us": round(f, 1), "separate_us": round(u, 1)}))
