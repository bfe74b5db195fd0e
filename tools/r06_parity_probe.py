"""[r6] Statistics behind the bounds of tests/test_train_parity_gpu.py: K-step co-training from several seeds, printing the loss / PSNR
distances, the count of sign-flipped table entries and the parameter distance with and without them."""
import json
import sys

import torch

sys.path.insert(0, ".")
from oracle.train_parity import cotrain  # noqa: E402

cases = [("static", 30, None, dict(rays=512, samples=64, prop_samples=(64, 32)), s) for s in (11, 12, 13, 14, 15, 16)]
cases += [("static", 30, 0.3, dict(rays=512, samples=64, prop_samples=(64, 32)), 11)]
cases += [(k, 8, 0.3, dict(rays=256, samples=32, prop_samples=(32, 16)), 11) for k in ("flow", "feature")]
for kind, K, init, shape, seed in cases:
    r = cotrain(kind, torch.device("cuda:0"), K=K, num_iters=200, table_init=init, seed=seed, **shape)
    mlp_max = max(st["max_abs_diff"] for n, st in r["param_stats"].items() if not n.endswith("tcnn_encoding.params"))
    print(json.dumps({"kind": kind, "K": K, "init": init, "seed": seed, "loss_rel": r["loss_max_rel_diff"],
                      "dpsnr": r["hip_psnr_vs_gt_db"] - r["ref_psnr_vs_gt_db"], "travel": r["travel"], "l2": r["param_l2_diff"],
                      "l2_excl": r["param_l2_diff_excl"], "flipped": r["n_sign_flipped"], "entries": r["n_table_entries"],
                      "l2_over_travel": r["param_l2_diff"] / r["travel"], "l2_excl_over_travel": r["param_l2_diff_excl"] / r["travel"],
                      "mlp_max_abs": mlp_max, "losses": [r["hip_losses"][0], r["hip_losses"][-1]]}), flush=True)
