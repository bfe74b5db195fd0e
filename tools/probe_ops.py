"""Which small torch ops (by input shapes) make up the glue of a training step."""
import os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emernerf_amd.trainer import Trainer, synthetic_rays
dev = torch.device("cuda:0")
tr = Trainer(kind="static", device=dev)
tr.set_step(1001)
data = synthetic_rays(8192, dev, seed=1000)
for _ in range(8):
    tr.train_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(6):
        tr.train_step(data)
    torch.cuda.synchronize()
cnt, tim = collections.Counter(), collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.device_time > 0:
        key = (e.name, str(e.input_shapes)[:110])
        cnt[key] += 1; tim[key] += e.device_time
for key, t in sorted(tim.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{cnt[key] / 6:5.1f}/step {t / 6:7.1f} us/step  {key[0]:28s} {key[1]}")
