"""Precision probe of the GMA aggregation GEMM alone (ACCFLOW_AGG_PAIRMASK = 7 / 6 / 5 / 4: which of the fp16 split's three
products it runs, every other kernel untouched): EPE of AccFlow(GMA) 7 x 720 x 1280 against the reference's own outputs
(tests/golden/accflow_gma_c5.npz) and time per sequence.   ACCFLOW_AGG_PAIRMASK=6 python tools/agg_precision_probe.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
from accflow_amd.networks import build_flow_estimator
from accflow_amd.networks.AccFlow_ import AccFlow
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "accflow_gma_c5.npz")))
model = AccFlow(build_flow_estimator("acc|gma"))
model.load_state_dict(make_state_dict(model), strict=True)
model = model.cuda().eval()
frames = [normalize(f).cuda() for f in make_sequence(int(g["seed"]) if "seed" in g else 1000, 7, 720, 1280)]
outs = model(images=frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    outs = model(images=frames)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
means, mx = [], 0.0
for k, o in enumerate(outs):
    d = (o[:1, :, ::8, ::8].cpu() - torch.from_numpy(g["out%d" % k])).pow(2).sum(1).sqrt()
    means.append(float(d.mean())); mx = max(mx, float(d.max()))
print("ACCFLOW_AGG_PAIRMASK=%s (bit0 v_lo*attn_hi, bit1 v_hi*attn_lo, bit2 hi*hi): EPE vs reference mean %.2e max %.2e px | %.2f ms per sequence"
      % (os.environ.get("ACCFLOW_AGG_PAIRMASK", "7"), max(means), mx, ms))
