"""The fusion chain of one AccFlow(RAFT) 7 x 480x1024 sequence alone, 10 times (for rocprofv3 --kernel-trace)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accflow_amd import ops
from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
from accflow_amd.networks import build_flow_estimator
from accflow_amd.networks.AccFlow_ import AccFlow
model = AccFlow(build_flow_estimator("acc|raft"))
model.load_state_dict(make_state_dict(model), strict=True)
model = model.cuda().eval()
frames = [normalize(f).cuda() for f in make_sequence(1000, 7, 480, 1024)]
pairs = model.pair_schedule(len(frames))
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
with torch.no_grad(), ops.guard_scope(flag):
    small = model.estimate_small(frames, pairs)
    by_pair = {p: small[k:k + 1] for k, p in enumerate(pairs)}
    for _ in range(3):
        model.fuse_chain(frames, by_pair)
    torch.cuda.synchronize()
    marker = torch.zeros(1, device="cuda")
    marker.add_(1)                     # (a torch kernel as a marker in the trace)
    for _ in range(10):
        model.fuse_chain(frames, by_pair)
    torch.cuda.synchronize()
