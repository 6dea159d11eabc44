"""Every ATen op of ONE eager fine-tuning step with its shapes and the nearest frame in this package: which small
launches a captured step still contains beside the extension's own kernels (tools/debug/train_glue.py [workload])."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from ultra_torchdrug_amd import engine
from ultra_torchdrug_amd.data import synthetic_triples, DEFAULT_SEED
from ultra_torchdrug_amd.graph import Graph
from ultra_torchdrug_amd.task import build_ultra

VIEWS = ("view", "reshape", "expand", "transpose", "t.default", "unsqueeze", "squeeze", "select", "slice", "alias", "detach",
         "as_strided", "permute", "unbind", "split", "_unsafe_view", "narrow", "unflatten", "flatten", "is_", "size", "stride",
         "empty", "_local_scalar", "item", "lift_fresh", "chunk", "view_as", "resize_", "set_")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        short = name.replace("aten.", "")
        if not any(short.startswith(v) for v in VIEWS):
            shapes = tuple(tuple(a.shape) if isinstance(a, torch.Tensor) else None for a in args[:3])
            frame = "-"
            for fs in reversed(traceback.extract_stack(limit=40)):
                if "ultra_torchdrug_amd" in fs.filename and "train_glue" not in fs.filename:
                    frame = "%s:%d" % (os.path.basename(fs.filename), fs.lineno)
                    break
            self.rows[(short, shapes, frame)] += 1
        return out


dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "S-wn18rr"
triples, n, r = synthetic_triples(wl, DEFAULT_SEED)
torch.manual_seed(DEFAULT_SEED)
task = build_ultra(r)
task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r))
task.to(dev).train()
opt = torch.optim.AdamW(task.parameters(), lr=5e-4)
data = torch.from_numpy(triples).to(dev)
rng = np.random.default_rng(0)
for i in range(3):
    idx = torch.from_numpy(rng.choice(len(triples), 16, replace=False)).to(dev)
    engine.train_step(task, opt, data[idx])
idx = torch.from_numpy(rng.choice(len(triples), 16, replace=False)).to(dev)
batch = data[idx]
log = Log()
with log:
    engine.train_step(task, opt, batch)
torch.cuda.synchronize()
total = 0
for (name, shapes, frame), count in sorted(log.rows.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print("%3d  %-34s %-28s %s" % (count, name[:34], frame, shapes))
    total += count
print("ops:", total)
