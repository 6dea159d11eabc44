"""Does a hipGraph replay a captured RCCL all-reduce correctly on this runtime?  (A) a tiny producer -> all_reduce on a side
stream -> consumer graph; (B) the full GraphedTrainStep with the collective replaced by a stream-only stand-in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29545", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")

# ---------------- A
x = torch.zeros(1 << 16, device=dev)
flat = torch.empty_like(x)
out = torch.empty_like(x)
side = torch.cuda.Stream()
for _ in range(2):                                   # warm RCCL up outside the capture
    dist.all_reduce(flat)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    torch.mul(x, 2.0, out=flat)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work = dist.all_reduce(flat, async_op=True)
    work.wait()
    torch.add(flat, 1.0, out=out)
bad = 0
for i in range(1, 21):
    x.fill_(float(i))
    g.replay()
    torch.cuda.synchronize()
    bad += int(not torch.equal(out, torch.full_like(out, 2.0 * i + 1.0)))
print("A: tiny graph with a captured all_reduce on a side stream: %d of 20 replays wrong" % bad)

# ---------------- B
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import copy
from ultra_torchdrug_amd import engine
from ultra_torchdrug_amd.data import synthetic_triples
from ultra_torchdrug_amd.graph import Graph
from ultra_torchdrug_amd.task import build_ultra


def build():
    triples, n, r = synthetic_triples("S-tiny", 1024)
    torch.manual_seed(1024)
    task = build_ultra(r)
    task.preprocess(Graph(torch.from_numpy(triples), num_node=n, num_relation=r))
    task.num_negative = 16
    return task.to(dev).train(), triples


class StreamOnlyWork:
    def __init__(self, stream):
        self.event = torch.cuda.Event()
        self.event.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


for mode in ("stand_in", "rccl"):
    task, triples = build()
    state = copy.deepcopy(task.state_dict())
    twin, _ = build()
    twin.load_state_dict(state)
    batches = [torch.from_numpy(triples[i:i + 8]).to(dev) for i in (0, 8, 16)]
    opt_g = torch.optim.AdamW(twin.parameters(), lr=1e-3)
    reducer = engine.GradientReducer(twin, overlap=True, single_rank=True)
    real = dist.all_reduce
    if mode == "stand_in":
        def fake(t, async_op=False):
            t.mul_(1.0)                                   # a kernel on the side stream in the collective's place
            return StreamOnlyWork(torch.cuda.current_stream())
        engine.dist.all_reduce = fake
    try:
        step = engine.GraphedTrainStep(twin, opt_g, batches[0], reducer=reducer, reduce_in_graph=True)
        opt_e = torch.optim.AdamW(task.parameters(), lr=1e-3)
        for i, b in enumerate(batches):
            lg = step(b)[0].item()
            task._static_negative = step.last_negatives.clone()
            le = engine.train_step(task, opt_e, b)[0].item()
            task._static_negative = None
            pbad = [k for (k, a), (_, c) in zip(task.named_parameters(), twin.named_parameters()) if not torch.equal(a, c)]
            print("B[%s] step %d: in graph %s, loss %.6f vs %.6f, params differing %d" % (mode, i, step.reduce_in_graph, lg, le, len(pbad)))
    finally:
        engine.dist.all_reduce = real
        reducer.remove_hooks()
dist.destroy_process_group()
