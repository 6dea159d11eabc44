"""Which side of the captured bucket traffic goes stale?  After ONE replay of the in-graph step: the true gradients (an
eager backward with the negatives the replay drew), the bucket buffers and the parameters' .grad, bucket by bucket."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import copy
import torch
import torch.distributed as dist
from ultra_torchdrug_amd import engine
from ultra_torchdrug_amd.data import synthetic_triples
from ultra_torchdrug_amd.graph import Graph
from ultra_torchdrug_amd.task import build_ultra

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29546", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")


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


mode = sys.argv[1] if len(sys.argv) > 1 else "rccl"
task, triples = build()
state = copy.deepcopy(task.state_dict())
twin, _ = build()
twin.load_state_dict(state)
batch = torch.from_numpy(triples[:8]).to(dev)
opt_g = torch.optim.SGD(twin.parameters(), lr=0.0)             # the step leaves the weights alone: grads stay comparable
reducer = engine.GradientReducer(twin, overlap=True, single_rank=True)
log = []
real_launch = reducer._launch
def logged(b, defer_collective=False):
    log.append((b, bool(defer_collective), bool(torch.cuda.is_current_stream_capturing()), int(torch.cuda.current_stream().cuda_stream)))
    return real_launch(b, defer_collective)
reducer._launch = logged
if mode == "stand_in":
    def fake(t, async_op=False):
        t.mul_(1.0)
        return StreamOnlyWork(torch.cuda.current_stream())
    engine.dist.all_reduce = fake
step = engine.GraphedTrainStep(twin, opt_g, batch, reducer=reducer, reduce_in_graph=True)
print(mode, "in graph:", step.reduce_in_graph, "| launches seen (bucket, deferred, capturing, stream):", log[-13:])
for it in range(3):
    for b in reducer.buckets:
        b["flat"].fill_(777.0)                                  # anything a replay does not rewrite stays 777
    step(batch)
    torch.cuda.synchronize()
    task.load_state_dict(state)
    task.zero_grad(set_to_none=True)
    task._static_negative = step.last_negatives.clone()
    loss, _ = task(batch)
    loss.backward()
    task._static_negative = None
    true = dict((k, p.grad) for k, p in task.named_parameters() if p.grad is not None)
    line = []
    for b in reducer.buckets:
        want = torch.cat([true[n].reshape(-1) for n in b["names"]])
        got_flat = b["flat"]
        got_grad = torch.cat([dict(twin.named_parameters())[n].grad.reshape(-1) for n in b["names"]])
        line.append("%s flat=%s grad=%s%s" % (b["name"].split(".")[-1] + ("e" if b["name"].startswith("model.") else "r"),
                                              "ok" if torch.equal(got_flat, want) else ("777" if bool((got_flat == 777).all()) else "BAD"),
                                              "ok" if torch.equal(got_grad, want) else ("777" if bool((got_grad == 777).all()) else "BAD"), ""))
    print("replay", it, " | ".join(line))
dist.destroy_process_group()
