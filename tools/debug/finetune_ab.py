"""Same-box A/B of config 3's step (bench.finetune_samples) under environment / module switches given as NAME=a,b ..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ultra_torchdrug_amd.data import DEFAULT_SEED

dev = torch.device("cuda:0")
wl = os.environ.get("WORKLOAD", "S-wn18rr")
B = int(os.environ.get("BATCH", "16"))
variants = sys.argv[1:] or ["base"]
task, triples, _, n_fact = bench.transductive_task(wl, dev, 512, DEFAULT_SEED)
bench.prepare_plans(task)
facts = torch.from_numpy(triples[:n_fact]).to(dev)
state = {k: v.clone() for k, v in task.state_dict().items()}
for rnd in range(2):
    for v in variants:
        for kv in v.split(";"):
            if "=" in kv:
                k, val = kv.split("=")
                os.environ[k] = val
                for mod in list(sys.modules.values()):
                    name = k.replace("ULTRA_", "")
                    if getattr(mod, "__name__", "").startswith("ultra_torchdrug_amd") and hasattr(mod, name) and isinstance(getattr(mod, name), bool):
                        setattr(mod, name, val != "0")
        task.load_state_dict(state)
        samples, _ = bench.finetune_samples(task, facts, B, 30, DEFAULT_SEED)
        print("%-60s median %.3f ms  min %.3f" % (v, statistics.median(samples), min(samples)), flush=True)
