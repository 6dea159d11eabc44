"""bench.py -- edges aggregated / second of the rspmm Bellman-Ford hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched by
``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric: "edges aggregated/sec ... FB15k237 6L x 64d rspmm"): S-fb15k237 -- a seeded
synthetic KG of FB15k237's size (N=14 541, 272 115 triples, 237 relations => E=544 230, R=474 after inverse
edges), seeded random-init Ultra weights (6 x 64d entity stack + 6 x 64d relation stack).
One STEP = one evaluation batch of B=16 test triples through ``predict`` (/root/reference/ultra/task.py:228-263):
relation-graph Bellman-Ford (6 rspmm) + tail pass + head pass over all N candidates (the reference: 2 x 6 rspmm +
epilogues + score MLP = 18 rspmm calls; here the tail and head queries share ONE 2B-wide Bellman-Ford, every score
bit-identical: tests/test_model_gpu.py), replayed as one hipGraph.  Unit of work = one edge message = one edge x one
batch element x 64 fp32 lanes; a step aggregates 12*E*B + 6*E_rel*B of them (sum of nnz * B over all
rspmm calls).  All inputs are resident in HBM before the timed region.  Multi-GPU: every rank holds the graph and
evaluates its own query batch (query sharding, no data-path collective) => weak scaling.

Reported beside ``value`` on the same line (every fraction can be recomputed from the numbers printed with it and
from profiles/):
* ``composition``      -- entity-graph and relation-graph edge messages per step and the entity-only rate; the step
                          with and without the first-layer frontier shortcut; predict + filtered ranking per step.
* ``roofline``         -- the dominant kernel of THIS workload (entity-graph forward, quad_kernel).  Its gathered
                          matrix (119 MB) lives in L2 / Infinity Cache, so it is priced as what it is, an L2-gather
                          kernel: algorithmic bytes per launch / launch time against the XCD-L2 peak (34.5 TB/s) and
                          the guide's measured ceiling for rows gathered from L2 (16.8-18.8 TB/s), plus the HBM
                          fraction on COMPULSORY bytes.  HIP events around exactly that kernel, on its stream.
* ``roofline_hbm``     -- config 5 (S-stress, 10 M nodes / 100 M edges / 1 k relations, B = 1) AT SIZE: the
                          DRAM-bound regime, algorithmic bytes against the 8 TB/s HBM peak ("S-stress
                          is the roofline reference").
* ``cpu_baseline``     -- the CPU oracle's row loop (restatement of the torchdrug CPU algorithm, kind "port") on all
                          host cores, one rspmm call of the bench graph.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

# MI355X_MICROARCH.md: chip-level parameters / L2 (per XCD) / Indexed rows: gather into LDS
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
L2_GATHER_CEILING_GBS = (16800.0, 18800.0)


class HipEvents:
    """hipEvent_t pairs created through the library (same HIP runtime as the kernels); the C ABI's profile hook
    records them on the kernel's own stream."""

    def __init__(self, lib):
        self.lib = lib
        self.pairs = []

    def new_pair(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.lib.ultra_rspmm_event_create(ctypes.byref(a)) == 0
        assert self.lib.ultra_rspmm_event_create(ctypes.byref(b)) == 0
        self.pairs.append((a, b))
        return a, b

    def elapsed_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            rc = self.lib.ultra_rspmm_event_elapsed_ms(a, b, ctypes.byref(ms))
            assert rc == 0, "event_elapsed_ms failed: %d (hip %d)" % (rc, self.lib.ultra_rspmm_last_hip_error())
            out.append(ms.value)
        return out


def bytes_algo(E, N, R, F):
    """every edge gathers one F-wide fp32 source row and a 12-byte (src, rel, w) triple; every
    destination row is written once; relation table and row pointers are read once."""
    return E * (4 * F + 12) + 4 * N * F + 4 * R * F + 4 * (N + 1)


def bytes_min(E, N, R, F):
    """SURVEY.md 8d, compulsory bytes (perfect reuse): input read once, output written once, indices, relation table."""
    return 8 * N * F + 4 * R * F + 12 * E + 4 * (N + 1)


def timed_kernel(lib, events, fn, n):
    """Average duration (ms) of the main rspmm kernel inside `fn`, from HIP events recorded by the library around it."""
    first = len(events.pairs)
    for _ in range(n):
        a, b = events.new_pair()
        lib.ultra_rspmm_profile_next(a, b)
        fn()
    torch.cuda.synchronize()
    ms = events.elapsed_ms()[first:]
    return float(np.mean(ms)), len(ms)


def cpu_baseline(graph_np, n_node, n_rel, F, budget_s=20.0):
    """The oracle's CSR row loop (OpenMP over rows) on ONE rspmm call of the bench graph; rank 0, N=1 only."""
    from oracle import oracle as O
    threads = len(os.sched_getaffinity(0))           # all host cores this process may use
    os.environ["OMP_NUM_THREADS"] = str(threads)
    O.build()
    rng = np.random.default_rng(1024)
    csr = O.coalesce_csr(graph_np["dst"], graph_np["src"], graph_np["rel"], None, n_node, n_node, n_rel)
    relation = rng.standard_normal((n_rel, F)).astype(np.float32)
    x = rng.standard_normal((n_node, F)).astype(np.float32)
    O.rspmm_forward(csr, relation, x, "add", "mul")            # warm-up (page-in, thread pool)
    times = []
    t_start = time.perf_counter()
    while len(times) < 10 and (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        O.rspmm_forward(csr, relation, x, "add", "mul")
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    # second CPU number: the reference's own O(E) formulation (ultra/layer.py:249-255,275-276) in PyTorch
    torch.set_num_threads(threads)
    ti = torch.from_numpy(csr.col.astype(np.int64)); tr = torch.from_numpy(csr.rel.astype(np.int64))
    td = torch.from_numpy(csr.row.astype(np.int64))
    tx, trel = torch.from_numpy(x), torch.from_numpy(relation)
    t0 = time.perf_counter()
    torch.zeros(n_node, F).index_add_(0, td, trel[tr] * tx[ti])
    t_torch = time.perf_counter() - t0
    return {"value": csr.n_edges * (F // 64) / med, "unit": "edges aggregated/s", "cores": threads,
            "torch_materialised_value": csr.n_edges * (F // 64) / t_torch,
            "kind": "port",
            "sample": "%d x one rspmm forward (add,mul) on S-fb15k237, E=%d, F=%d (B=%d); median %.3f s; "
                      "oracle/rspmm_oracle.c row loop, OpenMP, restatement of the torchdrug CPU algorithm"
                      % (len(times), csr.n_edges, F, F // 64, med)}


def stress_traffic():
    """HBM bytes per launch of the S-stress kernel from separate rocprofv3 --pmc passes of the same kernel and workload,
    committed under profiles/ (not measured in this run)."""
    path = os.path.join(ROOT, "profiles", "r02_traffic_stress.json")
    try:
        with open(path) as f:
            return json.load(f).get("hbm_bytes_per_launch"), "profiles/r02_traffic_stress.json (rocprofv3 --pmc passes, not this run)"
    except OSError:
        return None, None


def stress_roofline(dev, lib, n_node=10_000_000, n_triple=50_000_000, n_base_rel=500):
    """Config 5 at size: S-stress -- uniform triples + inverse edges => E = 100 M, R = 1 000, 64d,
    B = 1.  The gathered matrix (2.56 GB) cannot live in any cache: the HBM roofline of the operator."""
    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import functional as UF
    gen = torch.Generator(device=dev).manual_seed(1024)
    h = torch.randint(0, n_node, (n_triple,), device=dev, generator=gen)
    t = torch.randint(0, n_node, (n_triple,), device=dev, generator=gen)
    r = torch.randint(0, n_base_rel, (n_triple,), device=dev, generator=gen)
    t0 = time.perf_counter()
    csr = U.RelCSR(torch.cat([t, h]), torch.cat([h, t]), torch.cat([r, r + n_base_rel]), None, n_node, n_node,
                   2 * n_base_rel)
    plan = csr.fwd
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    del h, t, r
    F, R = 64, 2 * n_base_rel
    x = torch.randn(n_node, F, device=dev, generator=gen)
    relation = torch.randn(R, F, device=dev, generator=gen)
    for _ in range(3):
        UF.rspmm_forward(csr, relation, x, "add", "mul")
    events = HipEvents(lib)
    ms, n = timed_kernel(lib, events, lambda: UF.rspmm_forward(csr, relation, x, "add", "mul"), 12)
    # the chunked kernel the plan would run without the row-per-group kernel (A/B of the same launch)
    lib.ultra_rspmm_force_general_path(8)
    try:
        for _ in range(2):
            UF.rspmm_forward(csr, relation, x, "add", "mul")
        ms_chunked, _ = timed_kernel(lib, events, lambda: UF.rspmm_forward(csr, relation, x, "add", "mul"), 6)
    finally:
        lib.ultra_rspmm_force_general_path(0)
    E = csr.n_edges
    algo = bytes_algo(E, n_node, R, F)
    kernel = "rowgroup_kernel<add,mul,unit_w,624 of 1000 relation rows from LDS>" if plan.row_ptr is not None and plan.n_pieces == 0 \
        else "packed_kernel<FWD,add,mul,unit_w,VAR 2>"
    return {"bound": "hbm", "workload": "S-stress N=%d E=%d R=%d B=1 F=64 (input %.2f GB)" % (n_node, E, R, n_node * F * 4 / 1e9),
            "kernel": kernel, "launches_timed": n, "kernel_ms": ms, "algorithmic_bytes": algo,
            "achieved": algo / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": stress_traffic()[0],
            "traffic_source": stress_traffic()[1],
            "edges_per_s": E / (ms * 1e-3), "chunked_kernel_ms": ms_chunked, "plan_build_s": build_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16, help="queries per step (reference inference batch: 16)")
    ap.add_argument("--workload", default="S-fb15k237")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stress", dest="stress", action="store_false",
                    help="skip config 5 at size (S-stress: 10 M nodes / 100 M edges, ~15 s) reported as roofline_hbm")
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python instead of replaying a hipGraph")
    ap.add_argument("--mrr-queries", type=int, default=500,
                    help="seeded test triples ranked after the timed region (500 = the reference's fast_test, pretrain_3g.yaml:56)")
    ap.add_argument("--finetune-steps", type=int, default=50, help="seeded fine-tuning steps before the second MRR")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ULTRA_BENCH_SHARE_GPU=1 (development only): all ranks on the visible GPU(s), gloo instead of RCCL -- lets the
    # multi-rank code path (sharding, barriers, max-over-ranks timing) run on a one-GPU box
    share = os.environ.get("ULTRA_BENCH_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count()              # (does not initialise the GPU)
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if share:
        local_rank %= n_dev
    # the process group comes first: RCCL is initialised before this process makes any other GPU call
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    assert world == args.gpus or world == 1, "launch N ranks with torch.distributed.run for --gpus N"

    import ultra_torchdrug_amd as U
    from ultra_torchdrug_amd import layer as UL
    from ultra_torchdrug_amd.data import synthetic_triples, DEFAULT_SEED
    from ultra_torchdrug_amd.graph import Graph
    from ultra_torchdrug_amd.task import build_ultra
    lib = U.require_library()
    torch.backends.cuda.matmul.allow_tf32 = False          # script/run_full.py:19-20
    torch.backends.cudnn.allow_tf32 = False

    # ---------------- graph, split, model (identical on every rank) ----------------
    from ultra_torchdrug_amd.data import SHAPES
    n_node, n_fact, n_rel = SHAPES[args.workload]
    n_test = min(2048, n_fact // 20)
    # n_fact + n_test DISTINCT triples of the same distribution: the first n_fact are the fact graph (E = 2 * n_fact
    # exactly, as BASELINE.md states), the rest are held-out test queries -- never edges of the message-passing graph,
    # but part of `graph`, which the ranking is filtered against (the transductive protocol, task.py:31-63).
    triples, _, _ = synthetic_triples((n_node, n_fact + n_test, n_rel), DEFAULT_SEED,
                                      alpha=0.0 if args.workload == "S-stress" else 1.0)
    fact_mask = np.zeros(len(triples), dtype=bool)
    fact_mask[:n_fact] = True
    test_idx = np.arange(n_fact, n_fact + n_test)
    graph = Graph(torch.from_numpy(triples), num_node=n_node, num_relation=n_rel)
    torch.manual_seed(DEFAULT_SEED)
    task = build_ultra(n_rel)
    task.preprocess(graph, torch.from_numpy(fact_mask))
    task.to(dev).eval()
    und = task.model._undirected(task.fact_graph)
    _ = und.relcsr.fwd                  # coalesce + sort + chunk schedule: once per graph (torchdrug: every rspmm call)
    torch.cuda.synchronize()
    t_plan = time.perf_counter()        # steady-state cost of that build (the first one above also warms rocPRIM up)
    _ = U.RelCSR.from_edge_list(und.edge_list, und.edge_weight, und.num_node, und.num_relation).fwd
    torch.cuda.synchronize()
    plan_build_ms = 1e3 * (time.perf_counter() - t_plan)
    E, R2 = und.relcsr.n_edges, und.num_relation
    E_rel = task.rel_graphs[0].relcsr.n_edges
    for g in (und, task.rel_graphs[0]):
        _ = g.relcsr.fwd                                      # plans built before the timed region
        _ = g.relcsr.frontier_index
    for g in (task.graph, task.fact_graph):
        g.completion_keys(0), g.completion_keys(1)            # sorted filter keys: once per graph
    B = args.batch
    F = B * 64
    # predict() scores tails and heads in ONE Bellman-Ford over 2B queries (task.fuse_sides): 6 entity launches of
    # width 2F per step instead of 12 of width F -- the same edge messages
    Fk = 2 * F
    entity_edges_per_step = 12 * E * B
    rel_edges_per_step = 6 * E_rel * B
    edges_per_step = entity_edges_per_step + rel_edges_per_step

    # each rank evaluates its own strided shard of the seeded test triples (DistributedSampler-style)
    test = torch.from_numpy(triples[test_idx]).to(dev)
    shard = test[rank::world]
    n_batches = max(len(shard) // B, 1)

    # profile hook: HIP events around the entity-graph forward kernel (the dominant kernel), recorded by the library
    # on the kernel's own stream.  --eager: one pair per launch of the timed region.  Default (hipGraph replay): event
    # records cannot be captured with the HIP runtime PyTorch bundles, so the same kernel is launched eagerly, with the
    # step's own shapes and fused boundary epilogue, right after the timed region and timed there.
    events = HipEvents(lib)
    from ultra_torchdrug_amd import functional as UF
    real_forward = UF.rspmm_forward
    state = {"on": False}

    def timed_forward(csr, relation, input, sum="add", mul="mul", add_rows=None, boundary=None):
        if state["on"] and csr is und.relcsr:
            a, b = events.new_pair()
            lib.ultra_rspmm_profile_next(a, b)
        return real_forward(csr, relation, input, sum, mul, add_rows, boundary)

    UF.rspmm_forward = timed_forward

    # the evaluation batch is replayed as one hipGraph (engine.GraphedPredict); --eager times the un-captured path
    from ultra_torchdrug_amd.engine import GraphedPredict

    def capture():
        with torch.no_grad():
            task.predict(shard[:B])                 # plans, kernel attributes, allocator: before the capture
        try:
            return GraphedPredict(task, shard[:B], warmup=0)
        except Exception as err:                    # capture refused (driver / runtime): time the eager path
            print("bench: hipGraph capture failed (%s); falling back to eager launches" % err, file=sys.stderr)
            torch.cuda.synchronize()
            return None

    graphed = None if args.eager else capture()

    def step(i, g=None):
        g = graphed if g is None else g
        batch = shard[(i % n_batches) * B:(i % n_batches) * B + B]
        return task.predict(batch) if g is None else g(batch)

    def replays_equal_eager(g, n=3):
        """Replays of the captured step on batches OTHER than the captured one give the scores of eager launches (a
        graph node replayed out of order shows here, not on the captured batch)."""
        if g is None:
            return None
        same = True
        for i in range(1, n + 1):
            batch = shard[(i % n_batches) * B:(i % n_batches) * B + B]
            got = g(batch).clone()
            same = same and bool(torch.equal(got, task.predict(batch)))
        return same

    def time_steps(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        state["on"] = graphed is None
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        state["on"] = False

        # ---- after the timed region: the dominant kernel alone, and the other per-step numbers
        gen = torch.Generator(device=dev).manual_seed(DEFAULT_SEED)
        xk = torch.randn(n_node, Fk, device=dev, generator=gen)
        rk = torch.randn(R2, Fk, device=dev, generator=gen)
        # the layer's sparse boundary: the heads and tails of a REAL batch (Zipf hubs: a uniformly drawn boundary would
        # flatter the first-layer frontier kernel by an order of magnitude)
        bk = (torch.cat([shard[:B, 0], shard[:B, 1]])[:Fk // 64].to(torch.int32).contiguous(),
              torch.randn(Fk // 64, 64, device=dev, generator=gen))
        if graphed is not None:
            for _ in range(4):
                real_forward(und.relcsr, rk, xk, "add", "mul", None, bk)
            state["on"] = True
            for _ in range(48):
                timed_forward(und.relcsr, rk, xk, "add", "mul", None, bk)
            state["on"] = False
            torch.cuda.synchronize()
        kernel_ms = events.elapsed_ms()
        n_side = min(args.steps, 50)
        eager_ms = time_steps(lambda i: task.predict(shard[:B]), 5) if graphed is not None else 1e3 * elapsed / args.steps
        # predict + filtered ranking (what engine.evaluate does per batch): the ranks come from the sorted completion keys
        # on the device, nothing but (B, 2) int64 leaves it
        rank_ms = time_steps(lambda i: task.rank_batch(shard[:B], pred=step(i)), n_side)
        # the same step without the first-layer frontier shortcut (every layer walks all E edges)
        UL.FRONTIER_FIRST_LAYER = False
        try:
            plain = None if args.eager else capture()
            for i in range(5):
                step(i, plain)
            no_frontier_ms = time_steps(lambda i: step(i, plain), n_side)
        finally:
            UL.FRONTIER_FIRST_LAYER = True
        frontier_ms, _ = timed_kernel(lib, events, lambda: UF.rspmm_frontier(und.relcsr, rk, bk), 10)
        # the same step with the relation representations of all R relations computed once per evaluation run
        # (task.cache_relation_representations, what engine.evaluate does): identical scores, the relation stack leaves
        # the per-batch path.  Reported beside `value`, never as `value`.
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        task.cache_relation_representations(B)
        torch.cuda.synchronize()
        cache_build_ms = 1e3 * (time.perf_counter() - t_c)
        try:
            cached = None if args.eager else capture()
            for i in range(5):
                step(i, cached)
            cached_ms = time_steps(lambda i: step(i, cached), n_side)
            replay_same_cached = replays_equal_eager(cached)
        finally:
            task.clear_relation_cache()
        replay_same = replays_equal_eager(graphed)
    UF.rspmm_forward = real_forward

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # fabric bytes per launch of the dominant kernel: separate rocprofv3 --pmc passes of the same kernel and workload
    # (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction), committed under profiles/ -- not measured in this run
    traffic, traffic_source = None, None
    for name in ("r02_traffic_fwd_fb15k237.json", "traffic_fwd_fb15k237.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if args.workload == "S-fb15k237" and args.batch == 16 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("F") == Fk:
                traffic, traffic_source = tj.get("hbm_bytes_per_launch"), "profiles/%s (rocprofv3 --pmc passes, not this run)" % name
                break

    k_avg_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    algo = bytes_algo(E, n_node, R2, Fk)
    compulsory = bytes_min(E, n_node, R2, Fk)
    achieved = algo / (k_avg_ms * 1e-3) / 1e9

    # ---------------- MRR (after the timed region): HIP path, and HIP vs CPU-oracle path on the same weights --------
    def mrr_of(t, queries):
        with torch.no_grad():
            rk = torch.cat([t.rank_batch(queries[i:i + B]) for i in range(0, len(queries), B)])
        return rk

    def oracle_check(label):
        """First batch once more with the CPU oracle in place of every HIP kernel (checker, never the product)."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_ops import oracle_rspmm
        cpu_task = build_ultra(n_rel)
        cpu_task.load_state_dict({k: v.cpu() for k, v in task.state_dict().items()})
        cpu_task.preprocess(Graph(torch.from_numpy(triples), num_node=n_node, num_relation=n_rel),
                            torch.from_numpy(fact_mask)).eval()
        batch_cpu = shard[:B].cpu()
        with torch.no_grad(), oracle_rspmm(None):
            pred_cpu = cpu_task.predict(batch_cpu)
            ranks_cpu = cpu_task.get_ranking(pred_cpu, cpu_task.target(batch_cpu))
        with torch.no_grad():
            pred_gpu = task.predict(shard[:B])
            ranks_gpu = task.rank_batch(shard[:B], pred=pred_gpu).cpu()
        return {"weights": label, "queries": int(B), "mrr_hip": float((1.0 / ranks_gpu.float()).mean()),
                "mrr_cpu_oracle": float((1.0 / ranks_cpu.float()).mean()),
                "ranks_identical": int((ranks_gpu == ranks_cpu).sum()), "ranks_total": int(ranks_cpu.numel()),
                "max_abs_score_diff": float((pred_gpu.cpu() - pred_cpu).abs().max())}

    def metrics_of(ranks):
        """MR / MRR / Hits@k of filtered ranks (ultra/task.py:317-351), tails and heads together."""
        r = ranks.float().flatten()
        return {"queries": int(ranks.shape[0]), "mr": float(r.mean()), "mrr": float((1.0 / r).mean()),
                "hits@1": float((r <= 1).float().mean()), "hits@3": float((r <= 3).float().mean()),
                "hits@10": float((r <= 10).float().mean())}

    # ---------------- whole evaluation runs over the seeded test triples (engine.evaluate), three ways, same ranks ----
    eval_runs = None
    if rank == 0 and world == 1 and args.mrr_queries > 0:
        from ultra_torchdrug_amd import engine
        eval_runs = {"triples": int(len(test))}
        checks = []
        for name, kw in (("reference_loop", dict(cache_relations=False, unique_queries=False)),
                         ("cached_relations", dict(cache_relations=True, unique_queries=False)),
                         ("cached_relations_unique_queries", dict(cache_relations=True, unique_queries=True))):
            engine.evaluate(task, test[:4 * B], batch_size=B, **kw)        # first-use costs (code object loads) stay outside
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            _, ranking = engine.evaluate(task, test, batch_size=B, **kw)
            torch.cuda.synchronize()
            eval_runs[name + "_ms"] = 1e3 * (time.perf_counter() - t_e)
            checks.append(ranking)
        eval_runs["ranks_identical"] = bool(all(torch.equal(checks[0], c) for c in checks[1:]))

    mrr = mrr_tuned = None
    metrics = metrics_tuned = None
    mrr_check = []
    train_ms = None
    if args.mrr_queries > 0:
        # the seeded test triples are strided over the ranks; one all_gather of (n, 2) int64 ranks
        nq = min(-(-args.mrr_queries // world), len(test) // world)       # the same count on every rank

        def gathered(ranks):
            if world == 1:
                return ranks
            ranks = ranks.cpu() if share else ranks            # (development switch: gloo works on host tensors)
            parts = [torch.empty_like(ranks) for _ in range(world)]
            dist.all_gather(parts, ranks)
            return torch.cat(parts)

        metrics = metrics_of(gathered(mrr_of(task, shard[:nq])))
        mrr = metrics["mrr"]
        check = rank == 0 and world == 1 and not args.no_cpu_baseline
        if check:
            mrr_check.append(oracle_check("seeded random init (td_ultra_3g/4g.pth are missing blobs)"))
        if rank == 0 and world == 1 and args.finetune_steps > 0:
            # a short seeded fine-tuning run (config 3's step: rspmm fwd+bwd, AdamW 5e-4, 128 strict negatives) so that
            # the parity check also sees trained weights and an MRR that is not the random-init one
            from ultra_torchdrug_amd import engine
            task.train()
            opt = torch.optim.AdamW(task.parameters(), lr=5e-4)
            facts = torch.from_numpy(triples[:n_fact]).to(dev)
            pick = np.random.default_rng(DEFAULT_SEED)
            torch.manual_seed(DEFAULT_SEED)
            # the whole step (strict negatives, edge removal, forward, backward) replays as one hipGraph
            idx = torch.from_numpy(pick.choice(n_fact, B, replace=False)).to(dev)
            graphed_step = engine.GraphedTrainStep(task, opt, facts[idx])
            # the batches of all steps are drawn before the clock starts (a host-side draw without replacement over
            # 272 k facts costs more than the step)
            batches = [facts[torch.from_numpy(pick.choice(n_fact, B, replace=False)).to(dev)]
                       for _ in range(args.finetune_steps)]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for batch in batches:
                graphed_step(batch)
            torch.cuda.synchronize()
            train_ms = 1e3 * (time.perf_counter() - t1) / args.finetune_steps
            del graphed_step
            task.eval()
            metrics_tuned = metrics_of(gathered(mrr_of(task, shard[:nq])))
            mrr_tuned = metrics_tuned["mrr"]
            if check:
                mrr_check.append(oracle_check("after %d seeded fine-tuning steps on the HIP path" % args.finetune_steps))

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        result = {
            "metric": "edges aggregated/sec, FB15k237-shaped 6L x 64d rspmm Bellman-Ford (predict: 18 rspmm/batch)",
            "value": edges_per_step * args.steps * world / elapsed,
            "unit": "edges aggregated/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s N=%d E=%d R=%d B=%d F=%d, 6+6 layers x 64d, predict() tail+head over all "
                                   "entities (E_rel=%d)" % (args.workload, n_node, E, R2, B, F, E_rel),
                       "global_batch": B * world, "parallelism": "query-sharded replicas x%d" % world,
                       "launch": "eager" if graphed is None else "one hipGraph replay per step (engine.GraphedPredict)",
                       "device": "%s, %d CUs" % (torch.cuda.get_device_name(dev),
                                                 torch.cuda.get_device_properties(dev).multi_processor_count)},
            "edges_per_step": edges_per_step,
            "composition": {
                "entity_graph_edges_per_step": entity_edges_per_step,
                "relation_graph_edges_per_step": rel_edges_per_step,
                "relation_graph_note": "E_rel = %d over %d relation nodes and 4 edge types: with independently drawn Zipf "
                                       "heads / tails / relations (SURVEY 8d generator) every pair of relations co-occurs, "
                                       "so the relation graph is complete -- LDS-resident, not a sparse-gather workload"
                                       % (E_rel, R2),
                "value_entity_only": entity_edges_per_step * args.steps * world / elapsed,
                "ms_per_step_without_first_layer_frontier": no_frontier_ms,
                "value_without_first_layer_frontier": edges_per_step * world / (no_frontier_ms * 1e-3),
                "first_layer_frontier_kernel_ms": frontier_ms,
                "ms_per_step_with_cached_relation_representations": cached_ms,
                "value_entity_only_with_cached_relation_representations":
                    entity_edges_per_step / (cached_ms * 1e-3) if cached_ms else None,
                "relation_cache_build_ms": cache_build_ms,
                "evaluate_test_set": eval_runs,
                "graph_replays_identical_to_eager": {"per_batch_relations": replay_same, "cached_relations": replay_same_cached},
                "relation_cache_note": "opt-in (engine.evaluate default for long runs): the relation representations of a "
                                       "query depend on its relation only, so all R tables are computed once per "
                                       "evaluation run and a batch picks its rows -- bit-identical scores "
                                       "(tests/test_model_gpu.py); the relation-graph edge messages are then no longer "
                                       "aggregated per batch, so this line reports the entity-graph rate only",
                "first_layer_note": "layer 1 reads the boundary (zero outside one row per query): its E * B edge messages are "
                                    "+-0 except on the out-edges of the boundary nodes; the frontier kernel adds exactly "
                                    "those, bit-identically (tests/test_frontier_sampler_gpu.py); `value` counts the "
                                    "layer's edges as aggregated either way",
                "predict_plus_filtered_rank_ms_per_step": rank_ms,
                "value_predict_plus_rank": edges_per_step * world / (rank_ms * 1e-3),
                "finetune_ms_per_step": train_ms,
                "finetune_note": "config 3's step on this graph (B = %d, 128 strict negatives, AdamW): negatives, edge removal, "
                                 "forward and backward replayed as one hipGraph (engine.GraphedTrainStep)" % B,
            },
            "plan_build_ms": plan_build_ms,
            "eager_ms_per_step": eager_ms,
            "rspmm_kernel_only": {"kernel": "quad_kernel<FWD,add,mul,unit_w> (entity graph, F = %d: tail and head queries of the batch in one launch)" % Fk,
                                  "launches_timed": len(kernel_ms), "avg_ms": k_avg_ms,
                                  "timed": "every launch of the timed region" if graphed is None else
                                           "%d eager launches of the same kernel/shapes right after the timed region "
                                           "(event records cannot be captured into the hipGraph with this HIP runtime)" % len(kernel_ms),
                                  "edges_per_s": E * (Fk // 64) / (k_avg_ms * 1e-3) if kernel_ms else None},
            "roofline": {"bound": "l2-gather", "kernel": "quad_kernel<FWD,add,mul,unit_w,8>",
                         "achieved": achieved, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": achieved / L2_PEAK_GBS,
                         "algorithmic_bytes": algo, "kernel_ms": k_avg_ms,
                         "l2_gather_ceiling": list(L2_GATHER_CEILING_GBS),
                         "frac_of_l2_gather_ceiling": achieved / L2_GATHER_CEILING_GBS[1],
                         "hbm_compulsory_bytes": compulsory,
                         "hbm_frac_on_compulsory_bytes": compulsory / (k_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "note": "the gathered matrix (%.0f MB) is L2 / Infinity-Cache resident at this size: algorithmic "
                                 "bytes (SURVEY 8d) are priced against the XCD-L2 peak (MI355X_MICROARCH.md: 34.5 TB/s; "
                                 "measured ceiling for rows gathered from L2: 16.8-18.8 TB/s), HBM against the "
                                 "compulsory bytes; the DRAM-bound regime is roofline_hbm" % (n_node * Fk * 4 / 1e6)},
            "mrr_hip": mrr,
            "mrr_hip_after_finetune": mrr_tuned,
            "metrics_hip": metrics,
            "metrics_hip_after_finetune": metrics_tuned,
            "mrr_check": mrr_check,
        }
        if world == 1 and not args.no_cpu_baseline:
            und_np = {"dst": und.edge_list[:, 1].cpu().numpy(), "src": und.edge_list[:, 0].cpu().numpy(),
                      "rel": und.edge_list[:, 2].cpu().numpy()}
            result["cpu_baseline"] = cpu_baseline(und_np, n_node, R2, F)
        if world == 1 and args.stress:
            del xk, rk
            torch.cuda.empty_cache()
            result["roofline_hbm"] = stress_roofline(dev, lib)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
