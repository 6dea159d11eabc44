"""profiles/<tag>_configs.md from bench.py's JSON line (<tag>_bench_under_rocprof.json): one row per BASELINE config."""
import json
import sys


def main(path, out):
    j = json.load(open(path))
    rows = ["| config | what | shape | this run |", "|---|---|---|---|"]
    for c in j["config"]["configs"]:
        k = c["config"]
        if k == 1:
            got = "predict + filtered rank %.0f ms per batch on the CPU kernels" % c["predict_plus_rank_ms_per_batch"]
        elif k == 2:
            got = ("predict %.3f ms per batch; entity forward kernel %.1f us = %.1f TB/s algorithmic (%.2f of the XCD-L2 peak)"
                   % (c["predict_ms_per_batch"], c["entity_fwd_kernel_us"], c["entity_fwd_kernel_algorithmic_GBps"] / 1e3,
                      c["entity_fwd_kernel_frac_of_l2_peak"]))
            if "entity_fwd_kernel_us_one_side_width" in c:
                got += "; %.1f us at one side's width (F = B * 64)" % c["entity_fwd_kernel_us_one_side_width"]
        elif k == 3:
            s = c["finetune_step"]
            got = ("operator fwd %.1f us, bwd %.1f us; fine-tune step median %.2f ms (p10 %.2f, p90 %.2f, max %.2f; n = %d)"
                   % (c["operator_fwd_us"], c["operator_bwd_us"], s["median_ms"], s["p10_ms"], s["p90_ms"], s["max_ms"], s["n"]))
        elif k == 4:
            s = c["step"]
            got = ("captured step median %.1f ms (min %.1f, max %.1f; graphs drawn on rank 0: %s); %d rank(s): %.1f ms per step max over ranks"
                   % (s["median_ms"], s["min_ms"], s["max_ms"], c.get("graphs_drawn_rank0", c.get("graphs_drawn", "?")), c.get("n_gpus", 1),
                      c.get("step_ms_max_over_ranks", s["median_ms"])))
            if "eager_step_ms" in c:
                got += "; eager step %.1f ms" % c["eager_step_ms"]
            if "allreduce_exposed_ms_per_step" in c:
                got += "; gradient all-reduce exposed %.2f ms per step (mode %s)" % (c["allreduce_exposed_ms_per_step"], c.get("step_mode"))
        else:
            got = "operator fwd %.2f ms = %.2e edges/s = %.3f of the 8 TB/s HBM peak" % (c["operator_fwd_ms"], c["edges_per_s"], c["frac_of_hbm_peak"])
            if "predict_ms" in c:
                lb = c.get("layer_breakdown_b1", {})
                got += ("; INFERENCE: whole predict %.1f ms for one triple (2 queries), %.1f ms for four; one entity layer %.2f ms as ONE launch = %.3f of "
                        "the HBM peak on its %.1f GB (two launches: %.2f + %.2f ms); filtered rank %.2f ms; operator at B = 4 %.1f ms = %.3f"
                        % (c["predict_ms"], c["predict_ms_b4"], lb.get("layer_ms", float("nan")), c.get("frac_of_hbm_peak_whole_layer", float("nan")),
                           lb.get("layer_algorithmic_bytes", 0) / 1e9, lb.get("two_launch_rspmm_ms", float("nan")),
                           lb.get("two_launch_epilogue_ms", float("nan")), c.get("filtered_rank_ms", float("nan")),
                           c.get("operator_b4_ms", float("nan")), c.get("operator_b4_frac", float("nan"))))
        rows.append("| %d | %s | %s | %s |" % (k, c["name"], c["shape"], got))
    head = ("Per-config timings of `%s` (one `bench.py` run under rocprofv3; headline: %.3f ms per evaluation batch, "
            "%.3e entity-graph edge messages/s).\n\n" % (path.split("/")[-1], j["ms_per_step"], j["value"]))
    open(out, "w").write(head + "\n".join(rows) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
