"""Does a hipMemsetAsync captured into a hipGraph (a memset NODE) keep its place between the kernel nodes around it?

Context: ultra_rspmm_frontier_f32 used hipMemsetAsync for its zero fill; inside a short captured graph (few kernels in
front of it) replays on batches other than the captured one produced wrong first-layer outputs, and the fault went away
with a fill KERNEL in its place.  The pattern: a temporary is written, read and freed; `out` is allocated (inside the
graph's memory pool it takes over the temporary's block); memset(out) -> a kernel writes a few rows of out -> a kernel
reads out; captured once, replayed with other inputs, compared with eager execution.  Observed on MI355X / ROCm 7.2
with the HIP runtime of PyTorch 2.10+rocm7.0: 19 of 20 replays differ with hipMemsetAsync, 0 of 20 with a fill kernel
(and 0 of 20 with hipMemsetAsync when the temporary is left out: no block reuse, no hazard).
    python tools/debug/memset_node_repro.py [memset|memcpy|all]      (one variant per process: the fault depends on
    which blocks the graph's memory pool hands out, which earlier captures in the same process change)
"""
import ctypes

import torch


def main(which):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    dev = torch.device("cuda:0")
    n, f = 300, 1024
    static_idx = torch.randint(0, n, (16,), device=dev)
    static_val = torch.randn(16, f, device=dev)

    def body(idx, val, use_memset):
        # a temporary that is written, read and freed BEFORE `out` is allocated: inside the graph's memory pool `out`
        # takes over its block, so a fill that ran early (or not at all) would show
        tmp = val.repeat(n // 16 + 1, 1)[:n] * 3.0
        carry = tmp.sum()
        del tmp
        out = torch.empty(n, f, device=dev)
        if use_memset:
            rc = hip.hipMemsetAsync(out.data_ptr(), 0, out.numel() * 4, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        else:
            out.zero_()
        out.index_copy_(0, idx, val)               # a kernel that writes a few rows
        return out * 2.0 + 1.0 + carry * 0.0        # a kernel that reads everything

    zeros_src = torch.zeros(n, f, device=dev)

    def body_copy(idx, val, use_memcpy):
        """The same with a device-to-device copy of zeros in the fill's place (aten::copy_ of a contiguous tensor is a
        hipMemcpyAsync, i.e. a memcpy NODE under capture)."""
        tmp = val.repeat(n // 16 + 1, 1)[:n] * 3.0
        carry = tmp.sum()
        del tmp
        out = torch.empty(n, f, device=dev)
        if use_memcpy:
            out.copy_(zeros_src)
        else:
            torch.mul(zeros_src, 1.0, out=out)      # an elementwise kernel
        out.index_copy_(0, idx, val)
        return out * 2.0 + 1.0 + carry * 0.0

    for use_memcpy in ([True, False] if which in ("memcpy", "all") else []):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                body_copy(static_idx, static_val, use_memcpy)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            res = body_copy(static_idx, static_val, use_memcpy)
        bad = 0
        for trial in range(20):
            idx = torch.randperm(n, device=dev)[:16]
            val = torch.randn(16, f, device=dev)
            static_idx.copy_(idx)
            static_val.copy_(val)
            g.replay()
            torch.cuda.synchronize()
            want = body_copy(idx, val, False)
            bad += int(not torch.equal(res, want))
        print("zero fill by %s: %d of 20 replays differ from eager" % ("copy_ of zeros (memcpy node)" if use_memcpy else "elementwise kernel", bad))

    for use_memset in ([True, False] if which in ("memset", "all") else []):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                body(static_idx, static_val, use_memset)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            res = body(static_idx, static_val, use_memset)
        bad = 0
        for trial in range(20):
            idx = torch.randperm(n, device=dev)[:16]
            val = torch.randn(16, f, device=dev)
            static_idx.copy_(idx)
            static_val.copy_(val)
            g.replay()
            torch.cuda.synchronize()
            want = body(idx, val, False)
            bad += int(not torch.equal(res, want))
        print("zero fill by %s: %d of 20 replays differ from eager" % ("hipMemsetAsync (memset node)" if use_memset else "fill kernel", bad))


if __name__ == "__main__":
    import sys
    main(sys.argv[1] if len(sys.argv) > 1 else "all")
