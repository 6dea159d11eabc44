"""Time ultra_combine_backward_fused_f32 alone (rows = WN18RR x B=16) for the library named by ULTRA_RSPMM_LIB."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ULTRA_BINDING"] = "ctypes"
import numpy as np, torch
from ultra_torchdrug_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40943 * 16
g = torch.Generator(device=dev).manual_seed(0)
x, u, go = (torch.randn(rows, 64, device=dev, generator=g) for _ in range(3))
w = torch.randn(64, 128, device=dev, generator=g) * 0.1
b, gam, bet = torch.randn(64, device=dev, generator=g), torch.rand(64, device=dev, generator=g) + 0.5, torch.randn(64, device=dev, generator=g)
n_waves = ctypes.c_int(0)
lib.ultra_combine_backward_fused_waves(0, rows, ctypes.byref(n_waves))
ws = torch.empty(n_waves.value * (64 * 128 + 192), device=dev)
dx, du = torch.empty_like(x), torch.empty_like(u)
dw, db, dg, dbt = torch.empty(64, 128, device=dev), torch.empty(64, device=dev), torch.empty(64, device=dev), torch.empty(64, device=dev)
def run():
    rc = lib.ultra_combine_backward_fused_f32(x.data_ptr(), u.data_ptr(), w.data_ptr(), b.data_ptr(), gam.data_ptr(), bet.data_ptr(),
        1e-5, 1, 1, go.data_ptr(), dx.data_ptr(), du.data_ptr(), dw.data_ptr(), db.data_ptr(), dg.data_ptr(), dbt.data_ptr(),
        ws.data_ptr(), ws.numel() * 4, rows, 64, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 1e3)
print("%s rows=%d fused backward + reduce: median %.1f us min %.1f us" % (os.path.basename(_lib.LIB_PATH), rows, np.median(ts), min(ts)))
