"""Gather-rate micro-benchmark (tools/micro/gather_rate.hip): TB/s of row gathers out of L2 / Infinity Cache by access width."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "wisecondor_amd", "ab", "libgather.so"))
lib.gather_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                           ctypes.c_void_p, ctypes.c_void_p]
n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 55337
n_ref = 112
rng = np.random.RandomState(0)
for n_bins in (n_rows, 2 * n_rows):
    idx = torch.from_numpy(rng.randint(0, n_rows, size=(n_bins, n_ref)).astype(np.int32)).cuda()
    out = torch.zeros((n_bins, 64), dtype=torch.float64, device="cuda")
    for mode, row_bytes, useful in ((0, 1000, 512), (0, 1024, 512), (1, 1024, 1024), (2, 1024, 1024), (3, 1024, 1024)):
        X = torch.rand((n_rows, row_bytes // 8), dtype=torch.float64, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            lib.gather_run(mode, X.data_ptr(), row_bytes, idx.data_ptr(), n_ref, n_bins, out.data_ptr(), s)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            rc = lib.gather_run(mode, X.data_ptr(), row_bytes, idx.data_ptr(), n_ref, n_bins, out.data_ptr(), s)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        nbytes = float(n_bins) * n_ref * useful
        print("bins %6d mode %d row %4d B: %.3f ms, %.2f TB/s (%.1f B/clk/CU at 2.2 GHz), rc %d, check %.6g"
              % (n_bins, mode, row_bytes, ms, nbytes / ms / 1e9, nbytes / (ms * 1e-3) / 256 / 2.2e9, rc, float(out.sum())))
