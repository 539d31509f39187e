import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tssep_amd import _lib, hip_ops as h
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for M, N, K in ((388608, 2400, 513), (97152, 2400, 1280)):
    Kp = h.round_up(K, 16)
    planes = [(torch.randn(r, Kp, device="cuda") * 0.1).to(torch.bfloat16) for r in (M, M, N, N)]
    C = torch.empty(M, N, device="cuda")
    row = {"M": M, "N": N, "K": K}
    for base in (2, 3, 12, 13):
        for name, fl in (("full", 0), ("no_mfma", 256), ("no_frag", 512), ("no_dma", 1024), ("no_bar", 2048),
                         ("mfma_only", 512 + 1024 + 2048), ("data_only", 256), ("dma_bar_only", 256 + 512)):
            ms = timeit(lambda: h.check(L.tssep_probe_gemm_presplit(planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(), planes[3].data_ptr(), C.data_ptr(), M, N, K, N, base | fl, st()), "p"))
            row[f"r{base}_{name}"] = round(ms, 3)
    print(json.dumps(row), flush=True)
