import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from scene_graph_commonsense_amd import _lib
lib = _lib.load()
M, N, K = 32256, 65536, 4096
A = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
B = (torch.rand(N, K, device="cuda") - 0.5).bfloat16()
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
def run():
    return lib.sgc_dbg_gemm_nt(1, _lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, ctypes.c_long(K), ctypes.c_long(K), ctypes.c_long(N), None, _lib.stream_ptr())
for _ in range(2): assert run() == 0
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(4): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 4
print("EPI_LDS=%s SKIP=%s  %.3f ms  %.0f TFLOP/s" % (os.environ.get("SGC_EPI_LDS", "1"), os.environ.get("SGC_SKIP_EPI", "0"), ms, 2.0 * M * N * K / ms / 1e9))
