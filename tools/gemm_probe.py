import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sbv2_api_amd import _lib
l = _lib.lib()
for (m, n, kk) in ((3072, 2048, 1024), (1024, 2048, 1024), (4096, 2048, 1024), (1024, 2048, 4096), (576, 28704, 192), (192, 28704, 192), (192, 8224, 1024)):
    ms = C.c_float()
    _lib.check(l.sbv2_debug_time_conv1d(0, kk, m, 1, n, 1, 20, C.byref(ms)))
    print(f"GEMM M={m} N={n} K={kk}: {ms.value*1e3:8.1f} us  {2.0 * m * n * kk / ms.value / 1e9:8.1f} TFLOP/s", flush=True)
