#!/usr/bin/env python3
"""How many 256-thread blocks with a given dynamic LDS size are resident per CU?  (spin kernel, 1 ms each; a grid of
CUs x k blocks takes ceil(k / resident) ms.)  Shows the LDS allocation granule hipOccupancy does not account for."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vistrace_amd as va
P = C.CDLL(os.path.join(ROOT, "scripts", "_build", "libprobe.so"))
P.probe_spin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_int]
dev = torch.device("cuda", 0)
eng = va.Engine(0)
ncu = eng.get_option("cu_count")
sink = torch.zeros(4, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(dev)
for lds in (26880, 26896, 25600, 25616, 27136, 28160, 20480, 20496, 23040, 23056):
    for k in (5, 6, 7, 8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            e0.record()
            P.probe_spin(s.cuda_stream, ncu * k, lds, 100000, sink.data_ptr(), 0)
            e1.record()
        torch.cuda.synchronize()
        print(f"LDS {lds:6d} B/block, {k} blocks per CU in the grid: {e0.elapsed_time(e1):5.2f} ms", flush=True)
