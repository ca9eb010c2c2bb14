"""Kernel tuning harness: runs the stand-alone sorter (gs_sort_bench) of a given library build at the
config-C element count and prints the mean sort time.  Run under rocprofv3 --kernel-trace --stats to
get per-kernel durations.   python tools/sort_tune.py <lib.so> [n] [iters]"""
import ctypes as C, sys, os
lib = C.CDLL(os.path.abspath(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 13_121_624
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
h = C.c_void_p()
lib.gs_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
assert lib.gs_create(None, C.byref(h)) == 0
lib.gs_sort_bench.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
ms, ok = C.c_float(), C.c_uint32()
rc = lib.gs_sort_bench(h, n, 8160, iters, 1, C.byref(ms), C.byref(ok))
print(f"{sys.argv[1]}: rc={rc} n={n} sort_ms={ms.value:.4f} sorted_ok={ok.value} Melem/s={n/ms.value/1e3:.0f} GB/s(28B*12)={28*12*n/ms.value/1e6:.0f}")
