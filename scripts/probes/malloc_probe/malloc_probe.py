"""How long a fresh hipMalloc of N GB and the first kernel over it take (a fresh process each time: python malloc_probe.py GB [GB ...])."""
import sys, time, ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
assert hip.hipSetDevice(0) == 0
hip.hipDeviceSynchronize()
for gb in [float(x) for x in sys.argv[1:]]:
    n = int(gb * (1 << 30))
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), n); t1 = time.perf_counter()
    hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
    print("%5.1f GB: hipMalloc %7.1f ms (rc %d), first memset %7.1f ms, second memset %6.1f ms" % (gb, 1e3 * (t1 - t0), rc, 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
