"""How long does the first blocking HIP call take after the GPU sat idle for d milliseconds?  (A full iteration has host
phases of 10-30 ms — build_evidence, the commit of a one-batch class — and the first table upload after them was
measured at ~21 ms for a 48-byte copy.)"""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), 4096) == 0
buf = (C.c_char * 4096)()
for mode in ("memcpy", "devsync", "busy-then-memcpy"):
    for d in (0, 1, 2, 5, 10, 15, 20, 30, 50, 100):
        ts = []
        for rep in range(5):
            hip.hipMemcpy(dev, buf, 64, 1)
            if mode == "busy-then-memcpy":  # CPU spinning instead of sleeping (is it the host thread that goes to sleep?)
                t_end = time.perf_counter() + d * 1e-3
                while time.perf_counter() < t_end:
                    pass
            else:
                time.sleep(d * 1e-3)
            t0 = time.perf_counter()
            if mode == "devsync":
                hip.hipDeviceSynchronize()
            else:
                hip.hipMemcpy(dev, buf, 64, 1)
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"{mode:18s} idle {d:4d} ms -> first call {min(ts):7.3f} .. {max(ts):7.3f} ms", flush=True)
