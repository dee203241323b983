"""Device memory for tests straight from the HIP runtime (ctypes on libamdhip64): entry points that take device
pointers are handed buffers the library did not allocate.  Not torch: the tests' process has libsfx's HIP runtime
loaded already, and torch brings a second copy of the runtime that then finds no device."""
import ctypes as C

import numpy as np

H2D, D2H = 1, 2


class Hip:
    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.bufs = []

    def free_bytes(self):
        free, total = C.c_size_t(), C.c_size_t()
        assert self.lib.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    def malloc(self, nbytes):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        self.bufs.append(p)
        return p.value

    def to_device(self, arr):
        a = np.ascontiguousarray(arr)
        p = self.malloc(a.nbytes)
        assert self.lib.hipMemcpy(p, a.ctypes.data, a.nbytes, H2D) == 0
        return p

    def to_host(self, p, dtype, count):
        out = np.empty(count, dtype)
        assert self.lib.hipMemcpy(out.ctypes.data, p, out.nbytes, D2H) == 0   # blocking: ordered after the null stream only
        return out

    def fill(self, p, byte, nbytes):
        assert self.lib.hipMemset(p, byte, nbytes) == 0

    def free_all(self):
        for p in self.bufs:
            self.lib.hipFree(p)
        self.bufs = []
