"""Device buffers through the HIP runtime directly (no torch in the test process: the product library brings its
own HIP context)."""
import ctypes

import numpy as np


class Dev:
    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.hip.hipFree.argtypes = [ctypes.c_void_p]
        self.ptrs = []

    def alloc(self, n):
        p = ctypes.c_void_p()
        assert self.hip.hipMalloc(ctypes.byref(p), max(int(n), 16)) == 0
        self.ptrs.append(p)
        return p.value

    def upload(self, arr):
        p = self.alloc(arr.size)
        if arr.size:
            assert self.hip.hipMemcpy(p, arr.ctypes.data, arr.size, 1) == 0
        return p

    def download(self, p, n):
        out = np.empty(max(n, 1), np.uint8)
        if n:
            assert self.hip.hipMemcpy(out.ctypes.data, p, n, 2) == 0
        return out[:n]

    def free(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []
