"""Device buffers for the GPU tests through the HIP runtime libvi_amd.so itself is linked against.
No torch in the test process: a second HIP runtime initialised after the library's does not see the GPU."""
import numpy as np


class Hip:
    """device buffers through the HIP runtime the library itself is linked against (no torch in this process:
    a second HIP runtime initialised after the library's does not see the GPU)"""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.bufs = []

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), max(int(nbytes), 1)) == 0
        self.bufs.append(p)
        return p.value

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        assert self.rt.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p

    def upload_to(self, p, a):
        a = np.ascontiguousarray(a)
        assert self.rt.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0
        return out

    def close(self):
        for p in self.bufs:
            self.rt.hipFree(p)
