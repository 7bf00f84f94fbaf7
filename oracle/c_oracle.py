"""ctypes view of oracle/libmgoracle.so (the C restatement of the oracle; test / baseline infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "libmgoracle.so")


def load():
    if not os.path.exists(PATH) or os.path.getmtime(PATH) < os.path.getmtime(os.path.join(HERE, "mg_oracle.c")):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    lib = C.CDLL(PATH)
    lib.mgo_create.restype = C.c_void_p
    lib.mgo_create.argtypes = [C.c_int, C.c_int] + [C.c_double] * 5 + [C.c_int] * 5 + [C.c_double, C.c_double, C.c_int]
    lib.mgo_u.restype = C.POINTER(C.c_double); lib.mgo_u.argtypes = [C.c_void_p, C.c_int]
    lib.mgo_f.restype = C.POINTER(C.c_double); lib.mgo_f.argtypes = [C.c_void_p, C.c_int]
    lib.mgo_cycle.argtypes = [C.c_void_p, C.c_int]
    lib.mgo_destroy.argtypes = [C.c_void_p]
    lib.mgo_residual_norm.restype = C.c_double; lib.mgo_residual_norm.argtypes = [C.c_void_p]
    lib.mgo_levels.argtypes = [C.c_void_p]
    return lib


class COracle:
    def __init__(self, nx, ny, domain=(0.0, 1.0, 0.0, 1.0), coeff=-1.0, max_levels=4, cycle="V", pre=2, post=2,
                 smoother="jacobi", omega=0.8, coarse_tol=1e-12, coarse_maxit=1000, threads=None):
        self.lib = load()
        if threads:
            self.lib.mgo_set_threads(int(threads))
        self.nx, self.ny = nx, ny
        self.h = self.lib.mgo_create(nx, ny, *map(float, domain), float(coeff), max_levels, {"V": 0, "W": 1, "F": 2}[cycle],
                                     pre, post, {"jacobi": 0, "rbgs": 1}[smoother], float(omega), float(coarse_tol), coarse_maxit)
        self.threads = self.lib.mgo_threads()

    def _view(self, p):
        return np.ctypeslib.as_array(p, shape=(self.nx, self.ny))

    def set_problem(self, rhs, u0=None):
        self._view(self.lib.mgo_f(self.h, 0))[...] = rhs
        self._view(self.lib.mgo_u(self.h, 0))[...] = 0.0 if u0 is None else u0

    def cycle(self):
        self.lib.mgo_cycle(self.h, 0)

    def solution(self):
        return self._view(self.lib.mgo_u(self.h, 0)).copy()

    def residual_norm(self):
        return self.lib.mgo_residual_norm(self.h)

    def close(self):
        if self.h:
            self.lib.mgo_destroy(self.h)
            self.h = None
