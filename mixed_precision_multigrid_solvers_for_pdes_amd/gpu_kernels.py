"""Kernel wrapper classes of the reference's GPU layer (gpu/cuda_kernels.py: SmoothingKernels :75-436, TransferKernels
:439-828, MixedPrecisionKernels :831-967) over libmghip's stateless operators.

Interfaces only -- same class and method names, in-place output arrays -- NOT the reference's device arithmetic, which is
inconsistent with its own CPU path (wrong-sign updates, no omega, zeroed boundary residual, renormalised
boundary restriction, no far-edge quirk).  Every method computes what the reference's CPU plugin of the same name
computes (the oracle configuration, operator -Laplacian_h), on the device, through the C ABI.  Arrays are NumPy arrays
(the reference passes CuPy arrays); there is no CPU fallback."""
import numpy as np

from . import _lib


class HIPKernels:
    """Base (reference: CUDAKernels, gpu/cuda_kernels.py:17-72): device selection; nothing is compiled at run time."""

    def __init__(self, device_id=0):
        self.device_id = device_id
        self.lib = _lib.load()
        if _lib.device_count() <= device_id:
            raise RuntimeError(f"mghip: HIP device {device_id} is not available (no CPU fallback)")

    @staticmethod
    def _same(*arrays):
        dt = arrays[0].dtype
        if dt not in (np.float32, np.float64) or any(a.dtype != dt or a.shape != arrays[0].shape for a in arrays):
            raise ValueError("arrays must share one shape and one dtype (float32 or float64)")
        return dt


class SmoothingKernels(HIPKernels):
    def jacobi_smoothing(self, u_old, u_new, rhs, hx, hy, num_iterations=1, use_shared_memory=False, relaxation_parameter=2.0 / 3.0):
        """num_iterations weighted-Jacobi sweeps of u_old; the result is written to u_new (solvers/smoothers.py:41-86)."""
        dt = self._same(u_old, u_new, rhs)
        nx, ny = u_old.shape
        out = np.empty_like(np.ascontiguousarray(u_old))
        _lib.check(self.lib.mg_op_jacobi(_lib.dtype_code(dt), nx, ny, float(hx), float(hy), float(relaxation_parameter), int(num_iterations),
                                         _lib.ptr(np.ascontiguousarray(u_old)), _lib.ptr(np.ascontiguousarray(rhs)), _lib.ptr(out)))
        u_new[...] = out

    def red_black_gauss_seidel(self, u, rhs, hx, hy, num_iterations=1, relaxation_parameter=1.0):
        """in place; red = (i + j) even first (solvers/smoothers.py:175-207)"""
        dt = self._same(u, rhs)
        nx, ny = u.shape
        out = np.empty_like(np.ascontiguousarray(u))
        _lib.check(self.lib.mg_op_rbgs(_lib.dtype_code(dt), nx, ny, float(hx), float(hy), float(relaxation_parameter), int(num_iterations),
                                       _lib.ptr(np.ascontiguousarray(u)), _lib.ptr(np.ascontiguousarray(rhs)), _lib.ptr(out)))
        u[...] = out

    def sor_smoothing(self, u, rhs, hx, hy, omega=1.15, num_iterations=1):
        self.red_black_gauss_seidel(u, rhs, hx, hy, num_iterations, relaxation_parameter=omega)


class TransferKernels(HIPKernels):
    def restriction(self, fine_grid, coarse_grid):
        """full weighting, injection on the coarse boundary (operators/transfer.py:100-124); coarse_grid is overwritten"""
        nx, ny = fine_grid.shape
        if coarse_grid.shape != ((nx - 1) // 2 + 1, (ny - 1) // 2 + 1):
            raise ValueError(f"Cannot restrict from {fine_grid.shape} to {coarse_grid.shape}")
        out = np.empty(coarse_grid.shape, dtype=coarse_grid.dtype)
        _lib.check(self.lib.mg_op_restrict_fw(_lib.dtype_code(fine_grid.dtype), _lib.dtype_code(coarse_grid.dtype), nx, ny,
                                              _lib.ptr(np.ascontiguousarray(fine_grid)), _lib.ptr(out)))
        coarse_grid[...] = out

    def prolongation(self, coarse_grid, fine_grid):
        """bilinear with the CPU path's far-edge zeros (operators/transfer.py:234-267); fine_grid is OVERWRITTEN, as by
        the reference's kernel (gpu/cuda_kernels.py:497-541)"""
        cx, cy = coarse_grid.shape
        if fine_grid.shape != (2 * (cx - 1) + 1, 2 * (cy - 1) + 1):
            raise ValueError(f"Cannot prolong from {coarse_grid.shape} to {fine_grid.shape}")
        out = np.empty(fine_grid.shape, dtype=fine_grid.dtype)
        _lib.check(self.lib.mg_op_prolong_bilinear(_lib.dtype_code(coarse_grid.dtype), _lib.dtype_code(fine_grid.dtype), cx, cy,
                                                   _lib.ptr(np.ascontiguousarray(coarse_grid)), _lib.ptr(out)))
        fine_grid[...] = out

    def compute_residual(self, u, rhs, residual, hx, hy):
        """residual = rhs - A u with A = -Laplacian_h, boundary cells r = f (operators/laplacian.py:105-124)"""
        dt = self._same(u, rhs, residual)
        nx, ny = u.shape
        out = np.empty_like(np.ascontiguousarray(u))
        _lib.check(self.lib.mg_op_residual(_lib.dtype_code(dt), nx, ny, float(hx), float(hy), -1.0, _lib.ptr(np.ascontiguousarray(u)),
                                           _lib.ptr(np.ascontiguousarray(rhs)), _lib.ptr(out)))
        residual[...] = out


class MixedPrecisionKernels(HIPKernels):
    def compute_mixed_precision_residual(self, u, rhs, hx, hy):
        """fp32 iterate and rhs in, fp64 residual out, evaluated in double (gpu/cuda_kernels.py:937-967) -- the building
        block of defect correction (mg_config.precision = MG_PREC_DEFECT)."""
        if u.dtype != np.float32 or rhs.dtype != np.float32 or u.shape != rhs.shape:
            raise ValueError("compute_mixed_precision_residual takes float32 u and rhs of one shape")
        nx, ny = u.shape
        out = np.empty((nx, ny), dtype=np.float64)
        _lib.check(self.lib.mg_op_residual_mixed(nx, ny, float(hx), float(hy), -1.0, _lib.ptr(np.ascontiguousarray(u)),
                                                 _lib.ptr(np.ascontiguousarray(rhs)), _lib.ptr(out)))
        return out
