"""Device memory plumbing.  Mirrors multigrid.gpu.memory_manager (gpu/memory_manager.py:45-542): GPUMemoryPool (shape /
dtype keyed free lists) and GPUMemoryManager (to_gpu / to_cpu, streams, statistics) with the reference's method names and
statistics keys, over torch device tensors instead of CuPy arrays.

The solver itself does not allocate through a pool: `mg_create` allocates every level of the hierarchy once (one arena per
handle, csrc/mghip.hip) and a solve allocates nothing.  These classes serve callers that stage their own device fields for
the device-pointer entry points (`mg_dev_*`, include/mghip.h): 2-D allocations are PITCHED the way those entry points
expect -- rows start on 512-byte boundaries (mg_pitch_elems) -- and handed out as the logical (nx, ny) view, so
`t.stride(0)` is the `ld` argument and `t.data_ptr()` the field pointer."""
import ctypes as C
import threading
import time

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


def _tdtype(dtype):
    torch = _torch()
    dt = np.dtype(dtype)
    if dt == np.float32:
        return torch.float32
    if dt == np.float64:
        return torch.float64
    raise TypeError(f"unsupported dtype {dt}: the multigrid path computes in float32 or float64")


class GPUMemoryBlock:                                                       # gpu/memory_manager.py:21-42
    def __init__(self, array, allocation_id):
        self.array, self.allocation_id = array, allocation_id
        self.in_use, self.last_used = True, time.time()

    @property
    def size_bytes(self):
        return self.array.untyped_storage().nbytes()

    @property
    def shape(self):
        return tuple(self.array.shape)

    @property
    def dtype(self):
        return np.dtype(str(self.array.dtype).replace("torch.", ""))


class GPUMemoryPool:
    def __init__(self, max_pool_size_mb=2048.0, device_id=0, enable_statistics=True):
        torch = _torch()
        if not torch.cuda.is_available() or torch.cuda.device_count() <= device_id:
            raise RuntimeError(f"mghip: HIP device {device_id} is not available (the device path has no CPU fallback)")
        self.lib = _lib.load()
        self.max_pool_size_bytes = int(max_pool_size_mb * 1024 * 1024)
        self.device_id = device_id
        self.device = torch.device("cuda", device_id)
        self.enable_statistics = enable_statistics
        self.pools = {}
        self.stats = {"total_allocated_bytes": 0, "total_allocations": 0, "cache_hits": 0, "cache_misses": 0,
                      "pool_evictions": 0, "allocation_id_counter": 0, "device_transfers": 0}
        self._lock = threading.RLock()

    def _new(self, shape, dtype):
        torch = _torch()
        td = _tdtype(dtype)
        if len(shape) == 2:                                                  # a field: pitched rows (mg_pitch_elems)
            ld = C.c_int(0)
            _lib.check(self.lib.mg_pitch_elems(_lib.dtype_code(dtype), int(shape[1]), C.byref(ld)))
            return torch.empty((shape[0], ld.value), dtype=td, device=self.device)[:, :shape[1]]
        return torch.empty(tuple(shape), dtype=td, device=self.device)

    def allocate(self, shape, dtype=np.float64, zero_fill=True):            # gpu/memory_manager.py:96-150
        if isinstance(shape, int):
            shape = (shape,)
        shape = tuple(int(s) for s in shape)
        key = (shape, np.dtype(dtype))
        with self._lock:
            for block in self.pools.get(key, []):
                if not block.in_use:
                    block.in_use, block.last_used = True, time.time()
                    if zero_fill:
                        block.array.zero_()
                    if self.enable_statistics:
                        self.stats["cache_hits"] += 1
                    return block.array
            arr = self._new(shape, dtype)
            nbytes = arr.untyped_storage().nbytes()
            self._ensure_gpu_space_available(nbytes)
            block = GPUMemoryBlock(arr, self.stats["allocation_id_counter"])
            self.pools.setdefault(key, []).append(block)
            if zero_fill:
                arr.zero_()
            if self.enable_statistics:
                self.stats["cache_misses"] += 1
                self.stats["total_allocations"] += 1
                self.stats["allocation_id_counter"] += 1
                self.stats["total_allocated_bytes"] += nbytes
            return arr

    def deallocate(self, array):                                             # :152-179
        if array is None:
            return
        key = (tuple(array.shape), np.dtype(str(array.dtype).replace("torch.", "")))
        with self._lock:
            for block in self.pools.get(key, []):
                if block.array.data_ptr() == array.data_ptr():
                    block.in_use, block.last_used = False, time.time()
                    return

    def _ensure_gpu_space_available(self, required_bytes):                   # :212-254: evict idle blocks, oldest first
        cur = sum(b.size_bytes for p in self.pools.values() for b in p)
        if cur + required_bytes <= self.max_pool_size_bytes:
            return
        idle = sorted((b.last_used, k, b) for k, p in self.pools.items() for b in p if not b.in_use)
        for _, k, b in idle:
            self.pools[k].remove(b)
            cur -= b.size_bytes
            self.stats["pool_evictions"] += 1
            if cur + required_bytes <= self.max_pool_size_bytes:
                break

    def get_gpu_memory_info(self):                                           # :256-269
        torch = _torch()
        free, total = torch.cuda.mem_get_info(self.device)
        return {"device_id": self.device_id, "total_gpu_memory": total, "used_gpu_memory": total - free,
                "free_gpu_memory": free, "pool_stats": self.get_statistics()}

    def get_statistics(self):                                                # :271-298
        with self._lock:
            cur = sum(b.size_bytes for p in self.pools.values() for b in p)
            blocks = sum(len(p) for p in self.pools.values())
            used = sum(1 for p in self.pools.values() for b in p if b.in_use)
            hits, misses = self.stats["cache_hits"], self.stats["cache_misses"]
            return {**self.stats, "current_size_bytes": cur, "current_size_mb": cur / (1024 * 1024), "total_blocks": blocks,
                    "in_use_blocks": used, "available_blocks": blocks - used, "hit_rate": hits / max(1, hits + misses),
                    "pool_utilization": cur / max(1, self.max_pool_size_bytes), "num_pool_types": len(self.pools)}

    def clear(self):                                                         # :300-314
        with self._lock:
            self.pools.clear()
            if self.enable_statistics:
                self.stats["total_allocated_bytes"] = 0
                self.stats["pool_evictions"] += 1
        _torch().cuda.empty_cache()


class GPUMemoryManager:
    def __init__(self, device_id=0, max_pool_size_mb=2048.0, enable_pinned_memory=True, num_streams=4):
        torch = _torch()
        self.device_id = device_id
        self.enable_pinned_memory = enable_pinned_memory
        self.memory_pool = GPUMemoryPool(max_pool_size_mb, device_id)
        self.device = self.memory_pool.device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(num_streams)]
        self.current_stream_idx = 0
        self.pinned_arrays = []

    def allocate_gpu_array(self, shape, dtype=np.float64, zero_fill=True):   # gpu/memory_manager.py:362-369
        return self.memory_pool.allocate(shape, dtype, zero_fill)

    def allocate_like_gpu(self, reference, zero_fill=True):
        dt = reference.dtype if isinstance(reference, np.ndarray) else np.dtype(str(reference.dtype).replace("torch.", ""))
        return self.allocate_gpu_array(tuple(reference.shape), dt, zero_fill)

    def _stream(self, stream_idx):
        return self.streams[stream_idx] if stream_idx is not None and 0 <= stream_idx < len(self.streams) else None

    def to_gpu(self, cpu_array, stream_idx=None):                            # :375-396 (into a pitched pool block)
        torch = _torch()
        a = _lib.as_c(cpu_array)
        out = self.memory_pool.allocate(a.shape, a.dtype, zero_fill=False)
        st = self._stream(stream_idx)
        src = torch.from_numpy(a)
        if st is not None:
            with torch.cuda.stream(st):
                out.copy_(src, non_blocking=True)
        else:
            out.copy_(src)
        self.memory_pool.stats["device_transfers"] += 1
        return out

    def to_cpu(self, gpu_array, stream_idx=None):                            # :398-419
        torch = _torch()
        st = self._stream(stream_idx)
        if st is not None:
            with torch.cuda.stream(st):
                host = gpu_array.cpu()
            st.synchronize()
        else:
            host = gpu_array.cpu()
        self.memory_pool.stats["device_transfers"] += 1
        return host.numpy()

    def allocate_pinned_memory(self, shape, dtype=np.float64):               # :421-453
        torch = _torch()
        t = torch.empty(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=_tdtype(dtype),
                        pin_memory=self.enable_pinned_memory)
        self.pinned_arrays.append(t)
        return t.numpy()

    def get_next_stream(self):
        s = self.streams[self.current_stream_idx]
        self.current_stream_idx = (self.current_stream_idx + 1) % len(self.streams)
        return s

    def synchronize_all_streams(self):
        for s in self.streams:
            s.synchronize()

    def get_memory_usage(self):                                              # :467-477
        return {"gpu_memory_info": self.memory_pool.get_gpu_memory_info(), "pinned_memory_arrays": len(self.pinned_arrays),
                "cuda_streams": len(self.streams), "device_id": self.device_id,
                "enable_pinned_memory": self.enable_pinned_memory}

    def cleanup(self):
        self.synchronize_all_streams()
        self.memory_pool.clear()
        self.pinned_arrays.clear()

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


def check_gpu_availability():                                                # gpu/memory_manager.py:506-542
    n = _lib.device_count()
    info = {"cupy_available": False, "hip_available": n > 0, "gpu_count": n, "gpu_info": []}
    if n > 0:
        torch = _torch()
        for d in range(torch.cuda.device_count()):
            p = torch.cuda.get_device_properties(d)
            info["gpu_info"].append({"device_id": d, "name": p.name, "total_memory": p.total_memory,
                                     "multiprocessor_count": p.multi_processor_count})
    return info
