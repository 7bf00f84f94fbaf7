"""Builds libmghip.so (HIP, gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

Every csrc/*.hip translation unit is compiled to an object of its own (in parallel, and only when it or a header is newer
than its object), then linked: a kernel edit rebuilds one unit, not the library."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIBPATH = os.path.join(LIBDIR, "libmghip.so")
SOURCES = [os.path.join(CSRC, n) for n in ("mghip.hip", "mg_plan.hip", "mg_tail.hip")]
HEADERS = [os.path.join(CSRC, n) for n in ("mg_kernels.hpp", "mg_rb_kernels.hpp", "mg_tail_kernels.hpp", "mg_host.hpp")] + \
          [os.path.join(os.path.dirname(HERE), "include", "mghip.h")]
DEPS = SOURCES + HEADERS
# headers a unit does NOT include (directly or through another header): editing them leaves its object current
NOT_INCLUDED = {"mghip.hip": ("mg_tail_kernels.hpp",),
                "mg_plan.hip": ("mg_kernels.hpp", "mg_rb_kernels.hpp", "mg_tail_kernels.hpp", "mg_host.hpp")}
# -ffp-contract=off: the kernels reproduce the reference's rounding sequence (no FMA contraction).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def extra_flags():
    """MGHIP_EXTRA_FLAGS: extra compiler flags (measurement builds: -DMG_EXPERIMENTS turns the experiment switches of the
    kernels on; such a build goes to its own file through MGHIP_LIBRARY_OUT and is never the shipped library)."""
    return os.environ.get("MGHIP_EXTRA_FLAGS", "").split()


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def source_hash():
    """sha1 over the kernel sources (csrc/*.hip, *.hpp and the C header): which build a stored measurement belongs to"""
    import hashlib
    h = hashlib.sha1()
    for path in sorted(DEPS):
        if os.path.exists(path):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:12]


def is_stale():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def _obj_of(src, tag=""):
    return os.path.join(OBJDIR, os.path.splitext(os.path.basename(src))[0] + tag + ".o")


def _obj_stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    skip = NOT_INCLUDED.get(os.path.basename(src), ())
    deps = [src] + [hd for hd in HEADERS if os.path.basename(hd) not in skip]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False, out=None):
    """Compile csrc/*.hip -> lib/libmghip.so.  Returns the library path.

    Several ranks may import the package at once (torch.distributed.run): the build is serialised with a file
    lock and the library is written to a temporary name and renamed into place, so nobody maps a half-written file."""
    out = out or os.environ.get("MGHIP_LIBRARY_OUT") or LIBPATH
    extra = extra_flags()
    if out == LIBPATH and extra:
        raise RuntimeError("MGHIP_EXTRA_FLAGS builds are measurement builds: set MGHIP_LIBRARY_OUT to a file of their own")
    if not force and out == LIBPATH and not is_stale():
        return LIBPATH
    import fcntl
    os.makedirs(OBJDIR, exist_ok=True)
    tag = "" if out == LIBPATH else "." + os.path.splitext(os.path.basename(out))[0]
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and out == LIBPATH and not is_stale():          # another process built it while we waited
                return LIBPATH
            cc = hipcc_path()
            todo = [(s, _obj_of(s, tag)) for s in SOURCES if os.path.exists(s)]

            def compile_one(pair):
                src, obj = pair
                if not force and not _obj_stale(src, obj):
                    return None
                tmp = f"{obj}.{os.getpid()}.tmp"
                cmd = [cc] + FLAGS + extra + ["-c", "-o", tmp, src]
                if verbose:
                    print(" ".join(cmd))
                res = subprocess.run(cmd, capture_output=True, text=True)
                if res.returncode != 0:
                    if os.path.exists(tmp):
                        os.remove(tmp)
                    return f"{os.path.basename(src)}:\n{res.stdout}{res.stderr}"
                os.replace(tmp, obj)
                return None
            with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) - 1))) as pool:
                errors = [e for e in pool.map(compile_one, todo) if e]
            if errors:
                raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
            tmp = f"{out}.{os.getpid()}.tmp"
            cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [o for _, o in todo]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc (link) failed:\n" + res.stdout + res.stderr)
            os.replace(tmp, out)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
