"""Builds libmghip.so (HIP, gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libmghip.so")
SOURCES = [os.path.join(CSRC, "mghip.hip"), os.path.join(CSRC, "mg_plan.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "mg_kernels.hpp"), os.path.join(CSRC, "mg_rb_kernels.hpp"),
                  os.path.join(os.path.dirname(HERE), "include", "mghip.h")]
# -ffp-contract=off: the kernels reproduce the reference's rounding sequence (no FMA contraction).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip -> lib/libmghip.so.  Returns the library path.

    Several ranks may import the package at once (torch.distributed.run): the build is serialised with a file
    lock and the library is written to a temporary name and renamed into place, so nobody maps a half-written file."""
    if not force and not is_stale():
        return LIBPATH
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():          # another process built it while we waited
                return LIBPATH
            tmp = f"{LIBPATH}.{os.getpid()}.tmp"
            cmd = [hipcc_path()] + FLAGS + ["-o", tmp] + SOURCES
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
            os.replace(tmp, LIBPATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIBPATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
