"""TEST INFRASTRUCTURE: builds tests/emu/_build/libspiral_emu.so -- the sources of sdk_amd/csrc, unchanged, compiled for the HOST
against tests/emu/hip/hip_runtime.h (workgroups as fibers, device memory = host memory).  Same C ABI as libspiral_hip.so.

Used by tests/test_emulated_library.py to run parity tests of the kernels' source against the oracle without a GPU.  It is not
a fallback: nothing in sdk_amd/ builds or loads it, bench.py and smoke() refuse it, and it is several thousand times slower
than one CU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "sdk_amd", "csrc")
BUILD = os.environ.get("SPIRAL_EMU_BUILD") or os.path.join(HERE, "_build")   # (another directory: a second tree state side by side)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
PRODUCT = ["params.cpp", "ntt.hip", "fold.hip", "elementwise.hip", "sweep.hip", "sweep_planar.hip", "db.hip", "sparse.hip", "server.cpp", "capi.cpp",
           "comm.cpp", "endpoint.cpp"]
OWN = ["emu_runtime.cpp", "emu_streams.cpp", "emu_library.cpp", "emu_rccl.cpp"]
FLAGS = ["-std=c++17", "-O2", "-fPIC", "-pthread", "-I" + HERE, "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]




def _asan_runtime():
    try:
        out = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
        return out if os.path.isabs(out) and os.path.exists(out) else ""
    except OSError:
        return ""


ASAN_RUNTIME = _asan_runtime()   # "" when this compiler has no shared AddressSanitizer runtime


def library_path(asan=False):
    return os.path.join(BUILD, "libspiral_emu_asan.so" if asan else "libspiral_emu.so")


def build(force=False, asan=False):
    """asan=True: the same with AddressSanitizer -- every "device" buffer is a heap block with red zones, so a kernel that reads
    or writes one element out of bounds is reported (run python with LD_PRELOAD=ASAN_RUNTIME, ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0).  Heap and
    globals only: the work-items' stacks are fibers ASan does not know about, so stack variables are left uninstrumented."""
    if not os.path.exists(CLANG):
        return None
    flags = FLAGS + (["-fsanitize=address", "-shared-libasan", "-mllvm", "-asan-stack=0", "-g", "-fno-omit-frame-pointer"] if asan else [])
    sub = "obj_asan" if asan else "obj"
    os.makedirs(os.path.join(BUILD, sub), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers += [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "rccl", "rccl.h"),
                os.path.join(ROOT, "include", "spiral_hip.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    objs, jobs = [], []
    for src_dir, names in ((CSRC, PRODUCT), (HERE, OWN)):
        for name in names:
            src = os.path.join(src_dir, name)
            obj = os.path.join(BUILD, sub, name + ".o")
            objs.append(obj)
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
                jobs.append((name, subprocess.Popen([CLANG, "-x", "c++"] + flags + ["-c", src, "-o", obj],
                                                    stderr=subprocess.PIPE, text=True)))
    for name, p in jobs:
        err = p.communicate()[1]
        if p.returncode != 0:
            raise RuntimeError(f"emulated build of {name} failed:\n{err[-4000:]}")
    so = library_path(asan)
    if jobs or not os.path.exists(so):
        subprocess.run([CLANG, "-shared", "-pthread"] + (["-fsanitize=address", "-shared-libasan"] if asan else []) + ["-o", so] + objs +
                       ["-lrt"], check=True)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, asan="--asan" in sys.argv))
