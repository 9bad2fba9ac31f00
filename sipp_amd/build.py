"""Builds sipp_amd/libsipp_hip.so (hipcc, gfx950 only) in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libsipp_hip.so")


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "sipp_hip.h"))
    srcs.append(os.path.join(os.path.dirname(HERE), "data", "air_tables.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    if force or _stale():
        jobs = str(min(8, os.cpu_count() or 1))
        out = None if verbose else subprocess.DEVNULL
        subprocess.check_call(["make", "-C", CSRC, "-j", jobs], stdout=out)
    return SO


if __name__ == "__main__":
    print(build(verbose=True))
