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


def source_hash():
    """sha256 over the kernel sources the library is built from (csrc/*.hip, *.hpp, *.h, *.inc, *.cpp + data/air_tables.h), in
    name order: recorded next to the PMC counters under profiles/ (scripts/profile_round.sh) so that bench.py can tell counters of
    another code state from current ones"""
    import hashlib
    h = hashlib.sha256()
    # (verify.cpp is the host-only verifier: no kernel, no part of any profiled command's device or host time -- not in the hash)
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp", ".h", ".inc")) and f != "verify.cpp")
    for f in names:
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(b"air_tables.h\0")
    h.update(open(os.path.join(os.path.dirname(HERE), "data", "air_tables.h"), "rb").read())
    return h.hexdigest()


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
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "hash":
        print(source_hash())
    else:
        print(build(verbose=True))
