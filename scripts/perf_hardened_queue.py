"""The queue of instances (sipp_instances_prove) with the hardened G1 / G2 AIRs at the n = 128 size: ms per instance for two stream-level
choices, next to the plain queue.  usage: perf_hardened_queue.py [in_flight=5] [instances=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sipp_amd
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sipp_n128_ios.npz"))
ios = [d["g1"], d["g2"], d["fq12"]]
in_flight = int(sys.argv[1]) if len(sys.argv) > 1 else 5
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for hardened, prios in ((False, ("low", "", "high")), (True, ("low", "", "high")), (True, ("low", "high", "high"))):
    q = sipp_amd.InstanceQueue([a.shape[0] for a in ios], in_flight=in_flight, priorities=prios, hardened=hardened)
    try:
        q.prove([ios] * in_flight)
        t = time.perf_counter()
        q.prove([ios] * count)
        ms = (time.perf_counter() - t) / count * 1e3
    finally:
        q.close()
    print("hardened %s levels %s: %.2f ms per instance (%d in flight)" % (hardened, prios, ms, in_flight), flush=True)
