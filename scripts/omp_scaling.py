import os, sys, time, subprocess
import numpy as np
code = '''
import os,time,numpy as np,sys
sys.path.insert(0,".")
from tests import _oracle
d=np.load("tests/golden/sipp_n128_ios.npz")
t=time.time(); pf=_oracle.stark_prove(0,d["g1"][:32]); print(os.environ.get("OMP_NUM_THREADS"), "threads: G1 32 IO %.2f s"%(time.time()-t))
'''
for th in (16,32,64,128,256):
    env=dict(os.environ, OMP_NUM_THREADS=str(th), OMP_PROC_BIND="close" if th<=128 else "false")
    subprocess.run([sys.executable,"-c",code],env=env)
