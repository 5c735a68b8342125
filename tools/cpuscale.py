import os, time, threading, ctypes, zlib, sys
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
try: print("affinity:", len(os.sched_getaffinity(0)))
except Exception as e: print(e)
print("cpuset:", open("/sys/fs/cgroup/cpuset.cpus.effective").read().strip() if os.path.exists("/sys/fs/cgroup/cpuset.cpus.effective") else "n/a")
import numpy as np
data = zlib.compress(bytes(60000), 1)
def work(n, out, k):
    t=time.time(); c=0
    while time.time()-t < 1.0:
        for _ in range(50): zlib.decompress(data)
        c+=50
    out[k]=c
for n in (1,8,16,32,64,128,256):
    out=[0]*n; th=[threading.Thread(target=work,args=(n,out,k)) for k in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    print(n, "threads:", sum(out)/1e3, "k inflates/s", flush=True)
