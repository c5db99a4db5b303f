import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
import fastlanes_amd as fl
v=(np.arange(1024)%8).astype(np.uint16)
fl.BitPacking.pack(3,v)
t=time.perf_counter()
for _ in range(2000): p=fl.BitPacking.pack(3,v)
dt=(time.perf_counter()-t)/2000
print("host-tier single-block pack u16 W=3: %.1f us per call"%(dt*1e6))
t=time.perf_counter()
for _ in range(2000): u=fl.BitPacking.unpack(3,p)
dt=(time.perf_counter()-t)/2000
print("host-tier single-block unpack u16 W=3: %.1f us per call"%(dt*1e6))
