import numpy as np, time, sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import oracle_lib as orc, bench
print("usable", bench.usable_cores(), "machine", os.cpu_count())
n=1<<23; m=1024
x=orc.fill_uniform(2*n,1,-10,10,np.float32); h=orc.fill_uniform(2*m,2,-1,1,np.float32)/np.float32(m)
for th in (1,8,16,32,64,128,256):
    t0=time.perf_counter(); c,y=orc.overlap_save_mt(x,h,1024,th); t1=time.perf_counter(); f=orc.fft_pow2_mt(y,False,th); t2=time.perf_counter()
    print(th, "conv %.3f s  fft %.3f s"%(t1-t0,t2-t1))
