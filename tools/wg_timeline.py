import sys, numpy as np
d=np.loadtxt(sys.argv[1], dtype=np.int64)
t0=d[:,1].min(); st=(d[:,1]-t0)/100.0; en=(d[:,2]-t0)/100.0; hw=d[:,3]
print("n", len(d), "launch span us", en.max(), "dur mean", (en-st).mean(), "min", (en-st).min(), "max", (en-st).max())
print("start percentiles", np.percentile(st,[0,10,25,50,75,90,100]).round(1))
print("end percentiles", np.percentile(en,[0,10,25,50,75,90,100]).round(1))
first=st<10
print("first-round wgs", first.sum(), "dur", (en-st)[first].mean().round(1), "| later", (~first).sum(), "dur", (en-st)[~first].mean().round(1), "start mean", st[~first].mean().round(1))
xcc=(hw>>32)&0xf; cu=(hw>>8)&0xf; se=(hw>>13)&0x7; sh=(hw>>12)&1
key=xcc*1000+se*100+sh*20+cu
u,c=np.unique(key,return_counts=True)
print("distinct CUs", len(u), "wgs per CU: min", c.min(), "max", c.max(), "hist", np.bincount(c))
