import numpy as np, sys
sys.path.insert(0,'.')
import flowdenoising_amd as fdn
from flowdenoising_amd.synth import make_volume
from oracle import oracle as O
shape=(40,1024,1024)
vol = make_volume(shape,seed=1237,amplitude=100.0)
k=O.get_gaussian_kernel(2.0)
mean=vol.mean()
got=fdn.OF_filter_along_Z(vol,k,0,5,mean)
for t in (20,):
    lo,hi=max(0,t-8),min(40,t+9)
    want=O.filter_axis_range(vol[lo:hi],0,k,0,5,mean,t-lo,t-lo+1,nthreads=16)[t-lo]
    d=np.abs(got[t]-want)
    print(t,"max abs",d.max(),"max|want|",np.abs(want).max())
    bad=np.argwhere(d>1e-5)
    print(" n bad",len(bad)," ys:",sorted(set(bad[:,0].tolist()))[:40]," xs:",sorted(set(bad[:,1].tolist()))[:60])
    # partial chains: only back side, only 1 step etc.
for K in (3,5):
    kk=np.zeros(K); kk[:]=1.0/K
    got2=fdn.OF_filter_along_Z(vol[12:29],kk,0,5,mean)
    want2=O.filter_axis_range(vol[12:29],0,kk,0,5,mean,8,9,nthreads=16)[8]
    d=np.abs(got2[8]-want2); bad=np.argwhere(d>1e-5)
    print("K",K,"max abs",d.max()," n bad",len(bad)," ys:",sorted(set(bad[:,0].tolist()))[:40]," xs:",sorted(set(bad[:,1].tolist()))[:60])
