import numpy as np, sys
sys.path.insert(0,'.')
import flowdenoising_amd as fdn
from flowdenoising_amd import _lib
from flowdenoising_amd.synth import make_volume
from oracle import oracle as O
def rel(a,b): return float(np.abs(a.astype(np.float64)-b).max()/np.abs(b).max())
for shape in ((6,34,36),(8,34,36),(12,34,36),(13,34,36),(16,34,36)):
    vol = make_volume(shape,seed=4,amplitude=100.0)
    k=O.get_gaussian_kernel(0.5)
    got=fdn.OF_filter(vol,[None,k,None],0,5)
    want=O.OF_filter(vol,[None,k,None],0,5)
    d=np.abs(got-want)
    print(shape, "rel",rel(got,want), "bad z rows:", sorted(set(np.argwhere(d>1e-3*np.abs(want).max())[:,0].tolist())))
