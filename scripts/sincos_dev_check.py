import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, ctypes as C
from conftest import load_package
fg=load_package(); L=fg.lib()
rng=np.random.default_rng(0)
x=np.concatenate([rng.uniform(-1.5707963267948966,1.5707963267948966,4000000), rng.uniform(-0.126,0.126,1000000), rng.uniform(-2.4,2.4,1000000),
                  rng.uniform(-1,1,1000000)*2.0**(-rng.integers(0,40,1000000))])
s=np.empty_like(x); c=np.empty_like(x)
dp=lambda a:a.ctypes.data_as(C.POINTER(C.c_double))
assert L.fg_sincos_batch(x.size,dp(x),dp(s),dp(c),0)==0
import orc
O=orc.oracle(); O.orc_sincos.argtypes=[C.c_long]+[C.POINTER(C.c_double)]*3; O.orc_sincos.restype=None
rs=np.empty_like(x); rc=np.empty_like(x); O.orc_sincos(x.size,dp(x),dp(rs),dp(rc))
print('numpy vs libm: sin',(np.sin(x)!=rs).sum(),'cos',(np.cos(x)!=rc).sum())
bs=(s!=rs); bc=(c!=rc)
print("sin mismatches",bs.sum(),"cos mismatches",bc.sum(),"of",x.size)
if bs.any():
    i=np.flatnonzero(bs)[:5]; print(x[i].tolist(), s[i].tolist(), rs[i].tolist())
if bc.any():
    i=np.flatnonzero(bc)[:5]; print(x[i].tolist(), c[i].tolist(), rc[i].tolist())
