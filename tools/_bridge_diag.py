import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'openal-soft_amd')
import bridge_lib as bl, oalgpu
from test_bridge import render
todo=(1024,1024,700,1024,1024)
def cmp(name, a, b):
    off=0
    for k,n in enumerate(todo):
        e=np.abs(a[off:off+n].astype(np.float64)-b[off:off+n]).max(); m=np.abs(b[off:off+n]).max()
        print(name,'update',k,'err %.3e max %.3e'%(e,m), 'argmax', int(np.abs(a[off:off+n]-b[off:off+n]).max(axis=1).argmax()))
        off+=n
want,sw=render(bl.MODE_CPU)
got,sg=render(bl.MODE_ADAPTERS, math_mode=oalgpu.MATH_EXACT)
cmp('adapters', got, want); print(sg==sw)
for kw in (dict(filtered=True), dict(stop=True)):
    want,sw=render(bl.MODE_CPU, **kw)
    got,sg=render(bl.MODE_BATCH, math_mode=oalgpu.MATH_FAST, **kw)
    cmp(str(kw), got, want); print(sg==sw)
