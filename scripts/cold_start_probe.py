import time, sys, os
t0=time.perf_counter()
import numpy as np
t1=time.perf_counter()
sys.path.insert(0, os.getcwd())
from cellmapper_amd import _native
t2=time.perf_counter()
h=_native.Handle(0)
t3=time.perf_counter()
rng=np.random.default_rng(0)
ref=rng.standard_normal((100000,50),dtype=np.float32); qry=rng.standard_normal((100000,50),dtype=np.float32)
t4=time.perf_counter()
d,i=h.knn(ref,qry,30)
t5=time.perf_counter()
d,i=h.knn(ref,qry,30)
t6=time.perf_counter()
print(dict(numpy=t1-t0, import_native=t2-t1, handle=t3-t2, first_knn=t5-t4, second_knn=t6-t5))
