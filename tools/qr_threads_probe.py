import numpy as np, time
from threadpoolctl import threadpool_limits, threadpool_info
print([ (i['internal_api'], i['num_threads']) for i in threadpool_info()])
rs=np.random.RandomState(42); A=rs.randn(896,896)
np.linalg.qr(A)
t=time.time(); Q0,_=np.linalg.qr(A); print('default', time.time()-t)
for k in (1,4,8,16):
    with threadpool_limits(limits=k):
        t=time.time(); Q,_=np.linalg.qr(A); el=time.time()-t
    print(k, el, np.array_equal(Q,Q0), np.abs(Q-Q0).max())
