import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import chainer_faster_rcnn_amd as pkg
rt = pkg.runtime.default_runtime()
rs = np.random.RandomState(0)
for (ci,co,h,w,cfgs) in [(64,64,600,1000,(10,1010,2010)),(256,256,150,250,(210,1210,2210)),(512,512,75,125,(210,1210,2210)),(512,512,38,63,(205,1205,2205))]:
    x = rt.mem.from_numpy(rs.randn(1,ci,h,w).astype(np.float32)); W = rt.mem.from_numpy((rs.randn(co,ci,3,3)*0.05).astype(np.float32))
    wp = rt.pack_conv3x3_w(W); b = rt.mem.from_numpy(rs.randn(co).astype(np.float32))
    ys=[]
    for c in cfgs:
        y = rt.mem.empty((1,co,h,w),"f32"); rt.conv3x3(x, wp, b, relu=True, out=y, cfg=c); ys.append(rt.mem.to_numpy(y))
    print(ci,co,h,w,'bit-equal', all(np.array_equal(ys[0],yy) for yy in ys[1:]), flush=True)
