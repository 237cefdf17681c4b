import numpy as np, sys
sys.path.insert(0,'/root/repo')
import ndt_feature_graph_amd as N, oracle as O
from ndt_feature_graph_amd import synth
pr = synth.pair_2d([11, 12], 10000)
pts = np.concatenate([pr["fixed"].numpy(), pr["moving"].numpy()])
ms = N.MapSet(1.0,[0,0,0],[100,100,1], n_maps=4); ms.build(pts, range_limit=30.0)
for m in range(4):
    gm,gc,gi,gn = ms.export_cells(m)
    o = O.OracleMap(1.0,[0,0,0],[100,100,1]); o.load_points(pts[m],30.0); o.compute_cells()
    cm,cc,ci,cn = o.export_cells()
    print(m, len(gn), len(cn), gn.sum(), cn.sum(), ms.counters(m))
    gs = {tuple(i):k for k,i in enumerate(gi)}; cs = {tuple(i):k for k,i in enumerate(ci)}
    only_g = [k for k in gs if k not in cs]; only_c=[k for k in cs if k not in gs]
    print("  only gpu", only_g[:5], "only cpu", only_c[:5])
    bad=0
    for k in cs:
        if k in gs:
            a,b=gs[k],cs[k]
            if gn[a]!=cn[b] or abs(gm[a]-cm[b]).max()>1e-9 or abs(gc[a]-cc[b]).max()>1e-8*abs(cc[b]).max():
                bad+=1
                if bad<4: print("  ",k, gn[a], cn[b], gm[a]-cm[b], abs(gc[a]-cc[b]).max()/abs(cc[b]).max())
    print("  bad", bad)
m=1
gm,gc,gi,gn = ms.export_cells(m)
p = pts[m].astype(np.float64)
ok = ~np.isnan(p).any(axis=1)
ok &= ~(np.sqrt((p**2).sum(axis=1)) > 30.0)
idx = np.floor(p/1.0+0.5).astype(int)+np.array([50,50,0])
for k,(i,n) in enumerate(zip(gi,gn)):
    sel = ok & (idx==i).all(axis=1)
    if sel.sum()!=n:
        print("cell", i, "gpu n", n, "numpy n", sel.sum(), "point ids", np.nonzero((idx==i).all(axis=1))[0][:10], "ranges", np.sqrt((p[(idx==i).all(axis=1)]**2).sum(axis=1))[:6])
