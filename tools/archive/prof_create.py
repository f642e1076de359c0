import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import pyiga_amd as iga
p, n, dim = 4, 128, 3
geo = iga.geometry.tensor_product(iga.geometry.line_segment(0.0, 1.0), iga.geometry.quarter_annulus())
kv = iga.bspline.make_knots(p, 0., 1., n)
kvs = (kv,) * dim
patch = iga.assemblers.DevicePatch(kvs, geo); patch.close()
pr = cProfile.Profile()
pr.enable()
patch = iga.assemblers.DevicePatch(kvs, geo)
patch.ctx.sync()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
