"""Does the final-stage time depend on where the buffers land?  Re-creates the C4 patch several times in ONE
process with dummy device allocations of different sizes in between; prints stage times and buffer addresses."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyiga_amd import bspline, geometry, assemblers, _lib
hip = C.CDLL('libamdhip64.so')
kv = bspline.make_knots(4, 0.0, 1.0, 128)
geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
held = []
for trial, pad_mb in enumerate([0, 0, 3, 17, 64, 257, 1000, 0]):
    if pad_mb:
        p = C.c_void_p()
        hip.hipMalloc(C.byref(p), C.c_size_t(pad_mb * (1 << 20) + 4096 * trial))
        held.append(p)
    patch = assemblers.DevicePatch((kv, kv, kv), geo)
    for _ in range(3):
        patch.assemble('stiffness', to_host=False)
    t = patch.timing()
    lib = _lib.load()
    addr = lib.igx_d_csr_data(patch.handle)
    print('pad %4d MB  data@%#x  total %.2f  A %.2f  B %.2f  final %.2f' % (pad_mb, addr or 0, t['total_ms'], t['stage0_ms'], t['stage1_ms'], t['final_ms']), flush=True)
    patch.close()
