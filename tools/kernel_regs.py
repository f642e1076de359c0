#!/usr/bin/env python3
"""Register / scratch / LDS figures of the kernels in a device assembly file (hipcc --cuda-device-only -S)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for blk in s.split('  - .agpr_count:')[1:]:
    def g(k):
        m = re.search(r'\.' + k + r':\s+(\S+)', blk)
        return m.group(1) if m else '?'
    name = g('name')
    if pat and pat not in name:
        continue
    print('%-90s vgpr %s agpr %s sgpr %s spill v/s %s/%s scratch %s' % (name[:90], g('vgpr_count'), blk.split('\n')[0].strip(), g('sgpr_count'),
          g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size')))
