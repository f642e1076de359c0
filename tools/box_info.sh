#!/bin/bash
# which kind of box is this: partition modes, clocks, memory info, and the C4 kernel times
cd "$GRAFT_REPO_ROOT"
rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -v "^$" | head -20
rocm-smi --showclocks 2>&1 | grep -iE "mclk|sclk|fclk|socclk" | head
rocm-smi --showmeminfo vram --showbw 2>&1 | grep -v "^$" | head -12
rocminfo 2>/dev/null | grep -iE "Marketing|Compute Unit|Max Clock|Memory Properties|Size:.*KB|Cacheline|L2|L3" | head -30
cat /sys/class/drm/card*/device/current_memory_partition 2>/dev/null | head -3
cat /sys/class/drm/card*/device/current_compute_partition 2>/dev/null | head -3
cat /sys/class/drm/card*/device/mem_info_vram_vendor 2>/dev/null | head -2
cat /sys/class/drm/card*/device/vbios_version 2>/dev/null | head -2
timeout 300 python bench.py --config c4 --no-cpu-baseline --no-api-call --steps 10 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'])
"
