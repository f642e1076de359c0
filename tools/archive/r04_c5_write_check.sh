#!/bin/bash
# WRITE_SIZE of the K1 producer at C5: k_geoA (non-symmetric) against k_stageA (IGX_GEOA=0) -- both write the same 8 arrays
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_c5w; mkdir -p $OUT
IGX_GEOA=0 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/stageA -- python3 bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline --no-api-call --placement-tries 1 > $OUT/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/geoA -- python3 bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline --no-api-call --placement-tries 1 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob
for d in ('stageA', 'geoA'):
    for f in glob.glob('gpurun_out/r04_c5w/%s/**/*counter_collection.csv' % d, recursive=True):
        for row in csv.DictReader(open(f)):
            n = row['Kernel_Name']
            if 'k_geoA' in n or 'k_stageA' in n or 'k_geo_fields' in n:
                print(d, n[:44], row['Counter_Name'], round(float(row['Counter_Value']) * 1024 / 1e9, 3), 'GB')
PY
rm -rf $OUT/stageA $OUT/geoA
