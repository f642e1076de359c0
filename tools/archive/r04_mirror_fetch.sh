#!/bin/bash
# FETCH_SIZE of the mirror pass by gather shape (variant builds): does gathering more rows together fetch fewer lines?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in base ib9v5 ib6v3 ib2v1; do
    lib=$PWD/pyiga_amd/libigx_$v.so; [ "$v" = base ] && lib=$PWD/pyiga_amd/libigx.so
    rm -rf gpurun_out/mf_$v
    IGX_LIB=$lib timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/mf_$v -- python3 bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-api-call --placement-tries 1 > /dev/null 2>&1
    python3 - "$v" <<'PY'
import csv, glob, sys
v = sys.argv[1]
for f in glob.glob('gpurun_out/mf_%s/**/*counter_collection.csv' % v, recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'k_mirror2' in r['Kernel_Name']]
    if rows:
        r = rows[-1]
        print(v, 'k_mirror2 reads', round(2 * float(r['Counter_Value']) * 1024 / 1e9, 2), 'GB', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'ms (under the profiler)')
PY
    rm -rf gpurun_out/mf_$v
done
