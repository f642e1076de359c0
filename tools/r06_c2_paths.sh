cd "$GRAFT_REPO_ROOT"
for e in "" "IGX_PATH=fused" "IGX_PATH=single"; do
 echo "== c2 $e"
 env $e timeout 300 python bench.py --config c2 --no-cpu-baseline --steps 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['path'], d['roofline']['frac'])
    else: print(l.rstrip()[-300:])
"
done
