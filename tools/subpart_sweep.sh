cd $GRAFT_REPO_ROOT
for gb in 24 6.5 3.3 1.65 0.83; do
MOONRT_PATH_MAX_GB=$gb python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('max_gb $gb', 'frame', d['ms_per_step'], 'render', d['primary_ms'], 'paths', d['paths_ms'])"
done
