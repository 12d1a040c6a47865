for pool in 1 0; do for c in c3e c3; do
NLS_HOST_POOL=$pool timeout 400 python bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-end-to-end | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('pool=$pool $c wall', round(d['ms_per_step'],1), 'library', s['total'], 'chol', s['cholesky'], 'download', s['download'])"
done; done
