mkdir -p gpurun_out/p11
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/p11/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/p11/pytest.log
for i in 1 2; do python bench.py --no-cpu-baseline > gpurun_out/p11/bench$i.log 2>&1; done
python bench.py --no-cpu-baseline --eager --steps 6 > gpurun_out/p11/bench_eager.log 2>&1
for f in bench1 bench2 bench_eager; do python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/p11/$f.log').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])
except Exception as e: print('$f', 'failed', e)
PY
done
