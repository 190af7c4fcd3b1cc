for rep in 1 2 3; do
  for k in 1 0; do
    ms=$(PROBE_KEEP_CRITIC=$k python tests/probes/no_critic.py --steps 40 --warmup 5 --no-cpu-baseline 2>gpurun_out/nc_err.log | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[keep_critic=$k] pass $rep: $ms ms/step"
  done
done
