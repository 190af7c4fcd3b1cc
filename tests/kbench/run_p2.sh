mkdir -p gpurun_out/p2
for m in 0 1 2 4 8 16 32 64 128 255; do
  timeout -k 10 60 env BMHRL_ATTN_TRACE=1 tests/kbench/build/pair_abl_$m one 128 16 4 256 800 14 2 3 > gpurun_out/p2/abl_$m.full 2>&1; rc=$?
  tail -3 gpurun_out/p2/abl_$m.full > gpurun_out/p2/abl_$m.log; rm gpurun_out/p2/abl_$m.full
  echo "abl $m rc=$rc"; [ $rc -eq 124 ] && exit 1
done
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/p2/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/p2/pytest.log
