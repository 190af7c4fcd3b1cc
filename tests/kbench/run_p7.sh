mkdir -p gpurun_out/p7
for w in 1 0; do for d in 0 3 5 4; do for sh in "4096 1024 1024 0 0 1 0" "4096 1024 1024 0 0 1 1"; do
  echo "w8=$w dbg $d: $(timeout -k 10 60 env BMHRL_GEMM_W8=$w BMHRL_GEMM_DBG=$d tests/kbench/gemm_bench one $sh 30 2>&1 | tail -1)"; done; done; done > gpurun_out/p7/dbg.log 2>&1
cat gpurun_out/p7/dbg.log
