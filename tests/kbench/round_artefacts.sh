#!/bin/bash
# round-end artefacts: the round profile (tests/prof_round.sh), the captured step's kernel sequence, trimmed to summaries
bash tests/prof_round.sh > gpurun_out/prof_round.log 2>&1; echo "prof_round rc=$?"; tail -5 gpurun_out/prof_round.log
bash tests/probes/step_sequence.sh > gpurun_out/seq.log 2>&1; echo "seq rc=$?"; head -2 gpurun_out/seq/sequence.txt
cd gpurun_out/round && rm -rf attn128 attn256 fetch write pmc128a pmc128b pmc256a pmc256b eager benchtrace; cd ../..
du -sh gpurun_out
