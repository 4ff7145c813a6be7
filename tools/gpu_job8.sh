#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_job8; mkdir -p $O
timeout 900 python tools/gemm_insitu.py bf16 800 > $O/insitu_bf16_800.txt 2>&1; cat $O/insitu_bf16_800.txt
timeout 900 python tools/gemm_insitu.py bf16 1992 > $O/insitu_bf16_1992.txt 2>&1; cat $O/insitu_bf16_1992.txt
timeout 900 python tools/gemm_insitu.py f16x3 800 0 1 6 8 9 7 3 > $O/insitu_f16x3_800.txt 2>&1; cat $O/insitu_f16x3_800.txt
