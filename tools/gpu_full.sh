#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r2_full; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py 2>$O/bench.err | tee $O/bench_default.json | cut -c1-600
