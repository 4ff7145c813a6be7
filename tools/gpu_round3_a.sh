#!/bin/bash
# round 3, first GPU pass: full GPU test suite on the new plan-layer code, the default bench line, the cfg1x8 line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cat $O/bench_default.json; tail -3 $O/bench_default.err
for dt in bf16 f16x3 f32; do
  timeout 600 python bench.py --config cfg1x8 --dtype $dt --headline-only --no-cpu-baseline --steps 5 > $O/cfg1x8_$dt.json 2> $O/cfg1x8_$dt.err; echo "cfg1x8 $dt rc=$?"; cat $O/cfg1x8_$dt.json; tail -3 $O/cfg1x8_$dt.err
done
