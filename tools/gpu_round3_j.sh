#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
timeout 1200 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "different_lengths or style_loop or cfg1_end" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for dt in bf16 f16x3; do timeout 600 python tools/bench_testset.py --dtype $dt 2>/dev/null | tee $O/testset_$dt.json; done
T0=$SECONDS; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench wall $((SECONDS - T0)) s"
python3 -c "
import json; d=json.load(open('$O/bench_default.json'))
print('headline', d['dtype'], d['value'], 'frames/s', d['roofline']['frac'], 'parity', d['parity']['max_abs'], '| contract', d['contract_mode']['value'], d['contract_mode']['roofline']['frac'], d['contract_mode']['parity_max_abs'], '| cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], '| tiles match', d['roofline']['counters_tiles_match'], d['roofline']['counters_from'])"
