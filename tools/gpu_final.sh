#!/bin/bash
# end-of-round check: the full GPU suite, smoke, the driver's default bench line (timed), the cfg1x8 line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$SECONDS; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench wall $((SECONDS - T0)) s"
python3 -c "
import json; d=json.load(open('$O/bench_default.json'))
print('headline', d['dtype'], d['value'], 'frames/s', d['roofline']['frac'], 'parity', d['parity']['max_abs'], '| contract', d['contract_mode']['value'], d['contract_mode']['roofline']['frac'], d['contract_mode']['parity_max_abs'], '| cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], '| tiles match', d['roofline']['counters_tiles_match'], d['roofline']['counters_from'])"
timeout 600 python bench.py --config cfg1x8 --headline-only --no-cpu-baseline --steps 5 > $O/cfg1x8_bf16.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/cfg1x8_bf16.json')); print('cfg1x8', d['value'], 'vs sequential', d['sequential_loop']['value'], 'x', d['speedup_vs_sequential_loop'], d['sequential_loop']['bit_identical_to_batched'])"
