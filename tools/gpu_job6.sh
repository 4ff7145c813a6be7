#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_collect.sh cfg2_bf16 --dtype bf16 2>&1 | tail -12
bash tools/pmc_collect.sh cfg2_f16x3 --dtype f16x3 2>&1 | tail -12
