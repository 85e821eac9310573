#!/bin/bash
# the parity cases added in round 5, verbose so that the measured errors land in the log
set -u
out=gpurun_out/${1:-r5odds}
mkdir -p "$out"
( time timeout 1500 python -m pytest -x -q -s -m gpu tests/test_gpu_error_bound.py::test_line_accumulators_are_as_close_to_the_exact_mix_as_the_reference \
  "tests/test_gpu_baseline_configs.py::test_config4_parity_after_updates_1_2_8_50" \
  "tests/test_gpu_baseline_configs.py::test_config5_parity_after_updates_1_2_8_50_on_the_default_data_set" \
  "tests/test_gpu_baseline_configs.py::test_config3_block_driven_contexts_against_the_reference" ) > "$out/pytest_odds.log" 2>&1
grep -E "passed|failed|Error|error|assert|update (0|49):" "$out/pytest_odds.log" | tail -40
