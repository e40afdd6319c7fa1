#!/bin/bash
# round 4, call 4: the config-5 end-to-end test + the tightened logit / box bounds on every pipeline-level parity test
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c4
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_icp.py tests/test_gpu_pipeline.py tests/test_gpu_parity_full_size.py tests/test_gpu_edge_cases.py tests/test_gpu_zzzz_oracle_heavy.py -q -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -E "passed|failed|Error|assert|^\(|rc=" $O/pytest.log | tail -40
