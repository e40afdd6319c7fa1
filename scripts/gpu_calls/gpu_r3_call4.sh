#!/bin/bash
# Round 3, GPU call 4: Winograd in the backbone -- kernel tests vs torch, backbone / pipeline parity, quick bench
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c4
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "winograd or backbone" > $O/pytest_wino.log 2>&1; echo "rc=$?" >> $O/pytest_wino.log
timeout 400 python -m pytest tests/test_gpu_parity_full_size.py tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
timeout 200 python bench.py --steps 3 --warmup 1 --no-extras > $O/bench_quick.json 2> $O/bench_quick.err; echo "rc=$?" >> $O/bench_quick.err
tail -n 8 $O/pytest_wino.log $O/pytest_parity.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c4/bench_quick.json'))
print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'])[:1800]); print(d['kernel_ms_per_step']); print(d['parity'])
PY
tail -n 5 $O/bench_quick.err
