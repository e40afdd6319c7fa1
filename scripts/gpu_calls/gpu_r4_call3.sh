#!/bin/bash
# round 4, call 3: bf16x9 Winograd in the backbone: kernel tests, pipeline / full-size parity tests, quick bench
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_parity_full_size.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c3/bench.json"))
print(d["value"], d["ms_per_step"]); print(d["kernel_ms_per_step"]); r=d["roofline"]; print({k:r[k] for k in ("kernel","achieved","peak","frac","shader_clock_mhz","frac_at_measured_clock","k_loop_cycles_per_16_channel_step")}); print(r["per_kernel"]); print(r["clock"])
PY
tail -n 3 $O/bench.err
