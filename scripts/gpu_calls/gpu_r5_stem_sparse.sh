#!/bin/bash
# round 5: background-tile walk of the stem kernel (SP instances fed by the rasteriser's job flags): kernel + record + pipeline + parity tests,
# then the bench with MP_STEM_SPARSE=1 (default) / 0 alternating.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5sp
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stem_records.py tests/test_gpu_pipeline.py tests/test_gpu_parity_full_size.py tests/test_gpu_refiner_graph.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
for v in 1 0 1 0; do
  MP_STEM_SPARSE=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_sp${v}_$RANDOM.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5sp/bench_sp*.json")):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1]); k=b["kernel_ms_per_step"]
        print(f.split("/")[-1], round(b["value"],1), round(b["ms_per_step"],2), "stemQ5", k.get("conv_stem_bf16x3+maxpool<7x7,Q5>"), "stemQ2", k.get("conv_stem_bf16x3+maxpool<7x7,Q2>"), b.get("stem_background"))
    except Exception as e: print(f, "error", e)
PY
tail -n 5 $O/bench.err
