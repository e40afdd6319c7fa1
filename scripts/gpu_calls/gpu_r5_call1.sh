#!/bin/bash
# round 5, call 1: the tightened / new GPU tests, a baseline bench line of this round's box, the one-rank emulation of an 8-GPU run and the
# Winograd grid-threshold sweep on it (MP_WINO_MIN_WGS) + on the released K = 1 / K = 5 recipes.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c1
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stem_records.py tests/test_gpu_full_size.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-extras > $O/bench_n1.json 2> $O/bench_n1.err; echo "rc=$?" >> $O/bench_n1.err
for t in 256 128 64 16; do
  MP_WINO_MIN_WGS=$t timeout 200 python bench.py --emulate-rank-of 8 --config 4 --steps 2 > $O/emul8_c4_min$t.json 2> $O/emul8_c4_min$t.err
  MP_WINO_MIN_WGS=$t timeout 200 python bench.py --config 2 --k-hyp 5 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/k5_min$t.json 2> $O/k5_min$t.err
  MP_WINO_MIN_WGS=$t timeout 200 python bench.py --config 2 --k-hyp 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/k1_min$t.json 2> $O/k1_min$t.err
done
timeout 200 python bench.py --emulate-rank-of 8 --config 2 --steps 1 > $O/emul8_c2.json 2> $O/emul8_c2.err
tail -n 3 $O/pytest.log
python - <<'PY'
import json,glob,os
O="gpurun_out/r5c1"
def load(f):
    try: return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: return {"error": str(e)}
b=load(f"{O}/bench_n1.json"); print("bench", b.get("value"), b.get("ms_per_step"), (b.get("roofline") or {}).get("frac"))
for t in (256,128,64,16):
    e=load(f"{O}/emul8_c4_min{t}.json"); k5=load(f"{O}/k5_min{t}.json"); k1=load(f"{O}/k1_min{t}.json")
    print(t, "emul c4: whole", e.get("one_gpu_whole_workload_ms"), "share", e.get("one_rank_share_ms"), "x", e.get("projected_speedup"), e.get("stage_ms_rank_share"),
          "| K5 ms", k5.get("ms_per_step"), "| K1 ms", k1.get("ms_per_step"))
e=load(f"{O}/emul8_c2.json"); print("emul c2 weak:", e.get("one_gpu_whole_workload_ms"), e.get("one_rank_share_ms"), e.get("projected_speedup"))
PY
