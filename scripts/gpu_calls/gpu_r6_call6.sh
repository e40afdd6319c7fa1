#!/bin/bash
# round 6, call 6: chained parity of the headline config vs the pose head's gain and the kernel family of the 3x3 convolutions
# (scripts/parity_undamped_config2.py), + the whole GPU test suite on the restructured Winograd kernel.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6c6
mkdir -p $O
for sc in 0.05 0.25 1.0; do
  for fam in 2 1 0; do
    MP_CONV_WINO=$fam timeout 600 python scripts/parity_undamped_config2.py $sc > $O/und_${sc}_$fam.log 2>&1
    grep UNDAMPED $O/und_${sc}_$fam.log | cut -c1-900 || tail -n 3 $O/und_${sc}_$fam.log
  done
done
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 5 $O/pytest_gpu.log
