#!/bin/bash
# which ingredient of the conv K loop costs the cycles; clock probe; bench with clock
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c13
mkdir -p $O
timeout 120 scripts/microbench/_build/conv_loop_parts > $O/loop_parts.log 2>&1
timeout 120 scripts/microbench/_build/conv_loop_parts > $O/loop_parts_2.log 2>&1
timeout 100 python -c "
from megapose6d_amd import engine as eng
import torch
torch.zeros(1, device='cuda')
for ms in (5, 30, 100): print(ms, eng.clock_probe(ms))
" > $O/clock.log 2>&1
