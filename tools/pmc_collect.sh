#!/bin/bash
# HBM traffic counters of the bench command, one rocprofv3 --pmc pass per counter (guides/MI355X_MICROARCH.md: PMC passes
# on their own, never combined with the system / runtime trace domains).  Run on the GPU box from the repo root:
#   bash tools/pmc_collect.sh [extra bench.py flags]      -> gpurun_out/pmc/<counter>/..., then tools/pmc_summarise.py
set -u
cd /tmp && export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/pmc
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$ctr -- \
    python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 20 --warmup 40 --no-cpu-baseline --no-extras "$@" > $OUT.$ctr.log 2>&1
done
# occupancy / issue counters (one pass: 8 SQ slots + 2 GRBM slots): how busy the shader engines are while a kernel runs, how many waves
# it had and what they did with their cycles (SQ_WAVE_CYCLES and the SQ_WAIT_* / SQ_ACTIVE_* family count quad-cycles)
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/OCCUPANCY -- \
  python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 20 --warmup 40 --no-cpu-baseline --no-extras "$@" > $OUT.OCCUPANCY.log 2>&1
