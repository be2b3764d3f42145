#!/bin/bash
# Where do the cycles of the gather kernels go?  Separate rocprofv3 --pmc passes over one step of
# tools/tune_sweep.py -- AT MOST TWO counters of one hardware block per pass (more exceeds the
# block's counter slots: rocprofv3 aborts with error 38 and the aborted process hangs), each pass
# under its own timeout; summaries via tools/pmc_summary.py.
#   bash tools/pmc_diag.sh <outdir> [tune_sweep args...]
OUT=${1:-gpurun_out/pmc_diag}; shift || true
mkdir -p "$OUT"
HERE=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$HERE"
ARGS=("$@")
names=()
run() {
  name=$1; shift
  timeout -k 5 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- python tools/tune_sweep.py --steps 1 "${ARGS[@]}" > "$OUT/$name.log" 2>&1
  rc=$?
  echo "[diag] $name rc=$rc"
  [ $rc -eq 0 ] && names+=("$name")
  [ $rc -ne 0 ] && [ $rc -ne 124 ] && [ $rc -ne 137 ] && return 0
  return 0
}
run ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
run tcc1 TCC_TAG_STALL_sum TCC_BUSY_sum
run tcc2 TCC_REQ_sum TCC_EA0_RDREQ_sum
run sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run sq2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq4 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
run sq5 SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU
run grbm GRBM_GUI_ACTIVE
D=$(dirname "$(find "$OUT" -name '*_counter_collection.csv' | head -1)")
python tools/pmc_summary.py "$D" "${names[@]}" > "$OUT/summary.json"
echo "[diag] summary written"
