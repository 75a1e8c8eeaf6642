#!/bin/bash
# PMC pass over a lab binary: MFMA-busy and held clock per kernel variant (counters only).   tools/micro/pmc_lab.sh <out.json> <binary> args...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; shift
rm -rf gpurun_out/pmc_lab
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_lab -- "$@" > gpurun_out/pmc_lab.log 2>&1 || exit 1
python3 tools/pmc_mfma.py gpurun_out/pmc_lab > $out
rm -rf gpurun_out/pmc_lab
