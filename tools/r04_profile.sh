#!/bin/bash
# Round-4 profile pass (GPU box): kernel-trace stats in both stream modes, then the PMC passes (separately, counters only).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-phase-b-leg"
QEA_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ss -- $B > $O/prof_ss.json 2> $O/prof_ss.err || exit 1
cp $(ls -t $(find $O/prof_ss -name "*kernel_stats.csv") | head -1) $O/r04_kernel_stats_b2048_single_stream.csv && rm -rf $O/prof_ss
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ov -- $B > $O/prof_ov.json 2> $O/prof_ov.err || exit 2
cp $(ls -t $(find $O/prof_ov -name "*kernel_stats.csv") | head -1) $O/r04_kernel_stats_b2048_default_overlap.csv && rm -rf $O/prof_ov
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-phase-b-leg"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 3
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P > $O/pmc_write.json 2> $O/pmc_write.err || exit 4
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 2048 > $O/r04_pmc_traffic.json || exit 5
rm -rf $O/pmc_fetch $O/pmc_write
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_mfma -- $P > $O/pmc_mfma.json 2> $O/pmc_mfma.err || exit 6
python3 tools/pmc_mfma.py $O/pmc_mfma > $O/r04_pmc_mfma_busy.json || exit 7
rm -rf $O/pmc_mfma
echo done
