cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-phase-b-leg"
QEA_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ss -- $B > $O/prof_ss.json 2> $O/prof_ss.err || exit 1
cp $(ls -t $(find $O/prof_ss -name "*kernel_stats.csv") | head -1) $O/r04a_kernel_stats_ss.csv && rm -rf $O/prof_ss
