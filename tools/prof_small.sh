cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for b in 32 8; do
QEA_OVERLAP=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sb -- python3 bench.py --batch $b --phase-b-only --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/prof_sb$b.json 2> $O/prof_sb.err || exit 1
cp $(ls -t $(find $O/prof_sb -name "*kernel_stats.csv") | head -1) $O/r04_small_b${b}_kernel_stats.csv && rm -rf $O/prof_sb
done
