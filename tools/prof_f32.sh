cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_f32
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f32 -- python3 $R/bench.py --dtype ${1:-f32} --steps 2 --warmup 1 --steps-only > $R/gpurun_out/prof_f32.log 2>&1
find $R/gpurun_out/prof_f32 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $R/gpurun_out/r03_${1:-f32}_kernel_stats.csv
rm -rf $R/gpurun_out/prof_f32
head -25 $R/gpurun_out/r03_${1:-f32}_kernel_stats.csv | cut -c1-150
