set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "groupnorm or gn" > gpurun_out/t_gn.log 2>&1; tail -2 gpurun_out/t_gn.log
python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "unet_forward_512px" > gpurun_out/t_gn2.log 2>&1; tail -2 gpurun_out/t_gn2.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rg_stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/rg_bench.out 2> gpurun_out/rg_bench.err
ST=$(find gpurun_out/rg_stats -name "*kernel_stats.csv" | head -1); cp $ST gpurun_out/rg_kernel_stats.csv; rm -rf gpurun_out/rg_stats
grep -E "splitk_reduce|gn_|attn_kernel<40" gpurun_out/rg_kernel_stats.csv | cut -c1-60,200-400
