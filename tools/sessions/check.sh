set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/final_gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/final_gpu_tests.txt
timeout 200 python bench.py --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cat gpurun_out/final_gpu_tests.txt; cut -c1-400 gpurun_out/final_bench.json
