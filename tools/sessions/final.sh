cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/final_gpu_tests.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/final_gpu_tests.txt
cat gpurun_out/final_gpu_tests.txt
bash tools/sessions/profiles.sh r02
