# small probes in one call: RCCL one-rank DP step + the LDS fill probe
set -x
mkdir -p gpurun_out
timeout 300 python tools/probe_rccl_world1.py --steps 8 > gpurun_out/rccl_world1.json 2> gpurun_out/rccl_world1.err; echo "rc=$?" >> gpurun_out/rccl_world1.json
tail -3 gpurun_out/rccl_world1.err; cat gpurun_out/rccl_world1.json
timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/fill_probe.hip -o /tmp/fill 2>/dev/null && timeout 60 /tmp/fill | tee gpurun_out/fill_probe.txt
