#!/bin/bash
# round 4, session au: how sensitive are the dh = 64 attention kernels to workgroups per CU?  (LDS padding lowers it:
# occ_lo = forward 3 (default 4), dQ 2 (3), dK/dV 1 (2);  occ_lo2 = forward 2, dQ 1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=$PWD/transformertts_amd/lib
V=""
for n in occ_lo occ_lo2; do V="$V TTSMI_ALLOW_LIB_OVERRIDE=1,TTSMI_LIB=$L/libttsmi_$n.so"; done
timeout 600 python tools/kbench.py --only attn --variants base $V 2>&1 | grep -E "^attn|variant" | grep -v 192 > gpurun_out/r04au_kbench.txt
cat gpurun_out/r04au_kbench.txt
