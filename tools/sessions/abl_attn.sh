for a in 0 1 2 4 8 16 32 3 6 7 12 63; do
  TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_ablate.so TTSMI_ATTN_ABLATE=$a python tools/kbench.py --only attn 2>/dev/null | grep "fwd bits p=0.1      28800\|fwd p=0.0           28800" | awk -v a=$a '{print "ablate", a, $3, $4, $8}'
done
