#!/bin/bash
# A/B library: tools/build_variant.sh NAME "EXTRA HIPCC FLAGS" file1.hip [file2.hip ...]
# Recompiles only the named csrc files with the extra flags, links them with the objects of the regular build into
# transformertts_amd/lib/libttsmi_NAME.so; select it with TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=... (tools/kbench.py --variants, tools/ab_env.sh).
set -e
name=$1; extra=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/transformertts_amd/build; var=$obj/variant_$name; mkdir -p "$var"
python -c "import sys; sys.path.insert(0,'$root'); from transformertts_amd import build; build.build(verbose=False)"
objs=""
for o in "$obj"/*.o; do
  b=$(basename "$o" .o); use=$o
  for f in "$@"; do
    if [ "$b" = "$f" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $extra -x hip -c "$root/transformertts_amd/csrc/$f" -o "$var/$b.o" 2>"$var/$b.log" &
      use=$var/$b.o
    fi
  done
  objs="$objs $use"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/transformertts_amd/lib/libttsmi_$name.so" $objs -ldl
echo "$root/transformertts_amd/lib/libttsmi_$name.so"
