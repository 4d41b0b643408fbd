#!/bin/bash
# A/B builds of the product library: tools/build_variant.sh <name> <unit[,unit...]> <extra hipcc flags...>
#   recompiles only the listed translation units with the extra flags (objects under build_ab/<name>/), links them with the
#   default build's other objects into abtest/lib_<name>.so; run a test or bench.py against it with DART_STEPPER_LIB=abtest/lib_<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; UNITS=$2; shift 2
mkdir -p $R/build_ab/$NAME $R/abtest
OBJS=""
for u in dart_stepper planar_f32 planar_f64 spatial_f32 spatial_f64; do
  if [[ ",$UNITS," == *",$u,"* ]]; then
    UF=""; if [[ $u == spatial_* && -z "$DART_NO_UNIT_FLAGS" ]]; then UF="-mllvm -disable-machine-licm"; fi   # the product build's per-unit flags (__graft_entry__.UNIT_FLAGS)
    # (-save-temps=obj: the device assembly of this very compilation, for the EXEC-prologue lint below -- as __graft_entry__.build() does)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $UF "$@" -save-temps=obj -Rpass-analysis=kernel-resource-usage \
      -c $R/dart_env_amd/csrc/$u.hip -o $R/build_ab/$NAME/$u.o 2> $R/build_ab/$NAME/$u.res.txt &
    OBJS="$OBJS $R/build_ab/$NAME/$u.o"
  else
    OBJS="$OBJS $R/build/obj/$u.o"
  fi
done
wait
for a in $R/build_ab/$NAME/*-hip-amdgcn-amd-amdhsa-gfx950.s; do
  [ -f "$a" ] && python3 $R/tools/exec_prologue_lint.py "$a" | head -3
done
rm -f $R/build_ab/$NAME/*.bc $R/build_ab/$NAME/*.hipi $R/build_ab/$NAME/*.hipfb $R/build_ab/$NAME/*.out $R/build_ab/$NAME/*-host-*.s $R/build_ab/$NAME/*-host-*.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/abtest/lib_$NAME.so
echo "built abtest/lib_$NAME.so"
