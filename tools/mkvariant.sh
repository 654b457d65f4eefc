#!/bin/bash
# tools/mkvariant.sh <name> <source.hip> <-D flags...>: a variant of the library (tools/_build/lib_<name>.so) with ONE source
# recompiled with extra flags -- ablation builds for tools/ab_variants.sh (e.g. -DPT_NOB / -DPT_NOC / -DPT_NODMA in
# se_attention.hip: the LDS-staged P~ pass without phase B / phase C / the stage DMA).  Needs the objects of a normal build.
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p tools/_build; o=tools/_build/var_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c sketchedit_amd/csrc/$src -o $o || exit 1
objs=$(ls sketchedit_amd/lib/obj/*.o | grep -v "/${src%.hip}.o")
hipcc --offload-arch=gfx950 -fPIC -shared -o tools/_build/lib_$name.so $objs $o && echo built $name
