#!/bin/bash
# usage: bash scripts/ab_build.sh <git-rev> [tag]   -> id-grec_amd/lib_<tag>/libidgrec.so built from that revision's csrc/ + include/
# (same flags as id-grec_amd/build.py; the ABI version must equal the working tree's: run benches with IDG_LIB_PATH=<that file>)
set -e
rev=$1; tag=${2:-base}
tmp=$(mktemp -d)
git archive $rev id-grec_amd/csrc include | tar -x -C $tmp
out=id-grec_amd/lib_$tag; mkdir -p $out $tmp/obj
objs=""
for f in $tmp/id-grec_amd/csrc/*.hip $tmp/id-grec_amd/csrc/*.cpp; do
  o=$tmp/obj/$(basename ${f%.*}).o; objs="$objs $o"
  x=""; case $f in *.hip) x="-x hip";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -I$tmp/include -I$tmp/id-grec_amd/csrc $x -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libidgrec.so $objs
rm -rf $tmp; ls -la $out/libidgrec.so
