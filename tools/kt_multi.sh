#!/bin/bash
# kernel traces of several builds in one GPU-box session: tools/kt_multi.sh "<lib> <lib> .." "<grep pattern>"
for lib in $1; do
  [ "$lib" = default ] && lib=dsk_amd/libdskgpu.so
  echo "== $lib"; DSKGPU_LIB=$PWD/$lib bash tools/ktrace.sh m_$(basename $lib .so) | grep "$2"
done
