#!/bin/sh
# debug build of libcst_hip with cycle stamps in the persistent GEMM (tools/gemm8p_trace.py); output: tools/trace/libcst_trace.so
set -e
cd "$(dirname "$0")/../chimera-st_amd/csrc"
mkdir -p ../../tools/trace
rm -f ../../tools/trace/*.o
for src in *.hip; do
  f=${src%.hip}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCST_TRACE -Wno-unused-function -I../../include -c $f.hip -o ../../tools/trace/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../tools/trace/*.o -o ../../tools/trace/libcst_trace.so
