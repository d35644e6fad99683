#!/bin/sh
# debug build of libcst_hip with cycle stamps in the persistent GEMM (tools/gemm8p_trace.py); output: tools/trace/libcst_trace.so
set -e
cd "$(dirname "$0")/../chimera-st_amd/csrc"
mkdir -p ../../tools/trace
for f in cst_core gemm gemm8p attention layernorm conv0 elementwise loss_optim; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCST_TRACE -Wno-unused-function -I../../include -c $f.hip -o ../../tools/trace/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../tools/trace/*.o -o ../../tools/trace/libcst_trace.so
