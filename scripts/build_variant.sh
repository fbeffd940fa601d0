#!/bin/bash
# build an ablation variant of the library: build_variant.sh NAME [extra hipcc flags for fd_mega.hip]
# -> fourierdiffusion_amd/libfdiff_hip_NAME.so (select with FDIFF_LIB=...; see scripts/gpu_variants.sh)
set -e
NAME=$1; shift
cd /root/repo/fourierdiffusion_amd/csrc
make -s -j8
/opt/rocm/bin/hipcc -O3 -fno-honor-nans -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c fd_mega.hip -o build/var_fd_mega_$NAME.o
OBJS=$(ls build/fd_*.o | grep -v "build/fd_mega.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfdiff_hip_$NAME.so $OBJS build/var_fd_mega_$NAME.o -ldl
echo built libfdiff_hip_$NAME.so
