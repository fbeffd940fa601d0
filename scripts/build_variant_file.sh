#!/bin/bash
# build an ablation variant of the library from ONE recompiled source: build_variant_file.sh NAME FILE.hip [extra hipcc flags]
# -> fourierdiffusion_amd/libfdiff_hip_NAME.so (select with FDIFF_LIB=...)
set -e
NAME=$1; FILE=$2; shift; shift
STEM=${FILE%.hip}
cd /root/repo/fourierdiffusion_amd/csrc
make -s -j8
/opt/rocm/bin/hipcc -O3 -fno-honor-nans -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $FILE -o build/var_${STEM}_$NAME.o
OBJS=$(ls build/fd_*.o | grep -v "build/${STEM}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfdiff_hip_$NAME.so $OBJS build/var_${STEM}_$NAME.o -ldl
echo built libfdiff_hip_$NAME.so
