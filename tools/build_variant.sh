#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG ...] : utree_amd/libexp_<name>.so = the library with kernels.hip, lanes_kernel.hip, lanes_part.hip and dev_image.c
# compiled with extra flags (same-box A/B of kernel variants: UTREE_AMD_SO selects the library for bench.py).  ALL=1: every file that sees
# device_common.hpp / utree_internal.h is compiled with the flags (a variant of the image format, e.g. -DUTREE_CANON_MODE=2).
set -e
N=$1; shift
cd /root/repo/utree_amd/csrc
HF="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-parameter -Wno-unused-function"
/opt/rocm/bin/hipcc $HF -mllvm -amdgpu-load-store-vectorizer=0 "$@" -c kernels.hip -o /tmp/kernels_$N.o &
/opt/rocm/bin/hipcc $HF "$@" -c lanes_kernel.hip -o /tmp/lanes_kernel_$N.o &
PARTS=""
for P in 8_2_1_0 8_2_1_1 8_2_2_0 8_4_1_0 8_4_1_1 8_4_2_0 16_2_1_0 16_2_1_1 16_2_2_0; do
    IFS=_ read W I NL BS <<< "$P"
    /opt/rocm/bin/hipcc $HF "$@" -DLANES_W=$W -DLANES_I=$I -DLANES_NL=$NL -DLANES_BS=$BS -c lanes_part.hip -o /tmp/lanes_part_${P}_$N.o &
    PARTS="$PARTS /tmp/lanes_part_${P}_$N.o"
done
gcc -std=gnu11 -O2 -g -fPIC -fopenmp -I/opt/rocm/include "$@" -c dev_image.c -o /tmp/dev_image_$N.o
OBJS=$(echo text_kernels.o build_gpu.o ctr_host.o fasta.o search.o search_dev.o rccl_replicate.o compress.o rank.o build.o)
if [ -n "$ALL" ]; then
    /opt/rocm/bin/hipcc $HF "$@" -c image_build.hip -o /tmp/image_build_$N.o &
    /opt/rocm/bin/hipcc $HF "$@" -c rank_kernels.hip -o /tmp/rank_kernels_$N.o &
    OBJS="$OBJS /tmp/image_build_$N.o /tmp/rank_kernels_$N.o"
else
    OBJS="$OBJS image_build.o rank_kernels.o"
fi
wait
gcc -shared -fopenmp -o ../libexp_$N.so /tmp/kernels_$N.o /tmp/lanes_kernel_$N.o $PARTS /tmp/dev_image_$N.o $OBJS -L/opt/rocm/lib -lamdhip64 -lrccl -lstdc++ -lz -lm -lpthread -Wl,-rpath,/opt/rocm/lib
echo built libexp_$N.so
