#!/bin/bash
# tools/build_variant_k32.sh <name> [-DFLAG ...]: like build_variant.sh, but only the k = 32 / u16 lane kernels (lanes_part 8_2_*) are compiled with the flags --
# the other objects are the main build's (a third of the time; for variants that only matter to those kernels)
set -e
N=$1; shift
cd /root/repo/utree_amd/csrc
HF="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-parameter -Wno-unused-function"
PARTS=""
for P in 8_2_1_0 8_2_1_1 8_2_2_0; do
    IFS=_ read W I NL BS <<< "$P"
    /opt/rocm/bin/hipcc $HF "$@" -DLANES_W=$W -DLANES_I=$I -DLANES_NL=$NL -DLANES_BS=$BS -c lanes_part.hip -o /tmp/lanes_part_${P}_$N.o &
    PARTS="$PARTS /tmp/lanes_part_${P}_$N.o"
done
/opt/rocm/bin/hipcc $HF "$@" -c lanes_kernel.hip -o /tmp/lanes_kernel_$N.o &
wait
OBJS=$(ls *.o | grep -v "lanes_part_8_2_\|lanes_kernel.o" | tr '\n' ' ')
gcc -shared -fopenmp -o ../libexp_$N.so /tmp/lanes_kernel_$N.o $PARTS $OBJS -L/opt/rocm/lib -lamdhip64 -lrccl -lstdc++ -lz -lm -lpthread -Wl,-rpath,/opt/rocm/lib
echo built libexp_$N.so
