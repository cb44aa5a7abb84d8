#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side (csrc/*.c) and the oracle's C, driven by the CPU test suite (no GPU: sanitizers
# run on the CPU build only).  The HIP objects are linked as they are.  The genuine reference binaries (oracle/_ref) inherit LD_PRELOAD and three of
# their runs end in a SEGV under ASan's allocator (hostile labels through xtree-search; PACKSIZE=16 tables with duplicate / non-monotone bins):
# undefined behaviour in the reference, whose unsanitised answers the goldens hold -- not findings in this tree.
set -e
R=/root/repo
mkdir -p /tmp/asan
cd $R/utree_amd/csrc
SF="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g -std=gnu11 -fPIC -fopenmp -I/opt/rocm/include"
for f in ctr_host dev_image fasta search search_dev rccl_replicate compress rank build; do gcc $SF -c $f.c -o /tmp/asan/$f.o; done
HIPO=$(ls kernels.o lanes_kernel.o lanes_part_*.o rank_kernels.o text_kernels.o build_gpu.o image_build.o)
gcc -shared -fopenmp -fsanitize=address,undefined -o /tmp/asan/libutree_amd_asan.so $HIPO /tmp/asan/*.o -L/opt/rocm/lib -lamdhip64 -lrccl -lstdc++ -lz -lm -lpthread -Wl,-rpath,/opt/rocm/lib
cd $R/oracle
gcc -O1 -g -std=gnu11 -fopenmp -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o /tmp/asan/liboracle.so utree_oracle.c utree_build_oracle.c
cp liboracle.so /tmp/asan/liboracle_plain.so
cp /tmp/asan/liboracle.so liboracle.so
trap "cp /tmp/asan/liboracle_plain.so $R/oracle/liboracle.so" EXIT
cd $R
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
    UTREE_AMD_SO=/tmp/asan/libutree_amd_asan.so python -m pytest tests/test_host_cpu.py tests/test_oracle_golden.py tests/test_dist_gloo.py -q -m "not gpu" > /tmp/asan/tests.log 2>&1 || true
tail -6 /tmp/asan/tests.log
echo "sanitizer reports outside oracle/_ref:"
grep -n "runtime error" /tmp/asan/tests.log || true
grep -n "ERROR: AddressSanitizer" -A4 /tmp/asan/tests.log | grep -v "_ref\|libc.so\|ERROR: AddressSanitizer\|memory access\|^--" || echo "  none"
