#!/bin/bash
# The library's HOST code under AddressSanitizer, on the CPU tests that need no device (the graph behind the ABI: readers, picker and graphalign for graphs, the anchors'
# surgery, prune_nodes, the writer; chain()): bash tools/asan_cpu.sh        (GPU sanitizers are not available on the pool; device code is untouched by this build)
set -e
R=$PWD; B=${1:-/tmp/reveal_asan}
mkdir -p $B/out
make -C reveal_amd/csrc -j8 BUILD=$B/obj OUT=$B/out EXTRA="-fsanitize=address -fno-omit-frame-pointer -g" $B/out/libreveal_amd.so > $B/build.log 2>&1
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 RV_LIB_DIR=$B/out python -m pytest tests/test_cpu_graphrem_native.py tests/test_cpu_graph_native.py tests/test_cpu_chain.py tests/test_cpu_picker_fuzz.py -x -q
# the same under UndefinedBehaviorSanitizer (reports go to stderr: -s shows them; none = clean)
mkdir -p $B/ubout
make -C reveal_amd/csrc -j8 BUILD=$B/ubobj OUT=$B/ubout EXTRA="-fsanitize=undefined -fno-sanitize=vptr -fno-omit-frame-pointer -g" $B/ubout/libreveal_amd.so > $B/build_ub.log 2>&1
UB=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
LD_PRELOAD=$UB UBSAN_OPTIONS=halt_on_error=0 RV_LIB_DIR=$B/ubout python -m pytest tests/test_cpu_graphrem_native.py tests/test_cpu_graph_native.py tests/test_cpu_chain.py -x -q -s 2>&1 | grep -c "runtime error" || true
