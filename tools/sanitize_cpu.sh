#!/bin/bash
# AddressSanitizer + UBSan over the CPU builds (GPU sanitizers are not available on the pool): the host
# emulations of the wavefront program (LDS indexing, slot layout, reductions) and of the workgroup program (LDS layout, unit
# stride, every parallel region; generic and BOX instantiation, n = 2, 3, 6, a degree-24 star), of the region-terminal solve (work-array
# layout, every phase) and the oracle (region terminals included).
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -Igcs_admm_amd/csrc tests/hostemu/emu.cpp -o /tmp/libemu_asan.so
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -Igcs_admm_amd/csrc tests/hostemu/wg_emu.cpp -o /tmp/libwgemu_asan.so
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -Igcs_admm_amd/csrc tests/hostemu/term_emu.cpp -o /tmp/libtermemu_asan.so
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -fopenmp -shared -o /tmp/libo_asan.so oracle/gcs_oracle.c -lm
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
python tools/sanitize_cpu.py
