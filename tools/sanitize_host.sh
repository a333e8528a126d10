#!/bin/bash
# Host-side sanitizer check of the C++ blocks: builds gr80211_pump and the plain-C driver of the streaming ABI with
# AddressSanitizer and UBSan (host code only, plain g++ / gcc; the HIP library is linked as built) into
# tools/san_tmp/, to be run by the tests that start these programs, on a GPU machine:
#   tools/sanitize_host.sh
#   GR80211_PUMP=$PWD/tools/san_tmp/gr80211_pump_san GR80211_ABI_CHAIN=$PWD/tools/san_tmp/gr80211_abi_chain_san \
#       ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest tests/test_gpu_host_blocks.py -m gpu -q
# (leak detection stays off: the HIP runtime keeps allocations until exit).  Remove tools/san_tmp afterwards.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/san_tmp
cd fork-gr-ieee80211_amd/host
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -static-libasan -static-libubsan -pthread pump.cpp gr80211_blocks.cpp host_api.cpp \
    -o ../../tools/san_tmp/gr80211_pump_san -L.. -lgr80211_rx -Wl,-rpath,'$ORIGIN/../../fork-gr-ieee80211_amd'
gcc -O1 -g -std=c99 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -I../../include -c abi_chain.c -o ../../tools/san_tmp/abi_chain.o
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -static-libasan -static-libubsan -pthread ../../tools/san_tmp/abi_chain.o gr80211_blocks.cpp host_api.cpp \
    -o ../../tools/san_tmp/gr80211_abi_chain_san -L.. -lgr80211_rx -Wl,-rpath,'$ORIGIN/../../fork-gr-ieee80211_amd'
echo built tools/san_tmp/gr80211_pump_san tools/san_tmp/gr80211_abi_chain_san
