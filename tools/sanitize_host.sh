#!/bin/bash
# Host-side sanitizer check of the C++ blocks: builds gr80211_pump with AddressSanitizer and UBSan (host code only,
# plain g++; the HIP library is linked as built) into tools/san_tmp/, to be run by the pump tests on a GPU machine:
#   tools/sanitize_host.sh
#   GR80211_PUMP=$PWD/tools/san_tmp/gr80211_pump_san ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
#       python -m pytest tests/test_gpu_host_blocks.py -m gpu -q
# (leak detection stays off: the HIP runtime keeps allocations until exit).  Remove tools/san_tmp afterwards.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/san_tmp
cd fork-gr-ieee80211_amd/host
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined \
    -static-libasan -static-libubsan pump.cpp gr80211_blocks.cpp host_api.cpp \
    -o ../../tools/san_tmp/gr80211_pump_san -L.. -lgr80211_rx -Wl,-rpath,'$ORIGIN/../../fork-gr-ieee80211_amd'
echo built tools/san_tmp/gr80211_pump_san
