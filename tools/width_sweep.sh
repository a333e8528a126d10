#!/bin/bash
# Decoder forms (one, two, four waves per group) over batch sizes, VHT MCS8 1500-byte frames: kernel times from HIP
# events (forward pass, traceback where it is a kernel of its own) and the whole step.
#   bash tools/width_sweep.sh [frame counts ...]
SIZES=${@:-"8000 16000 24000 32000 40000 48000 56000 65536 72000"}
for n in $SIZES; do
  for w in 1 2 4; do
    export GR80211_GROUP_WAVES=$w
    timeout -k 10 200 python3 bench.py --workload vht_mcs8_1500B_snr30 --frames $n --steps 4 --warmup 1 --cpu-frames 0 --in-flight 1 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); k=r['kernel_ms']
        print('frames %6d groups %4d waves $w: step %.2f ms  decode %.2f (forward %.2f traceback %.2f)  %5.1f Gsamples/s  ok %d/%d' % ($n, ($n+63)//64, r['ms_per_step'], k['decode'], k['viterbi'], k['traceback'], r['value']/1e3, r['psdu']['bit_exact_vs_tx'], r['psdu']['fcs_ok']))
" || exit 1
  done
done
