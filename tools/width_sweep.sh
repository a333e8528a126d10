for wl_n in "vht_mcs4_1500B_snr30 10000" "vht_mcs8_1500B_snr30 16000" "vht_mcs8_1500B_snr30 32000" "ht_mcs15_2x2_1500B_snr30 50000"; do
  set -- $wl_n
  for w in 1 4; do
    export GR80211_GROUP_WAVES=$w
    timeout -k 10 200 python3 bench.py --workload $1 --frames $2 --steps 4 --warmup 1 --cpu-frames 0 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); k=r['kernel_ms']
        print('$1 n=$2 w=$w value %.0f ms %.2f  viterbi %.2f tb %.2f demod %.2f pre %.2f ok %d/%d' % (r['value'], r['ms_per_step'], k['viterbi'], k['traceback'], k['demod'], k['preamble'], r['psdu']['bit_exact_vs_tx'], r['psdu']['frames']))
" || exit 1
  done
done
