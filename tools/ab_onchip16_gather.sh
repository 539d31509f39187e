# alternating: where the exchange waves request the next phase's operands (alt builds of lstm_onchip.hip with
# -DONCHIP16_FWD_GATHER=v / -DONCHIP16_BWD_GATHER=v linked into tssep_amd/libtssep_hip_g<v>.so); $1 = fwd | bwd, $2.. = values
dir=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ -f "$PWD/tssep_amd/libtssep_hip_g$v.so" ]; then L="$PWD/tssep_amd/libtssep_hip_g$v.so"; else L=""; fi
    if [ $dir = bwd ]; then
      TSSEP_HIP_LIB=$L python tools/bench_onchip16.py ${SIZES:-768 3072} 2>/dev/null | grep backward | sed "s/^{/{\"gather\": $v, /"
    else
      TSSEP_HIP_LIB=$L python tools/bench_onchip16.py ${SIZES:-768 1536} 2>/dev/null | grep -v backward | sed "s/^{/{\"gather\": $v, /"
    fi
  done
done
