for r in 1 2; do
for v in cur t1; do
  if [ $v = cur ]; then unset TSSEP_HIP_LIB; else export TSSEP_HIP_LIB=$PWD/tssep_amd/libtssep_hip_$v.so; fi
  echo $v $(TSSEP_GEMM_PRECISION=bf16x3 timeout 300 python tools/bench_gemm.py 192 2>/dev/null | grep name | grep -v '"tn"' | python -c "
import sys,json
print(' '.join(str(json.loads(l)['tflops']) for l in sys.stdin))")
done; done
