# alternating A/B of two builds of the library (GPU box): bash tools/ab_stft_noslp.sh libA.so libB.so
B="--no-cpu-baseline --no-exact-f32"
LIBS="${@:-libtssep_hip.so libtssep_hip_noslp.so}"
for rep in 1 2; do
for lib in $LIBS; do
  TSSEP_HIP_LIB=$GRAFT_REPO_ROOT/tssep_amd/$lib python bench.py --steps 12 --warmup 3 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(lib='$lib', ms_per_step=d['ms_per_step'], mask_head_frac=d['roofline_mask_head']['frac'], mask_head_avg_ms=d['roofline_mask_head']['avg_ms'])))"
done; done
for lib in $LIBS; do
  TSSEP_HIP_LIB=$GRAFT_REPO_ROOT/tssep_amd/$lib python bench.py --workload cfg5 --steps 5 --warmup 2 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(workload='cfg5', lib='$lib', ms_per_step=d['ms_per_step'], mask_head_frac=d['roofline_mask_head']['frac'], mask_head_avg_ms=d['roofline_mask_head']['avg_ms'])))"
done
