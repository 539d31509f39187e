export TMPDIR=/tmp; mkdir -p gpurun_out/r4c
python tools/dbg/gemm_mismatch.py > gpurun_out/r4c/mismatch.log 2>&1; cat gpurun_out/r4c/mismatch.log | grep -v amdgpu.ids
python -m pytest tests/test_gpu_kernels.py -q -k "320_row_tile" -x 2>&1 | grep -E "^E |passed|failed" | head -30
python -m pytest tests/test_gpu_modules.py -q -k "bench_default" -x 2>&1 | grep -E "^E |passed|failed" | head -30
