set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_aligner.py tests/test_gpu_surface.py -m gpu -q -x > gpurun_out/r3_gputest3.log 2>&1; echo exit=$? >> gpurun_out/r3_gputest3.log
grep -E "passed|failed|exit=" gpurun_out/r3_gputest3.log | tail -3
bash tools/shape_trace.sh 128 512 5 > gpurun_out/r3_shape_128_512_dma.txt 2>&1; head -12 gpurun_out/r3_shape_128_512_dma.txt
KIRAG_AMD_ATTN_LDS=1 bash tools/shape_trace.sh 128 512 5 > gpurun_out/r3_shape_128_512_lds.txt 2>&1; grep -E "shape|attn" gpurun_out/r3_shape_128_512_lds.txt
bash tools/shape_trace.sh 256 256 5 > gpurun_out/r3_shape_256_256_dma.txt 2>&1; grep -E "shape|attn" gpurun_out/r3_shape_256_256_dma.txt
KIRAG_AMD_ATTN_LDS=1 bash tools/shape_trace.sh 256 256 5 > gpurun_out/r3_shape_256_256_lds.txt 2>&1; grep -E "shape|attn" gpurun_out/r3_shape_256_256_lds.txt
bash tools/shape_trace.sh 1024 128 5 > gpurun_out/r3_shape_1024_128.txt 2>&1; grep -E "shape|attn" gpurun_out/r3_shape_1024_128.txt
