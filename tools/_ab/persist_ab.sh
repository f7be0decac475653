cd /root/repo
REED_GEMM_PERSIST=2 timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 || exit 1
for b in 256 128; do
for rep in 1 2; do
for v in 0 1 2; do
  echo "== b=$b REED_GEMM_PERSIST=$v"
  REED_GEMM_PERSIST=$v timeout -k 10 200 python tools/gemm_table.py $b 20 || exit 1
done
done
done
