cd /root/repo
REED_HIP_LIB=tools/_ab/libreed_ring3.so timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -p no:cacheprovider -k "wgrad or tn" 2>&1 | tail -2 || exit 1
for rep in 1 2; do
for lib in tools/_ab/libreed_ring2.so tools/_ab/libreed_ring3.so; do
  echo "== lib=${lib:-current}"
  REED_HIP_LIB=$lib timeout -k 10 200 python tools/bench_wgrad_group.py 256 128 32 2>&1 | sed 's/| 4-wave.*//' || exit 1
done
done
REED_HIP_LIB=tools/_ab/libreed_seg.so timeout -k 10 300 python tools/_ab/seg_tn.py 256
