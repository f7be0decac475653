# build the variant first: python tools/_ab/build_variant.py <name> [-D...]; pass tools/_ab/libreed_<name>.so as $1
cd /root/repo
LIBV=${1:-tools/_ab/libreed_tnld.so}
REED_HIP_LIB=$LIBV timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -p no:cacheprovider -k "wgrad or tn" 2>&1 | tail -2 || exit 1
for rep in 1 2; do
for lib in "" $LIBV; do
  echo "== lib=${lib:-current}"
  REED_HIP_LIB=$lib timeout -k 10 200 python tools/bench_wgrad_group.py 256 128 32 2>&1 | sed 's/| 4-wave.*//' || exit 1
done
done
