# the -DREED_EPI_EXP=1 probe this script timed has been removed from gemm_common.hpp (result recorded)
cd /root/repo
for rep in 1 2; do
for lib in "" tools/_ab/libreed_epi1.so; do
  echo "== lib=${lib:-current}"
  REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 || exit 1
done
done
