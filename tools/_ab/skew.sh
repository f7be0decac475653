cd /root/repo
for cfg in "0 2" "10 2" "20 2" "5 4" "10 4" "0 2"; do
  set -- $cfg
  echo "== skew_us=$1 groups=$2"
  REED_GEMM_SKEW_US=$1 REED_GEMM_SKEW_G=$2 timeout -k 10 200 python tools/gemm_table.py 256 20 || exit 1
done
