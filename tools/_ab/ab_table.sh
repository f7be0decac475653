# needs a baseline build first: git worktree of the commit to compare + python tools/_ab/build_variant.py hip_base there,
# or any variant built with tools/_ab/build_variant.py <name> [flags]; pass its path as $1
cd /root/repo
for rep in 1 2; do
for lib in tools/_ab/libreed_hip_base.so ""; do
  echo "== lib=${lib:-current}"
  REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 || exit 1
done
done
