cd /root/repo
for gm in 4 1 2 8 16 4; do echo "GM=$gm"; REED_GEMM256_GM=$gm timeout -k 10 200 python tools/gemm_table.py 256 100 2>&1 | grep -v amdgpu.ids; done
