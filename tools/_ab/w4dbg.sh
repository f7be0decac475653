cd /root/repo
for d in 0 1 2 3; do echo "== dbg=$d"; REED_W4_DBG=$d timeout -k 10 120 python tools/_ab/lat_probe.py 2>&1 | grep -E "M=256 |M=3584 |M=14336 " ; done
