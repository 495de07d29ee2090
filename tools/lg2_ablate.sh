set -e
export SLAK_BUILD_DEFS="-DSLAK_LG2_DEV -DSLAK_DEV_KNOBS"
python -m slak_amd.build > /dev/null 2>&1 || { echo build failed; python -m slak_amd.build 2>&1 | tail -20; exit 1; }
for d in 0 4 5 6 3; do echo "== dbg $d"; SLAK_LG2_DBG=$d timeout 200 python tools/time_gemm2.py 384 2>&1 | grep own | sed 's/|.*//'; done
