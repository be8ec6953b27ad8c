#!/bin/bash
# Parity subset + throughput of the challenge hash with each matrix-core knob off in turn.
# Build the variants first (CPU, ~3 min each, in parallel):
#   for v in "valu:-DDSV_HADES_MFMA=0" "noedge:-DDSV_HADES_MFMA_EDGE=0" "nomds:-DDSV_HADES_MFMA_MDS=0" \
#            "rolled:-DDSV_HADES_MFMA_UNROLL5=0"; do n=${v%%:*}; f=${v#*:}
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o build/ab/libdsv_$n.so schnorr_amd/csrc/dsv.hip $f & done; wait
# then on the GPU box:  bash tools/ab_hades_variants.sh
for v in valu noedge nomds rolled; do
  echo "== $v"
  DSV_LIB_PATH=$PWD/build/ab/libdsv_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r02.py -m gpu -x -q -k "challenge or tamper or hand_off or ragged or sweep" 2>&1 | tail -2
done
bash tools/ab_multi.sh 2 valu noedge nomds rolled cur
