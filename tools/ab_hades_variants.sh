for v in valu noedge nomds rolled; do
  echo "== $v"
  DSV_LIB_PATH=$PWD/build/ab/libdsv_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r02.py -m gpu -x -q -k "challenge or tamper or hand_off or ragged or sweep" 2>&1 | tail -2
done
bash tools/ab_multi.sh 2 valu noedge nomds rolled cur
