for V in NOW NOX "NOW -DCV_EXP_NOX"; do
  (cd superpixel-align_amd/csrc && touch spa_conv.hip && make EXTRA="-DCV_EXP_$V" > /dev/null 2>&1)
  echo "== variant $V" >> gpurun_out/r2_cv2.log
  python3 tools/conv_bench.py 2>&1 | grep "B 30   512" >> gpurun_out/r2_cv2.log
done
cat gpurun_out/r2_cv2.log
