mkdir -p gpurun_out
for b in db4_aux0 db8_aux0 db4_aux2 db4_aux1 db4_aux17; do
  echo "== $b" >> gpurun_out/r04_skeleton6.txt
  timeout -k 10 100 tools/mrf_skeleton_$b >> gpurun_out/r04_skeleton6.txt 2>&1 || exit 1
done
grep -E "^==|W P S E V: |without the stores|store burst|timed" gpurun_out/r04_skeleton6.txt
