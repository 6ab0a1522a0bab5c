mkdir -p gpurun_out
# build first: for v in "4 0" "8 0" "4 2" "4 1" "4 17"; do set -- $v; hipcc --offload-arch=gfx950 -O3 -w -DSKEL_DB=$1 -DSKEL_STORE_AUX=$2 -o tools/mrf_skeleton_db$1_aux$2 tools/mrf_skeleton.hip; done
for b in db4_aux0 db8_aux0 db4_aux2 db4_aux1 db4_aux17; do
  echo "== $b" >> gpurun_out/r04_skeleton6.txt
  timeout -k 10 100 tools/mrf_skeleton_$b >> gpurun_out/r04_skeleton6.txt 2>&1 || exit 1
done
grep -E "^==|W P S E V: |without the stores|store burst|timed" gpurun_out/r04_skeleton6.txt
