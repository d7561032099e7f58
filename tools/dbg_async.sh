cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "== run $i"
  timeout -k 10 280 rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop print" -ex run -ex "bt 40" --args python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_async" > gpurun_out/dbg_$i.log 2>&1
  tail -3 gpurun_out/dbg_$i.log
  if grep -q "SIGSEGV" gpurun_out/dbg_$i.log; then grep -A45 "SIGSEGV" gpurun_out/dbg_$i.log | head -80; break; fi
done
