set +e
mkdir -p gpurun_out/r3
for v in a b; do
  python -m pytest tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r3/bisect_$v.log 2>&1
  echo "run $v: $(tail -1 gpurun_out/r3/bisect_$v.log)"
done
python -m pytest tests -x -q -m gpu > gpurun_out/r3/bisect_all.log 2>&1
echo "all: $(tail -1 gpurun_out/r3/bisect_all.log)"
grep -n "At index" gpurun_out/r3/bisect_*.log | cut -c1-300
exit 0
