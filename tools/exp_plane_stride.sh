export PFFDTD_VERBOSE=1
for ny in 1024 1025 1028 1032 1040 1056 1088; do
  echo "=== ny $ny" 
  timeout 300 python bench.py --steps 8 --warmup 4 --repeats 3 --no-rigid-run --no-cpu-baseline --ny $ny 2>&1 | grep -v "^\[W\|Warning" | tail -4
done
