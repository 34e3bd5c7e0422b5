for rep in 1 2; do
for lib in "" tools/bin/ft512.so; do
  echo "== lib=${lib:-default(256)} rep $rep"
  for wl in bcast_mul pow; do
    SMHIP_LIBRARY=$lib python bench.py --workload $wl --no-cpu-baseline --configs none | python -c "import json,sys; d=json.load(sys.stdin); print('$wl replay', round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"
    SMHIP_LIBRARY=$lib python bench.py --workload $wl --setting cold --no-cpu-baseline --configs none | python -c "import json,sys; d=json.load(sys.stdin); print('$wl cold  ', round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"
  done
done
done
