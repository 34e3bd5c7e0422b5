for rep in 1 2; do
for lib in "" tools/bin/preload.so; do
  echo "== lib=${lib:-default} rep $rep"
  for wl in add bcast_mul pow chain; do
    SMHIP_LIBRARY=$lib python bench.py --workload $wl --no-cpu-baseline --configs none | python -c "import json,sys; d=json.load(sys.stdin); print('$wl', round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"
  done
  SMHIP_LIBRARY=$lib python bench.py --workload bcast_mul --setting cold --no-cpu-baseline --configs none | python -c "import json,sys; d=json.load(sys.stdin); print('c3 cold', round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['roofline']['frac'],4))"
done
done
SMHIP_LIBRARY=tools/bin/preload.so timeout 300 python tools/chain_fused_rates.py "" short
