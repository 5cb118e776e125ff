for f in 1 0; do
PN2_POOL_EPILOGUE=$f python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab_$f.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$f.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("flag $f step", d["ms_per_step"], "serial", d["roofline"]["device_ms_all_kernels_per_step"])
for n in ("pn2_conv1x1_fwd","pn2_bn_relu_max","pn2_bn_pool_select","pn2_bn_finalize"):
    print("  ", n, k.get(n))
PY
done
