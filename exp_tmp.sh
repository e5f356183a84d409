set -e
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for v in 1 0; do
  export ADF_NO_OVERLAP=$v
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_$v.json"))
k=d["kernels"]
print("no_overlap=$v", d["ms_per_step"], d["checked"]["disparity_max_abs_lsb"], {n:round(k[n]["ms_per_step"]/k[n]["launches"]*10,4) for n in k})
PY
done
