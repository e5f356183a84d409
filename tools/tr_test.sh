set -e
for v in "" tr4 tr8; do
  if [ -n "$v" ]; then export ADF_WLS_LIB=$GRAFT_REPO_ROOT/addingdisparityfiltering_amd/libadf_wls_$v.so; else unset ADF_WLS_LIB; fi
  echo "== variant ${v:-default}"
  timeout -k 10 500 python -m pytest tests/test_gpu_wave.py tests/test_gpu_fuzz.py tests/test_gpu_conf_band.py::test_thresholds_and_batch tests/test_scaled_path.py -x -q 2>&1 | tail -2
  ADF_NO_OVERLAP=1 python bench.py --cpu-seconds 0 --matcher-pairs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
done
