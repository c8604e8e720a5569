set -e
for rep in 1 2 3; do
for v in main nw2; do
  if [ $v = main ]; then lib=$PWD/image-processing_amd/libfx.so; else lib=$PWD/gpurun_alt/libfx_$v.so; fi
  echo "== $v rep $rep"
  FX_LIB=$lib timeout -k 10 200 python bench.py --workload c4 --steps 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  FX_LIB=$lib timeout -k 10 120 python tools/kbench.py pca 2>/dev/null | grep "N=256"
done
done
