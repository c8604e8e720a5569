timeout -k 10 300 python -u -m pytest tests/test_gpu_hu.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for v in "" hx1 hx2 hx3 hx4; do
  if [ -n "$v" ]; then export FX_LIB=build_dbg/libfx_$v.so; else unset FX_LIB; fi
  echo "== variant ${v:-base}"
  timeout -k 10 120 python3 tools/kbench.py hutail 2>&1 | grep '^hu' || exit 1
done
