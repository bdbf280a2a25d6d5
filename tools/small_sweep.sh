# Frame time of african_head subdivided lv times at 4K against the small-triangle threshold
# (TR_SMALL_RES; 0 = path off).  Run on the GPU box: bash tools/small_sweep.sh
export TINYRENDERER_HIP_DIAG=1  # TR_SMALL_TRIS / TR_SMALL_RES are read by the diagnostic build only
for lv in 2 3 4; do for r in 0 5 6 7 8; do
  if [ $r = 0 ]; then export TR_SMALL_TRIS=0; else export TR_SMALL_TRIS=1; export TR_SMALL_RES=$r; fi
  echo "level $lv small_res $r: $(timeout -k 10 120 python -u tests/manual/big_mesh_check.py $lv 3840 2160 0 0 | grep 'frame ms' | cut -c1-70)"
done; done
