# usage (GPU box): bash tools/probes/waves_size_sweep.sh [tess ...] — where the seven-block trace variant stops paying:
# the Cornell room with spheres of tess x tess/2 quads, six against seven blocks per CU forced (TWK_TRACE_WAVES_RUNTIME)
for t in ${@:-180 360 500 700 1000}; do
  for w in 6 7; do
    echo -n "tess $t blocks $w: "
    TWK_TRACE_WAVES_RUNTIME=$w timeout -k 10 300 python tools/big_scene_probe.py $t
  done
done
