mkdir -p gpurun_out
run() { # name env...
  name=$1; shift
  env "$@" python bench.py --workload $WL --no-extras --steps 10 --warmup 3 --repeats 2 --detail-out gpurun_out/r05_ab_${WL}_$name.json 2>/dev/null > /dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r05_ab_${WL}_$name.json")); r=d["roofline"]
print("%-8s %-12s %7.3f it/s  %s" % ("$WL","$name", d["value"], "  ".join("%s %.1f us %.3f" % (k, v["avg_us"], r["frac_by_site"][k]) for k,v in r["use_sites"].items())), flush=True)
PY
}
# "default" = what the library picks by itself: since round 5 that is 8 waves per block at 128 float32 columns on EVERY layout (engine.hpp
# ct_kw), so the A/B arm is the forced 4-wave block (LCX_CT8_KW=4) on both layouts - the comparison behind profiles/r05_layout_ab_one_box.txt
for rep in 1 2; do
WL=c4shard; run rows_default LCX_X_LAYOUT=rows; run rows_kw4 LCX_X_LAYOUT=rows LCX_CT8_KW=4; run panel_default LCX_X_LAYOUT=; run panel_kw4 LCX_CT8_KW=4
WL=c3; run rows_default LCX_X_LAYOUT=rows; run rows_kw4 LCX_X_LAYOUT=rows LCX_CT8_KW=4; run panel_default LCX_X_LAYOUT=; run panel_kw4 LCX_CT8_KW=4
done
