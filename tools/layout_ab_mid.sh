#!/bin/bash
# mid-size shards: does the panel layout / the 8-wave block of 128 float32 columns hold outside BASELINE's shapes?
# (bench.py --no-extras per shape under LCX_X_LAYOUT / LCX_CT8_KW, one box)
mkdir -p gpurun_out
run() { # name env...
  name=$1; shift
  env "$@" python bench.py --workload $WL --no-extras --steps 10 --warmup 3 --detail-out gpurun_out/r04_abmid_$name.json 2>/dev/null > /dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r04_abmid_$name.json")); r=d["roofline"]; c=d["config"]
print("%-22s %-10s %9.2f it/s  %-24s %s" % ("$WL","$name", d["value"], c["bytes_resident"]["x_layout"][:24], "  ".join("%s %.1f us" % (k, v["avg_us"]) for k,v in r["use_sites"].items())), flush=True)
PY
}
for WL in c2m128f32 c2m64f32 mid64f32 mid32f32 20000x20000x128:f32 1000x120000x60:f32 30000x40000x128:f32 c3f64; do
  run rows LCX_X_LAYOUT=rows LCX_CT8_KW=4; run panel_kw4 LCX_CT8_KW=4; run panel_kw8 LCX_CT8_KW=8
done
