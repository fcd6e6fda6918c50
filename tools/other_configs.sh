# The single-GPU configurations besides the default bench line, and the mode benches:
#   bash tools/other_configs.sh <tag>     (on the GPU box, from the repository root)
TAG=${1:-r02d}
out=gpurun_out/$TAG
mkdir -p $out
timeout -k 10 200 python bench.py --no-cpu-baseline --particles-log2 20 --grid 128,128,128 > $out/bench_cfg2_1M_128cube.json 2>> $out/other.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --particles 81920 --grid 128,64,64 --noise 1.0 --steps 400 --warmup 50 > $out/bench_cfg1_81920_128x64x64.json 2>> $out/other.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --particles-log2 25 --scene dam_break --steps 200 > $out/bench_cfg5_dam_break_32M_256cube.json 2>> $out/other.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --noise 1.0 > $out/bench_16M_256cube_noise1.json 2>> $out/other.err || exit 1
for m in front_gravity_cylinder sphere polygon color_spheres segment image_grid everything; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --mode $m --steady-windows 0 > $out/bench_mode_$m.json 2>> $out/other.err || exit 1
done
python - $out <<'PY'
import json, glob, sys, os
out = sys.argv[1]
modes = {}
for f in sorted(glob.glob(out + "/bench_*.json")):
    d = json.load(open(f))
    name = os.path.basename(f)[6:-5]
    s = d.get("steady") or {}
    print("%-34s %6.2f G/s %7.4f ms | steady %s | rebins %s (%.3f ms/step) | %s" % (
        name, d["value"] / 1e9, d["ms_per_step"], ("%6.2f G/s" % (s["value"] / 1e9)) if s else "   --    ",
        d["rebin"]["performed"], d["rebin"]["ms_per_step"], {n: x["avg_ms"] for n, x in d["kernels"].items() if n in ("p2g1", "p2g2", "g2p")}))
    if name.startswith("mode_"):
        modes[name[5:]] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "kernels": {n: x["avg_ms"] for n, x in d["kernels"].items()},
                           "rebin": d["rebin"], "config": d["config"]}
json.dump(modes, open(out + "/bench_modes_16M_256cube.json", "w"), indent=1)
PY
echo done
