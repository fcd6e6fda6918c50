# The single-GPU configurations besides the default bench line, and the mode benches:
#   bash tools/other_configs.sh <tag>     (on the GPU box, from the repository root)
TAG=${1:-r01e}
out=gpurun_out/$TAG
mkdir -p $out
timeout -k 10 200 python bench.py --no-cpu-baseline --particles-log2 20 --grid 128,128,128 > $out/bench_cfg2_1M_128cube.json 2>> $out/other.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --particles-log2 25 --scene dam_break --steps 200 > $out/bench_cfg5_dam_break_32M_256cube.json 2>> $out/other.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --noise 1.0 > $out/bench_16M_256cube_noise1.json 2>> $out/other.err || exit 1
for m in front_gravity_cylinder sphere polygon color_spheres segment image_grid everything; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --mode $m > $out/bench_mode_$m.json 2>> $out/other.err || exit 1
done
echo done
