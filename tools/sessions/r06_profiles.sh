#!/bin/bash
# round-6 closing profiles (TAG, default r06a): default bench (kernel trace + the four PMC passes), the same with the blur off, BASELINE
# configs 2-5 and naive_interp / sharp (kernel trace + bench line + PMC passes), the technique table, the by-depth table with the scene8
# column, the tie-path bench lines, the host bench; profiles/pmc_traffic.json AND profiles/pmc_valu.json regenerated from this run's passes
TAG=${1:-r06a}
bash tools/gpu_profile.sh ${TAG}_blur_on > /dev/null 2>&1
bash tools/gpu_profile.sh ${TAG}_blur_off --no-blur > /dev/null 2>&1
for c in cfg2 cfg3 cfg4 cfg5 naive_interp sharp; do bash tools/gpu_profile_cfg.sh $TAG $c > /dev/null 2>&1; bash tools/gpu_profile.sh ${TAG}_${c}_pmc --config $c > /dev/null 2>&1; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2>/dev/null
O=gpurun_out/${TAG}_ties; mkdir -p $O
for k in clipped random8; do
  timeout 900 python bench.py --depth $k --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${k}_blur_on.json 2>/dev/null
  timeout 1800 python bench.py --depth $k --no-blur --steps 2 --warmup 1 --no-cpu-baseline --frames 64 > $O/bench_${k}_blur_off.json 2>/dev/null
  for b in on off; do python3 -c "
import json; j=json.load(open('$O/bench_${k}_blur_$b.json')); print('$k blur $b', round(j['value'],1), 'fps', round(j['ms_per_step'],1), 'ms', j['config']['frames_total'], 'frames', j['diagnostics'])"; done
done 2>&1 | tee $O/summary.txt
{
for f in none naive naive_interpolating inverse polylines_soft polylines_sharp hybrid_edge gpu_warp; do
  printf "%-28s " $f; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 10 --fill $f 2>&1 | tail -1 | sed 's/.*: //'
done
printf "%-28s " "polylines_soft n=64"; timeout 300 python tools/quick_bench.py --n 64 --blur 0 --iters 5 --fill polylines_soft 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "polylines_sharp n=64"; timeout 300 python tools/quick_bench.py --n 64 --blur 0 --iters 5 --fill polylines_sharp 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "naive_interpolating n=64"; timeout 300 python tools/quick_bench.py --n 64 --blur 0 --iters 5 --fill naive_interpolating 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "polylines_soft anaglyph"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 10 --fill polylines_soft --mode red-cyan-anaglyph 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "polylines_sharp anaglyph"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 10 --fill polylines_sharp --mode red-cyan-anaglyph 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "gpu_warp 1080p n=128"; timeout 300 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --blur 0 --iters 10 --fill gpu_warp --kind radial --div 4.5 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "gpu_warp 1080p (k_gpuwarp)"; CS_PT_VARIANT=27 timeout 300 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --blur 0 --iters 10 --fill gpu_warp --kind radial --div 4.5 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "gpu_warp mesh 1080p"; CS_MESH=1 timeout 300 python tools/quick_bench.py --n 128 --h 1080 --w 1920 --blur 0 --iters 10 --fill gpu_warp --kind radial --div 4.5 2>&1 | tail -1 | sed 's/.*: //'
printf "%-28s " "polylines_sharp blur on"; timeout 300 python tools/quick_bench.py --n 32 --blur 1 --iters 10 --fill polylines_sharp 2>&1 | tail -1 | sed 's/.*: //'
for d in f64-disparity D64; do for f in none naive_interpolating inverse polylines_soft polylines_sharp; do
  printf "%-28s " "$f $d"; timeout 300 python tools/quick_bench.py --n 16 --blur 0 --iters 5 --fill $f --dialect $d 2>&1 | tail -1 | sed 's/.*: //'
done; done
} 2>&1 | tee gpurun_out/${TAG}_table.txt
for k in stepped scene8 blobs clipped; do for b in 0 1; do for f in none naive naive_interpolating inverse polylines_soft polylines_sharp hybrid_edge gpu_warp; do
  printf "%-8s blur %s %-22s " $k $b $f; timeout 300 python tools/quick_bench.py --n 16 --blur $b --iters 4 --fill $f --kind $k 2>&1 | tail -1 | sed 's/.*: //'
done; done; done 2>&1 | tee gpurun_out/${TAG}_table_by_depth.txt
timeout 600 python tools/node_host_bench.py --n 32 --iters 3 --prewarm 0 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_host_4k.txt
timeout 600 python tools/node_host_bench.py --n 32 --iters 3 --prewarm 0 --pin-cap-gb 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_host_4k_pinned.txt
timeout 600 python tools/node_host_bench.py --n 24 --iters 3 --prewarm 0 --fill "GPU Warp (Fast)" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_host_4k_gpuwarp.txt
timeout 600 python tools/node_host_bench.py --n 24 --iters 3 --prewarm 0 --pin-cap-gb 64 --fill "GPU Warp (Fast)" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_host_4k_gpuwarp_pinned.txt
head -12 gpurun_out/${TAG}_blur_on/kernel_trace.txt | cut -c1-140
python3 -c "
import json; j=json.load(open('gpurun_out/${TAG}_bench_default.json')); r=j['roofline']; print('metric', round(j['value'],1), 'fps; kernel_ms', round(r['kernel_ms'],3), 'frac (contract)', round(r['frac'],3), 'own', round(r['frac_own_bytes'],3), 'pipeline', round(r['pipeline_frac'],3), 'blur off', round(j.get('value_blur_off',0),1), 'other', j.get('value_other_depths'), 'tiles', r.get('blurred_tile_fraction'), 'cpu', j.get('cpu_baseline'))"
for c in cfg2 cfg3 cfg4 cfg5 naive_interp sharp; do python3 -c "
import json,sys
j=json.load(open('gpurun_out/${TAG}_$c/bench.json')); print('$c', round(j['value'],1), 'fps frac (contract)', round(j['roofline']['frac'],3), 'own', round(j['roofline']['frac_own_bytes'],3), 'kernel_ms', round(j['roofline']['kernel_ms'],3), 'pipeline', round(j['roofline']['pipeline_frac'],3))"; done
# HBM traffic and the vector-issue side of the dominant kernels from this run's PMC passes -> profiles/pmc_traffic.json, pmc_valu.json
python3 tools/make_traffic.py gpurun_out/${TAG}_blur_on 64 polylines_soft_4k_blur1 k_polypoint
python3 tools/make_traffic.py gpurun_out/${TAG}_blur_off 64 polylines_soft_4k_blur0 k_polypoint
python3 tools/make_traffic.py gpurun_out/${TAG}_cfg2_pmc 32 polylines_soft_1080p_blur1 k_polypoint
python3 tools/make_traffic.py gpurun_out/${TAG}_cfg3_pmc 16 hybrid_edge_4k_blur1 "k_hybrid_splat_tile+k_hybrid_gaps"
python3 tools/make_traffic.py gpurun_out/${TAG}_cfg4_pmc 256 gpu_warp_1080p_blur1 "k_gpuwarp_q<"
python3 tools/make_traffic.py gpurun_out/${TAG}_cfg5_pmc 64 none_4k_blur1 k_fwdtile
python3 tools/make_traffic.py gpurun_out/${TAG}_naive_interp_pmc 64 naive_interpolating_4k_blur1 k_fwdtile
python3 tools/make_traffic.py gpurun_out/${TAG}_sharp_pmc 64 polylines_sharp_4k_blur1 "k_polypoint<"
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
bash tools/make_valu_all.sh gpurun_out/${TAG} > gpurun_out/${TAG}_valu.txt 2>&1
cp profiles/pmc_valu.json gpurun_out/${TAG}_pmc_valu.json
python3 -c "
import json; d=json.load(open('profiles/pmc_valu.json'))
for k,v in d.items(): print(k, 'valu/wave', round(v['valu_per_wave']), 'busy', round(v['simd_busy'],2), 'floor_us', round(v.get('issue_floor_us',0)), 'of', round(v['kernel_us_profile']), '=', round(v.get('frac_of_issue_floor',0),3))"
