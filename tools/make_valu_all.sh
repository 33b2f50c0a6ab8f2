#!/bin/bash
# profiles/pmc_valu.json from the PMC passes of one closing session (TAG: directory prefix under profiles/ or gpurun_out/), every bench
# config: tools/make_valu_all.sh profiles/r05d   (cfg 4's kernel name: pass GW="k_gpuwarp<8, 2, false, true>" for profiles older than round 6)
T=${1:-profiles/r06a}
GW=${GW:-"k_gpuwarp_q<8, 2>"}
GWRE=${GWRE:-"k_gpuwarp_qILi8ELi2E"}
python3 tools/make_valu.py ${T}_blur_on polylines_soft_4k_blur1 "k_polypoint<256, 4, 0, 4, 5, 7, 0, 0>" comfystereo_amd/csrc/cs_polypoint.hip "k_polypointILi256ELi4ELi0ELi4ELi5ELi7ELi0ELi0E"
python3 tools/make_valu.py ${T}_blur_off polylines_soft_4k_blur0 "k_polypoint<256, 4, 0, 4, 5, 7, 0, 0>" comfystereo_amd/csrc/cs_polypoint.hip "k_polypointILi256ELi4ELi0ELi4ELi5ELi7ELi0ELi0E"
python3 tools/make_valu.py ${T}_cfg2_pmc polylines_soft_1080p_blur1 "k_polypoint<256, 4, 0, 4, 5, 7, 0, 0>" comfystereo_amd/csrc/cs_polypoint.hip "k_polypointILi256ELi4ELi0ELi4ELi5ELi7ELi0ELi0E"
python3 tools/make_valu.py ${T}_cfg3_pmc hybrid_edge_4k_blur1 "k_hybrid_splat_tile<true, false>" comfystereo_amd/csrc/cs_rowwarp.hip "k_hybrid_splat_tileILb1ELb0E"
python3 tools/make_valu.py ${T}_cfg4_pmc gpu_warp_1080p_blur1 "$GW" comfystereo_amd/csrc/cs_gpuwarp.hip "$GWRE"
python3 tools/make_valu.py ${T}_cfg5_pmc none_4k_blur1 "k_fwdtile<256, 3, 0, false, false>" comfystereo_amd/csrc/cs_fwdtile.hip "k_fwdtileILi256ELi3ELi0ELb0ELb0E"
python3 tools/make_valu.py ${T}_naive_interp_pmc naive_interpolating_4k_blur1 "k_fwdtile<256, 4, 2, false, false>" comfystereo_amd/csrc/cs_fwdtile.hip "k_fwdtileILi256ELi4ELi2ELb0ELb0E"
python3 tools/make_valu.py ${T}_sharp_pmc polylines_sharp_4k_blur1 "k_polypoint<256, 4, 0, 6, 9, 6, 1, 0>" comfystereo_amd/csrc/cs_polypoint.hip "k_polypointILi256ELi4ELi0ELi6ELi9ELi6ELi1ELi0E"
