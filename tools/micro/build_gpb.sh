#!/bin/bash
# builds the gemm_pl_bench variants (tools/micro/gpb_*), no GPU needed
cd "$(dirname "$0")/../.." || exit 1
F="-O3 -std=c++17 --offload-arch=gfx950 -I reid-gan_amd/csrc tools/micro/gemm_pl_bench.hip"
H=/opt/rocm/bin/hipcc
$H $F -o tools/micro/gpb_full 2>/dev/null &
$H $F -DBENCH_NPF=2 -o tools/micro/gpb_pf2 2>/dev/null &
$H $F -DBENCH_NPF=2 -DXCD_REMAP -o tools/micro/gpb_pf2_xcd 2>/dev/null &
$H $F -DBENCH_NPF=2 -DXCD_REMAP -DLDA_PAD=32 -o tools/micro/gpb_pf2_xcd_pad 2>/dev/null &
$H $F -DBENCH_NPF=2 -DXCD_REMAP -DCONSUMER_PRIO=3 -o tools/micro/gpb_pf2_xcd_prio 2>/dev/null &
$H $F -DBENCH_NPF=2 -DSAME_TILE_LOADS -o tools/micro/gpb_pf2_same 2>/dev/null &
$H $F -DXCD_REMAP -o tools/micro/gpb_xcd 2>/dev/null &
$H $F -DNO_GLOAD -o tools/micro/gpb_nogload 2>/dev/null &
wait
ls -la tools/micro/gpb_*
