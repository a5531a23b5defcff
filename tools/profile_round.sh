#!/bin/bash
# One GPU-box pass that produces every rocprofv3 artefact of a round (copy the summaries from gpurun_out/<tag>/ to profiles/):
#   tools/profile_round.sh <tag>          (run from the repo root on the MI355X box)
#  1. kernel trace of the HEADLINE configuration (hipGraph replay, side streams on)        -> <tag>_kernel_stats.csv
#  2. kernel trace of `bench.py --no-overlap` (one stream: durations are additive)         -> <tag>_serial_kernel_stats.csv,
#     <tag>_serial_timeline.txt, <tag>_serial_bench.json  (sum of the conv kernels == roofline.achieved of that line)
#  3. PMC passes, each in its own run: FETCH_SIZE, WRITE_SIZE (HBM traffic of the conv launches) -> <tag>_pmc_traffic.json
#  4. PMC pass: SQ counters of the conv families (MFMA busy, waits, LDS conflicts)         -> <tag>_sq_pmc.txt
#  5. kernel traces + per-launch conv tables of BASELINE configs 3 and 5 as stated (tools/bench_configs.py, tools/layer_table.py)
#     -> <tag>_c3_kernel_stats.csv, <tag>_c3_conv_layers.txt, <tag>_c5_km3d_kernel_stats.csv, <tag>_c5_km3d_conv_layers.txt
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONPATH=$ROOT
cd /tmp
run() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    rm -rf $OUT/$name
    timeout 600 rocprofv3 "$@" > $OUT/$name.log 2>&1
    echo "[$name] rc=$?"
}
run trace_overlap --kernel-trace --stats -d $OUT/trace_overlap -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs
run trace_serial --kernel-trace --stats -d $OUT/trace_serial -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-overlap
run pmc_fetch --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-other-configs
run pmc_write --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -- python $ROOT/bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-other-configs
run pmc_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-overlap --no-cpu-baseline --no-other-configs
VD3D_CONV_NO_STAGGER=1 run pmc_sq_nostg --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq_nostg -- python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-overlap --no-cpu-baseline --no-other-configs
run trace_c3 --kernel-trace --stats -d $OUT/trace_c3 -- python $ROOT/tools/bench_configs.py "C3 Stereo3D R50 +"
run trace_c5 --kernel-trace --stats -d $OUT/trace_c5 -- python $ROOT/tools/bench_configs.py "fp16 (as BASELINE"
# the reference's own call pattern: one frame per module([...]) call through the hipGraph cache (BASELINE config 1 and the stereo pair at batch 1)
run trace_b1_mono --kernel-trace --stats -d $OUT/trace_b1_mono -- python $ROOT/tools/bench_b1.py --calls 30 mono
run trace_b1_stereo --kernel-trace --stats -d $OUT/trace_b1_stereo -- python $ROOT/tools/bench_b1.py --calls 30 stereo
# config 5's own SQ / traffic passes (DCN and fused-head kernels), eager launches so that every dispatch carries its counters
run pmc_c5_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_c5_sq -- python $ROOT/tools/bench_configs.py --eager "fp16 (as BASELINE"
run pmc_c5_valu --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc_c5_valu -- python $ROOT/tools/bench_configs.py --eager "fp16 (as BASELINE"
run pmc_c5_fetch --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_c5_fetch -- python $ROOT/tools/bench_configs.py --eager "fp16 (as BASELINE"
run pmc_c5_write --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_c5_write -- python $ROOT/tools/bench_configs.py --eager "fp16 (as BASELINE"
# config 3's own SQ / traffic passes (the 2176-channel strips, the column GEMM, the columns kernel, the point-wise streaming kernels), eager launches
run pmc_c3_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_c3_sq -- python $ROOT/tools/bench_configs.py --eager "C3 Stereo3D R50 +"
run pmc_c3_fetch --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_c3_fetch -- python $ROOT/tools/bench_configs.py --eager "C3 Stereo3D R50 +"
run pmc_c3_write --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_c3_write -- python $ROOT/tools/bench_configs.py --eager "C3 Stereo3D R50 +"
cd $ROOT
db() { find $OUT/$1 -name "*.db" | sort | tail -1; }
python tools/rocpd_stats.py $(db trace_overlap) > $OUT/${TAG}_kernel_stats.csv
python tools/rocpd_stats.py $(db trace_serial) > $OUT/${TAG}_serial_kernel_stats.csv
python tools/rocpd_timeline.py $(db trace_serial) > $OUT/${TAG}_serial_timeline.txt
grep '^{' $OUT/trace_serial.log | tail -1 > $OUT/${TAG}_serial_bench.json
grep '^{' $OUT/trace_overlap.log | tail -1 > $OUT/${TAG}_overlap_bench_under_rocprof.json
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json
python tools/rocpd_pmc_table.py $OUT/pmc_sq conv_ > $OUT/${TAG}_sq_pmc.txt
python tools/family_busy.py $OUT/${TAG}_sq_pmc.txt > $OUT/${TAG}_family_mfma_busy.txt
python tools/rocpd_pmc_table.py $OUT/pmc_sq_nostg conv_ > $OUT/${TAG}_sq_pmc_no_stagger.txt
{ echo "# the same SQ pass with VD3D_CONV_NO_STAGGER=1 (every wave of the 352 / 288 strips on one DMA schedule)"; python tools/family_busy.py $OUT/${TAG}_sq_pmc_no_stagger.txt; } >> $OUT/${TAG}_family_mfma_busy.txt
{ echo "# BASELINE config 5 (KM3D DLA-34, fp16, 16 x 512 x 1760), eager launches: DCN, fused-head (km3d_head_kernel) and level-pair kernels"; echo "## SQ pass"; python tools/rocpd_pmc_table.py $OUT/pmc_c5_sq dcn_; python tools/rocpd_pmc_table.py $OUT/pmc_c5_sq km3d_head_kernel | tail -n +2; python tools/rocpd_pmc_table.py $OUT/pmc_c5_sq conv_pair_kernel | tail -n +2;
  echo "## instruction mix"; python tools/rocpd_pmc_table.py $OUT/pmc_c5_valu dcn_; python tools/rocpd_pmc_table.py $OUT/pmc_c5_valu km3d_head_kernel | tail -n +2; python tools/rocpd_pmc_table.py $OUT/pmc_c5_valu conv_pair_kernel | tail -n +2;
  echo "## FETCH_SIZE (x2 for bytes on gfx950: 32-byte units reported as 64)"; python tools/rocpd_pmc_table.py $OUT/pmc_c5_fetch dcn_; python tools/rocpd_pmc_table.py $OUT/pmc_c5_fetch km3d_head_kernel | tail -n +2; python tools/rocpd_pmc_table.py $OUT/pmc_c5_fetch conv_pair_kernel | tail -n +2;
  echo "## WRITE_SIZE"; python tools/rocpd_pmc_table.py $OUT/pmc_c5_write dcn_; python tools/rocpd_pmc_table.py $OUT/pmc_c5_write km3d_head_kernel | tail -n +2; python tools/rocpd_pmc_table.py $OUT/pmc_c5_write conv_pair_kernel | tail -n +2; } > $OUT/${TAG}_c5_pmc.txt 2>&1
{ echo "# BASELINE config 3 (Stereo3D R50 + base DCNv2 head, bf16, 32 x 288 x 1280), eager launches: conv families, DCN columns kernel"; echo "## SQ pass"; python tools/rocpd_pmc_table.py $OUT/pmc_c3_sq conv_; python tools/rocpd_pmc_table.py $OUT/pmc_c3_sq dcn_ | tail -n +2;
  echo "## FETCH_SIZE (x2 for bytes on gfx950: 32-byte units reported as 64)"; python tools/rocpd_pmc_table.py $OUT/pmc_c3_fetch conv_; python tools/rocpd_pmc_table.py $OUT/pmc_c3_fetch dcn_ | tail -n +2;
  echo "## WRITE_SIZE"; python tools/rocpd_pmc_table.py $OUT/pmc_c3_write conv_; python tools/rocpd_pmc_table.py $OUT/pmc_c3_write dcn_ | tail -n +2; } > $OUT/${TAG}_c3_pmc.txt
python tools/serial_roofline_check.py $OUT/${TAG}_serial_kernel_stats.csv $OUT/${TAG}_serial_bench.json > $OUT/${TAG}_serial_roofline_check.txt
cat $OUT/${TAG}_serial_roofline_check.txt
python tools/rocpd_timeline.py $(db trace_b1_mono) > $OUT/${TAG}_b1_mono_timeline.txt
python tools/rocpd_timeline.py $(db trace_b1_stereo) > $OUT/${TAG}_b1_stereo_timeline.txt
grep 'ms per' $OUT/trace_b1_mono.log $OUT/trace_b1_stereo.log > $OUT/${TAG}_b1_under_rocprof.txt
python tools/bench_b1.py 2>/dev/null | grep 'ms per' > $OUT/${TAG}_b1_calls.txt
python tools/rocpd_stats.py $(db trace_c3) > $OUT/${TAG}_c3_kernel_stats.csv
python tools/rocpd_stats.py $(db trace_c5) > $OUT/${TAG}_c5_km3d_kernel_stats.csv
grep 'img/s' $OUT/trace_c3.log $OUT/trace_c5.log > $OUT/${TAG}_c3_c5_under_rocprof.txt
python tools/layer_table.py "C3 Stereo3D R50 +" 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_c3_conv_layers.txt
python tools/layer_table.py "fp16 (as BASELINE" 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_c5_km3d_conv_layers.txt
python tools/bench_configs.py 2>/dev/null | grep 'img/s' > $OUT/${TAG}_bench_configs.txt
# the headline line of this box with default flags (right after the profiling passes), the same through the RCCL path at world size 1
# (VD3D_BENCH_FORCE_DIST: process group, graph-captured pack, double-buffered all_gather, D2H) and with host-fed uint8 frames (--feed host)
cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json      # (the bench line below stamps its `roofline.traffic` from THIS pass)
python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench_line.err; tail -1 $OUT/bench_line.err
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 VD3D_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-other-configs 2> $OUT/bench_force_dist.err | grep '^{' | tail -1 > $OUT/${TAG}_bench_force_dist.json; tail -1 $OUT/bench_force_dist.err
# the driver's own N > 1 command line at N = 1 (through torch.distributed.run), RCCL path forced on: stdout must be the one JSON line
VD3D_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/${TAG}_bench_torchrun_nproc1.json 2> $OUT/bench_torchrun.err; tail -1 $OUT/bench_torchrun.err
python bench.py --feed host --no-cpu-baseline --no-other-configs > $OUT/${TAG}_bench_feed_host.json 2> $OUT/bench_feed_host.err; tail -1 $OUT/bench_feed_host.err
# the rocpd databases stay on the box (too big); only the summaries travel back
rm -rf $OUT/pmc_sq_nostg $OUT/trace_b1_mono $OUT/trace_b1_stereo $OUT/trace_overlap $OUT/trace_serial $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/trace_c3 $OUT/trace_c5 $OUT/pmc_c5_sq $OUT/pmc_c5_valu $OUT/pmc_c5_fetch $OUT/pmc_c5_write $OUT/pmc_c3_sq $OUT/pmc_c3_fetch $OUT/pmc_c3_write
ls -la $OUT
