# Random parity campaigns under the round-6 product options (opencv_fma 1 / 2 with two lane widths, remap_model 1), the oracle
# switched the same way: narrow and wide windows, integer volumes, multi-band images.
# usage (through gpurun): bash tools/campaign_options.sh <seed> [scale]  -> gpurun_out/campaign_opt_*.log
s=${1:-7600}; k=${2:-1}
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python "$@" > gpurun_out/campaign_opt_$name.log 2>&1; echo "$name: $(tail -1 gpurun_out/campaign_opt_$name.log)"; }
FDN_OPENCV_FMA=1 run fma1_narrow tools/random_campaign.py $((150*k)) $s
FDN_OPENCV_FMA=1 run fma1_wide tools/random_campaign.py $((100*k)) $((s+1)) wide
FDN_OPENCV_FMA=2 run fma2_narrow tools/random_campaign.py $((150*k)) $((s+2))
FDN_OPENCV_FMA=2 FDN_OPENCV_FMA_LANES=16 run fma2_16_wide tools/random_campaign.py $((100*k)) $((s+3)) wide
FDN_OPENCV_FMA=2 FDN_OPENCV_FMA_LANES=4 run fma2_4_big tools/random_campaign_big.py $((20*k)) $((s+4))
FDN_REMAP_MODEL=1 run remap1_narrow tools/random_campaign.py $((150*k)) $((s+5))
FDN_REMAP_MODEL=1 run remap1_wide tools/random_campaign.py $((100*k)) $((s+6)) wide
FDN_REMAP_MODEL=1 FDN_OPENCV_FMA=1 run remap1_fma1_int tools/random_campaign_int.py $((100*k)) $((s+7))
FDN_REMAP_MODEL=1 FDN_SUB_BATCHES=1 run remap1_one_stream tools/random_campaign.py $((100*k)) $((s+8))
